"""MI355XBackend -- the comms plug-in for one process per GPU on an 8 x MI355X xGMI node.

Implements the ``backendFunctions`` interface (reference ``train/comms/pt/pytorch_backend_utils.py:
156-411``) the way ``PyTorchDistBackend`` does for c10d (reference ``pytorch_dist_backend.py``), for
the collectives on the DLRM sparse-feature path plus the ``emb_lookup`` compute function:

  * collectives go through ``torch.distributed`` (backend ``"nccl"`` IS RCCL on ROCm; ``"gloo"`` is
    accepted so the host logic is testable on CPU).  An all-to-all maps 1:1 onto the xGMI mesh
    (one direct link per peer), so it is issued as ONE ``all_to_all_single`` per call -- the
    per-peer grouped send/recv RCCL runs underneath -- never re-chunked on the host;
  * async collectives run on the process group's own HIP stream: ``asyncOp=True`` returns a work
    handle immediately and the collective overlaps whatever the compute stream does next
    (the embedding lookup of the next table group / batch chunk);
  * ``emb_lookup`` / ``alloc_embedding_tables`` are served by the hand-written HIP kernels
    (:mod:`param_amd.embedding_bag`), not by torch or fbgemm;
  * ``lookup_all_to_all`` is the fused entry the reference reaches at
    ``pytorch_dist_backend.py:214-234`` (one lookup per table group, each group's pooled output
    handed to the all-to-all while the next group is being looked up).

Register it with :func:`register` (own registry, and the reference's when ``param_bench`` is
importable) and select it with ``--backend rccl_xgmi``.
"""
from __future__ import annotations

import logging
import os
from time import sleep

import numpy as np
import torch
import torch.distributed as dist

from .pytorch_backend_utils import backendFunctions, register_customized_backend

logger = logging.getLogger(__name__)

BACKEND_NAME = "rccl_xgmi"


class MI355XBackend(backendFunctions):
    def __init__(self, bootstrap_info, commsParams) -> None:
        super().__init__()
        self.bootstrap_info = bootstrap_info
        self.commsParams = commsParams
        self.use_ext_dist = False
        self.groups = {}
        self.num_pgs = 0
        self.collectiveFunc["wait"] = self.wait
        self.computeFunc["emb_lookup"] = self.emb_lookup

    # ------------------------------------------------------------------ helpers
    def _cp(self, name, default=None):
        cp = self.commsParams
        return cp.get(name, default) if isinstance(cp, dict) else getattr(cp, name, default)

    def _is_gpu(self) -> bool:
        return self._cp("device", "cuda") in ("cuda", "rocm")

    def _group(self, collectiveArgs):
        return collectiveArgs.group if collectiveArgs.group is not None else self.get_default_group()

    def _post(self, collectiveArgs, work, retFlag):
        if collectiveArgs.asyncOp:
            collectiveArgs.waitObj.append(work)
        if retFlag:
            return work
        return None

    def sayHello(self, global_rank, local_rank, world_size, master_ip):
        myhost = os.uname()[1]
        dev = self.get_device()
        hw = torch.cuda.get_device_name(dev) if self._is_gpu() and torch.cuda.is_available() else "cpu"
        print(f"[Rank {global_rank:>3}] host {myhost}, device: {dev} ({hw}), local_rank: {local_rank} "
              f"world_size: {world_size}, master_ip: {master_ip}")

    # ------------------------------------------------------------------ collectives
    def all_reduce(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        work = dist.all_reduce(collectiveArgs.ipTensor, op=collectiveArgs.op or dist.ReduceOp.SUM,
                               group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def reduce(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        work = dist.reduce(collectiveArgs.ipTensor, dst=collectiveArgs.srcOrDst,
                           op=collectiveArgs.op or dist.ReduceOp.SUM, group=self._group(collectiveArgs),
                           async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_to_all(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        """List-form all_to_all (``opTensor`` / ``ipTensor`` are lists of per-peer tensors), or the
        fused lookup+all-to-all when batched embedding tables are attached (reference
        ``pytorch_dist_backend.py:207-260``)."""
        if collectiveArgs.num_emb_tables_batched > 0 and collectiveArgs.emb is not None:
            work = self.lookup_all_to_all(collectiveArgs)
        elif isinstance(collectiveArgs.opTensor, (list, tuple)):
            if dist.get_backend(self._group(collectiveArgs)) == "gloo":
                # gloo has no list-form alltoall (reference survey probe): flatten to the single-tensor form
                ip = torch.cat([t.reshape(-1) for t in collectiveArgs.ipTensor])
                op = torch.empty(sum(t.numel() for t in collectiveArgs.opTensor), dtype=ip.dtype, device=ip.device)
                w = dist.all_to_all_single(op, ip, [t.numel() for t in collectiveArgs.opTensor],
                                           [t.numel() for t in collectiveArgs.ipTensor],
                                           group=self._group(collectiveArgs))
                assert w is None
                o = 0
                for t in collectiveArgs.opTensor:
                    t.copy_(op[o:o + t.numel()].view_as(t))
                    o += t.numel()
                work = None
            else:
                work = dist.all_to_all(collectiveArgs.opTensor, collectiveArgs.ipTensor,
                                       group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        else:
            work = dist.all_to_all_single(collectiveArgs.opTensor, collectiveArgs.ipTensor,
                                          group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_to_allv(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        work = dist.all_to_all_single(
            collectiveArgs.opTensor, collectiveArgs.ipTensor,
            list(collectiveArgs.opTensor_split) if collectiveArgs.opTensor_split is not None and len(collectiveArgs.opTensor_split) else None,
            list(collectiveArgs.ipTensor_split) if collectiveArgs.ipTensor_split is not None and len(collectiveArgs.ipTensor_split) else None,
            group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_to_all_single(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        return self.all_to_allv(collectiveArgs, retFlag)

    def wait(self, collectiveArgs, retFlag=False):
        """With request ids recorded (trace replay: ``waitObjIds[req] = work``) wait on the request named by
        ``collectiveArgs.collectiveId``; otherwise on the FIRST outstanding request only (reference
        ``wait`` / ``complete_single_op``, pytorch_dist_backend.py:724-744)."""
        if collectiveArgs.waitObjIds:
            w = collectiveArgs.waitObjIds.get(collectiveArgs.collectiveId)
            if w is not None:
                w.wait()
            return
        if collectiveArgs.waitObj:
            w = collectiveArgs.waitObj.pop(0)
            if w is not None:
                w.wait()
            self.device_sync(collectiveArgs)

    def complete_accel_ops(self, collectiveArgs, devSync=True):
        for w in collectiveArgs.waitObj:
            if w is not None:
                w.wait()
        if devSync:
            self.device_sync(collectiveArgs)
        collectiveArgs.waitObj.clear()
        collectiveArgs.waitObjIds.clear()

    def device_sync(self, collectiveArgs=None):
        if self._is_gpu() and torch.cuda.is_available():
            torch.cuda.synchronize(self.get_device())

    def barrier(self, collectiveArgs, name="dummy", retFlag=False):
        group = self._group(collectiveArgs)
        ids = [self.get_device().index] if dist.get_backend(group) == "nccl" else None
        work = dist.barrier(group, async_op=bool(collectiveArgs.asyncOp), device_ids=ids)
        return self._post(collectiveArgs, work, retFlag)

    def sync_barrier(self, collectiveArgs, desc="dummy"):
        # all streams drained before and after the barrier: it fences timed regions (dlrm.py:119 ...)
        self.complete_accel_ops(collectiveArgs)
        self.barrier(collectiveArgs, name=desc)
        self.complete_accel_ops(collectiveArgs)

    def get_reduce_op(self, opName: str):
        return {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}.get(opName, dist.ReduceOp.SUM)

    # ------------------------------------------------------------------ compute
    def gemm(self, collectiveArgs) -> None:
        raise NotImplementedError("the GEMM overlap kernel is outside the MI355X embedding build (SURVEY.md 2.2 X1)")

    def emb_lookup(self, collectiveArgs):
        """Batched embedding lookup as the compute kernel (reference ``pytorch_dist_backend.py:832-857``):
        forward = one HIP launch per op over all of its tables; backward = fused in-place update."""
        if collectiveArgs.direction == "forward":
            for i, (indices, offsets, weights) in enumerate(collectiveArgs.embRequests):
                collectiveArgs.LookupOut = collectiveArgs.emb[i].forward(indices, offsets, weights)
        else:
            for i, (indices, offsets, weights) in enumerate(collectiveArgs.embRequests):
                collectiveArgs.emb[i].scatter_add_(collectiveArgs.grad_output, indices, offsets,
                                                   alpha=-collectiveArgs.emb[i].learning_rate,
                                                   per_sample_weights=weights)

    def lookup_all_to_all(self, collectiveArgs):
        """Pipelined lookup -> pooled all-to-all (reference ``:214-234``): op i's pooled embeddings
        ``[global_batch, tables_i * D]`` leave through RCCL while op i+1 is being looked up.  Returns the
        list of work handles (already waited unless ``asyncOp``)."""
        works = []
        outs = getattr(collectiveArgs, "a2a_recv", None)
        if outs is None:
            outs = collectiveArgs.a2a_recv = [None] * collectiveArgs.num_emb_ops
        for i in range(collectiveArgs.num_emb_ops):
            indices, offsets, weights = collectiveArgs.embRequests[i]
            pooled = collectiveArgs.emb[i].lookup(indices, offsets, weights)
            if outs[i] is None or outs[i].shape != pooled.shape:
                outs[i] = torch.empty_like(pooled)
            works.append(dist.all_to_all_single(outs[i], pooled, group=self._group(collectiveArgs), async_op=True))
        if not collectiveArgs.asyncOp:
            for w in works:
                w.wait()
        return works

    # ------------------------------------------------------------------ memory
    def get_mem_size(self, collectiveArgs, pair=False, pairIdx=0) -> int:
        """bytes of the OUTPUT tensor(s) (reference ``:860-897``): the algBW numerator"""
        op, ip = collectiveArgs.opTensor, collectiveArgs.ipTensor
        if isinstance(op, (list, tuple)):
            return sum(t.nelement() * t.element_size() for t in op)
        if isinstance(ip, (list, tuple)):
            return sum(t.nelement() * t.element_size() for t in ip)
        return op.nelement() * op.element_size()

    def alloc_random(self, sizeArr, curRankDevice="cuda", dtype=torch.float32, scaleFactor=1.0):
        if dtype in (torch.int8, torch.uint8, torch.short, torch.int16, torch.int32, torch.long):
            return torch.randint(low=0, high=10, size=tuple(sizeArr), device=curRankDevice, dtype=dtype)
        if dtype == torch.bool:
            return torch.rand(sizeArr, device=curRankDevice, dtype=torch.float32) < 0.5
        t = torch.rand(sizeArr, device=curRankDevice, dtype=dtype)
        return t / scaleFactor if scaleFactor != 0 else t

    def alloc_ones(self, sizeArr, curRankDevice="cuda", dtype=torch.float32, scaleFactor=1.0):
        t = torch.ones(sizeArr, device=curRankDevice, dtype=dtype)
        return t * scaleFactor if scaleFactor != 1.0 else t

    def alloc_embedding_tables(self, n, m, curRankDevice, dtype):
        """A callable ``E(indices, offsets)`` with ``.weight.data`` (reference ``:923-934``), U(-1/sqrt(n),
        1/sqrt(n)) initialised, served by the HIP forward."""
        from ...embedding_bag import EmbeddingBagMI355, fill_random_

        w = torch.empty(n, m, dtype=dtype, device=curRankDevice)
        lim = float(np.sqrt(1.0 / n))
        if w.is_cuda:
            fill_random_(w, "uniform", -lim, lim, seed=int(np.random.randint(0, 2**31 - 1)))
        else:
            w.uniform_(-lim, lim)
        return EmbeddingBagMI355(n, m, mode="sum", _weight=w)

    def alloc_batched_embedding_tables(self, rows, dim, curRankDevice, dtype, layout="bd"):
        from ...embedding_bag import BatchedEmbeddingBagMI355

        return BatchedEmbeddingBagMI355(rows, dim, dtype=dtype, device=curRankDevice, layout=layout,
                                        init="uniform_dlrm", seed=int(np.random.randint(0, 2**31 - 1)))

    def alloc_empty(self, sizeArr, curRankDevice, dtype):
        # the reference's concrete backends take (size, device, dtype) although the ABC says (size, dtype, device)
        if isinstance(curRankDevice, torch.dtype):
            curRankDevice, dtype = dtype, curRankDevice
        return torch.empty(sizeArr, device=curRankDevice, dtype=dtype)

    def clear_memory(self, collectiveArgs):
        del collectiveArgs.ipTensor
        del collectiveArgs.opTensor
        collectiveArgs.ipTensor, collectiveArgs.opTensor = [], []
        if self._is_gpu() and torch.cuda.is_available():
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ topology
    def get_local_rank(self):
        return self.bootstrap_info.local_rank

    def get_global_rank(self):
        return dist.get_rank()

    def get_world_size(self):
        return dist.get_world_size()

    def get_local_size(self):
        return self.bootstrap_info.local_size

    def get_group_rank(self, group):
        return dist.get_rank(group)

    def get_group_size(self, group):
        return dist.get_world_size(group)

    def get_device(self):
        if self._is_gpu():
            n = max(1, torch.cuda.device_count())
            return torch.device("cuda", self.get_local_rank() % n)
        return torch.device("cpu")

    def get_hw_device(self):
        return self.get_device()

    def get_default_group(self):
        return dist.group.WORLD

    def get_groups(self):
        return self.groups

    def get_num_pgs(self):
        return self.num_pgs

    def tensor_list_to_numpy(self, tensorList):
        if isinstance(tensorList, list):
            tensorList = [t.cpu().detach().numpy() for t in tensorList]
        return np.array(tensorList)

    # ------------------------------------------------------------------ streams / events
    def get_new_stream(self):
        return torch.cuda.Stream(device=self.get_device(), priority=0) if self._is_gpu() else None

    def get_new_event(self, enable_timing=False):
        return torch.cuda.Event(enable_timing) if self._is_gpu() else None

    def get_current_stream(self, device=None):
        return torch.cuda.current_stream(device) if self._is_gpu() else None

    def switch_stream(self, stream, device=None):
        if device is None:
            device = self.get_device()
        if stream is not None and device.type == "cuda":
            cur = torch.cuda.current_stream(device=device)
            torch.cuda.set_stream(stream)
            return cur
        return None

    def sync_stream(self, stream=None, device=None):
        if device is None:
            device = self.get_device()
        if device.type == "cuda":
            (stream if stream is not None else torch.cuda.current_stream(device=device)).synchronize()

    # ------------------------------------------------------------------ init
    def store_set(self, key, val):
        self.tcp_store.set(key, val)

    def store_get(self, key):
        return self.tcp_store.get(key)

    def set_device(self, local_rank, global_rank):
        if self._is_gpu():
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))

    def initialize_backend(self, master_ip, master_port, backend="nccl", eager_mode=False) -> None:
        """TCPStore rendezvous + init_process_group, as the reference (``:1145-1200``); RCCL needs
        HSA_ENABLE_IPC_MODE_LEGACY=0 for dmabuf IPC (set if absent)."""
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.set_device(self.bootstrap_info.local_rank, self.bootstrap_info.global_rank)
        rank, world = self.bootstrap_info.global_rank, self.bootstrap_info.world_size
        if backend == BACKEND_NAME or backend in ("rccl", "rocm"):
            backend = "nccl"
        if not dist.is_initialized():
            if self.tcp_store is None:
                self.tcp_store = dist.TCPStore(master_ip, int(master_port), world, is_master=(rank == 0),
                                               use_libuv=True)
            dist.init_process_group(backend, rank=rank, world_size=world, store=self.tcp_store,
                                    device_id=self.get_device() if (eager_mode and self._is_gpu()) else None)
        self.groups = {0: self.get_default_group()}
        self.num_pgs = 1

    def initialize_groups(self, groupRanks=None, backend="nccl", force_new_group=False):
        groups = {}
        for pg_id, ranks in (groupRanks or {}).items():
            if len(ranks) == self.get_world_size() and not force_new_group:
                groups[pg_id] = self.get_default_group()
            else:
                groups[pg_id] = dist.new_group(ranks=ranks, backend="nccl" if backend == BACKEND_NAME else backend)
        if groups:
            self.groups = groups
        self.num_pgs = len(self.groups)

    def benchmark_comms(self, benchTime, commsParams) -> None:
        if getattr(commsParams, "init_only", False):
            sleep(10)
        else:
            benchTime(0, commsParams, self)

    def shutdown(self):
        if dist.is_initialized():
            dist.destroy_process_group()


def register() -> None:
    """Register under ``--backend rccl_xgmi`` in this package's registry and, when the reference
    package is importable (``param_bench`` on PYTHONPATH), in the reference's own
    ``customized_backend`` table so its unmodified ``comms.py`` can select it
    (reference comms.py:1507-1524)."""
    register_customized_backend(BACKEND_NAME, MI355XBackend, device="rocm")
    try:
        from param_bench.train.comms.pt import pytorch_backend_utils as ref_utils  # type: ignore

        ref_utils.register_customized_backend(BACKEND_NAME, MI355XBackend, "rocm")
    except Exception:  # reference not installed: nothing to do
        pass
