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
from itertools import cycle
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
        # the fused lookup -> all-to-all entry of ``all_to_all`` is what the reference only takes with its
        # ``extend_distributed`` package (pytorch_dist_backend.py:214); here it is an explicit opt-in
        self.fused_lookup_a2a = bool(self._cp("fused_lookup_a2a", False)) or os.environ.get("PARAM_AMD_FUSED_LOOKUP_A2A") == "1"
        self.groups = {}
        self.num_pgs = 0
        self.round_robin_group = None
        # every entry of the reference ABC's table (pytorch_backend_utils.py:161-180) + its PyTorchDistBackend extras
        # (pytorch_dist_backend.py:1100-1115): the reference drivers index this dict by ``--collective`` name
        self.collectiveFunc.update({
            "broadcast": self.broadcast, "broadcast_object_list": self.broadcast_object_list, "gather": self.gather,
            "all_gather": self.all_gather, "all_gather_base": self.all_gather_base,
            "all_gather_object": self.all_gather_object, "reduce_scatter": self.reduce_scatter,
            "reduce_scatter_base": self.reduce_scatter_base, "scatter": self.scatter, "incast": self.incast,
            "multicast": self.multicast, "wait": self.wait, "send": self.send, "recv": self.recv,
            "isend": self.isend, "irecv": self.irecv, "pt2pt": self.noop,
        })
        self.computeFunc["emb_lookup"] = self.emb_lookup

    # ------------------------------------------------------------------ helpers
    def _cp(self, name, default=None):
        cp = self.commsParams
        return cp.get(name, default) if isinstance(cp, dict) else getattr(cp, name, default)

    def _is_gpu(self) -> bool:
        return self._cp("device", "cuda") in ("cuda", "rocm")

    def _group(self, collectiveArgs):
        return collectiveArgs.group if collectiveArgs.group is not None else self.get_default_group()

    get_collective_group = _group   # the reference's name for it (pytorch_dist_backend.py:79)

    def _post(self, collectiveArgs, work, retFlag):
        if collectiveArgs.asyncOp and work is not None:
            collectiveArgs.waitObj.append(work)
        if retFlag:
            return work
        return None

    @staticmethod
    def _io(collectiveArgs, pair, pairIdx):
        """(input, output) tensors of the call: the pair-mode twins when ``pair`` (comms.py:392-409)"""
        if pair:
            return collectiveArgs.ipTensor_pair[pairIdx], collectiveArgs.opTensor_pair[pairIdx]
        return collectiveArgs.ipTensor, collectiveArgs.opTensor

    # ------------------------------------------------------------------ quantised payloads (--bitwidth < 32)
    def _timed(self, timer, fn):
        """run ``fn`` under the reference's quant / de-quant timers (``paramProfile(timer=...)``,
        pytorch_dist_backend.py:124-130): host clock, closed by a device sync on the GPU so that the report's
        quant / comms / de-quant split (comms.py:1028-1036) is one of device work, not of launch times"""
        import time

        t0 = time.monotonic_ns()
        out = fn()
        self.device_sync()
        if timer is not None and hasattr(timer, "incrTimeNS"):
            timer.incrTimeNS(float(time.monotonic_ns() - t0))
        return out

    # Row codec for HOST tensors: ``bits -> (prepack(rows_f32[n, dim]) -> uint8[n, row_bytes], unpack(uint8[n, row_bytes]) -> f32[n, dim])``.
    # None in the product: the quantised collectives run on the HIP row quantisers (param_amd.quant) and a host tensor raises.  The
    # gloo tests, which drive this plug-in without a GPU (``--device cpu``), inject torch's CPU operators for the same formats here
    # (tests/dist_workers.py) -- the operators oracle/rowquant.py is pinned to; they are a checker, not a fallback the backend ships.
    host_row_codec = None

    def _host_codec(self, bits):
        if self.host_row_codec is None:
            raise RuntimeError("param_amd: quantised collectives run on ROCm tensors only (HIP row quantisers, no CPU fallback); "
                               "a host-tensor run needs MI355XBackend.host_row_codec to be injected (tests/dist_workers.py)")
        return self.host_row_codec(bits)

    def _quantize_rows(self, t, dim, bits, out=None):
        """fp32 -> quantised rows, flat uint8.  GPU tensors: the HIP kernels (no fallback).  Host tensors (``--device cpu``,
        the gloo mode the plug-in is tested in without GPUs): the injected ``host_row_codec``, else an error."""
        if t.is_cuda:
            from ... import quant
            return quant.quantize_rows(t, dim, bits, out=out).view(-1)
        q = (t.reshape(-1).to(torch.float16).view(torch.uint8) if bits == 16
             else self._host_codec(bits)[0](t.reshape(-1, dim)).reshape(-1))
        if out is not None:
            out.view(-1).copy_(q)
            return out.view(-1)
        return q

    def _dequantize_rows(self, q, dim, bits, out):
        if q.is_cuda:
            from ... import quant
            quant.dequantize_rows(q, dim, bits, out=out)
            return out
        from ...quant import host_row_bytes
        d = (q.view(torch.float16).to(torch.float32) if bits == 16
             else self._host_codec(bits)[1](q.view(-1, host_row_bytes(dim, bits))))
        out.view(-1).copy_(d.reshape(-1))
        return out

    def _wants_quant_a2a(self, collectiveArgs, ip, op, pair, threshold: bool = True) -> bool:
        """the reference's test (pytorch_dist_backend.py:262-272): armed, float32, at or above the threshold, not pair mode.
        ``threshold=False``: the list-form all_to_all, which the reference quantises whenever the flag is set (:211); only
        float32 payloads have a row-wise format, so the dtype test stays."""
        bits = getattr(collectiveArgs, "all2all_qcomm", None)
        if not bits or int(bits) >= 32 or pair:
            return False
        ips = ip if isinstance(ip, (list, tuple)) else [ip]
        ops = op if isinstance(op, (list, tuple)) else [op]
        if any(t.dtype != torch.float32 for t in ips):
            return False
        if not threshold:
            return True
        thr = getattr(collectiveArgs, "quant_threshold", 0)
        return sum(t.numel() for t in ops) >= thr or sum(t.numel() for t in ips) >= thr

    def _quantized_all_to_all(self, collectiveArgs, ip, op, isp, osp, retFlag):
        """Row-wise quantised exchange: quantise every peer's chunk in place of the send buffer, ONE byte all-to-all,
        restore into the caller's output.  Chunks are whole rows of ``quant_a2a_embedding_dim`` values (pooled
        embeddings); blocking, as the reference requires (comms_utils.py:424-425)."""
        from ...quant import host_row_bytes
        bits, dim = int(collectiveArgs.all2all_qcomm), int(collectiveArgs.quant_a2a_embedding_dim)
        group = self._group(collectiveArgs)
        world = dist.get_world_size(group)
        in_list = list(ip) if isinstance(ip, (list, tuple)) else None
        out_list = list(op) if isinstance(op, (list, tuple)) else None
        isp = [t.numel() for t in in_list] if in_list is not None else (list(isp) if isp is not None and len(isp) else [ip.numel() // world] * world)
        osp = [t.numel() for t in out_list] if out_list is not None else (list(osp) if osp is not None and len(osp) else [op.numel() // world] * world)
        if any(n % dim for n in isp + osp):
            raise ValueError(f"quantized all_to_all: every per-peer chunk must be a whole number of rows of "
                             f"--quant-a2a-embedding-dim {dim} elements (chunks {isp} / {osp})")
        rb = host_row_bytes(dim, bits)
        qi = [n // dim * rb for n in isp]
        qo = [n // dim * rb for n in osp]
        dev = (in_list[0] if in_list is not None else ip).device
        q_in = torch.empty(sum(qi), dtype=torch.uint8, device=dev)
        q_out = torch.empty(sum(qo), dtype=torch.uint8, device=dev)

        def quantise():
            if in_list is None:
                self._quantize_rows(ip.reshape(-1)[:sum(isp)], dim, bits, out=q_in)
            else:
                o = 0
                for t, nb in zip(in_list, qi):
                    if nb:
                        self._quantize_rows(t.contiguous(), dim, bits, out=q_in[o:o + nb])
                    o += nb
        self._timed(getattr(collectiveArgs, "quant_time", None), quantise)
        dist.all_to_all_single(q_out, q_in, qo, qi, group=group)
        self.device_sync()

        def restore():
            if out_list is None:
                if op.is_contiguous():
                    self._dequantize_rows(q_out, dim, bits, op.view(-1)[:sum(osp)])
                else:   # reshape() of a non-contiguous tensor is a copy: restore into a flat buffer, then write back
                    flat = torch.empty(sum(osp), dtype=op.dtype, device=op.device)
                    self._dequantize_rows(q_out, dim, bits, flat)
                    full = op.reshape(-1)
                    full[:sum(osp)] = flat
                    op.copy_(full.view(op.shape))
            else:
                o = 0
                for t, nb in zip(out_list, qo):
                    if nb:
                        self._dequantize_rows(q_out[o:o + nb], dim, bits, t)
                    o += nb
        self._timed(getattr(collectiveArgs, "dequant_time", None), restore)
        return None

    def _downcast_reduce(self, collectiveArgs, ip, bits, issue, retFlag):
        """reference all_reduce / reduce with ``--bitwidth`` (pytorch_dist_backend.py:107-206): the collective runs on a
        downcast COPY (fp16 for 16 bits, int8 for 8: ``_downcast``, :48-54) and the restored tensor is the returned /
        waited object -- ``ipTensor`` keeps its values, the result is for benchmarking only, as the reference notes"""
        if bits == 16:
            cast = torch.float16
        elif bits == 8:
            cast = torch.int8
        else:
            raise NotImplementedError("Unsupported bitwidth. Set --bitwidth to 8/16/32")
        quantized = self._timed(getattr(collectiveArgs, "quant_time", None), lambda: ip.to(cast))
        work = issue(quantized)
        if collectiveArgs.asyncOp and work is not None:
            ret = work.get_future().then(lambda fut: fut.value()[0].to(torch.float32))
        else:
            ret = self._timed(getattr(collectiveArgs, "dequant_time", None), lambda: quantized.to(torch.float32))
        if collectiveArgs.asyncOp:
            collectiveArgs.waitObj.append(ret)
        return ret if retFlag else None

    def sayHello(self, *_ignored):
        """Where each process runs.  The reference drivers call it with no arguments (comms.py:1533, dlrm.py:1350,
        commsTraceReplay.py:1331; implementation pytorch_dist_backend.py:85-99: every rank posts its line to the
        TCP store, rank 0 prints them all); positional arguments of the ABC's declared signature
        (pytorch_backend_utils.py:273-277) are accepted and ignored -- the values are read from the backend itself."""
        myhost = os.uname()[1]
        global_rank, local_rank, world_size = self.get_global_rank(), self.get_local_rank(), self.get_world_size()
        dev = self.get_device()
        hw = f" ({torch.cuda.get_device_name(dev)})" if self._is_gpu() and torch.cuda.is_available() else ""
        msg = (f"[Rank {global_rank:3}] host {myhost}, device: {dev}{hw}, local_rank: {local_rank} "
               f"world_size: {world_size}, master_ip: {self.bootstrap_info.master_ip}")
        if self.tcp_store is None:
            print(msg)
            return
        self.store_set(f"hello_msg_{global_rank}", msg)
        if global_rank == 0:
            for rank in range(world_size):
                print(f"Hello from Rank {rank}: {self.store_get(f'hello_msg_{rank}').decode()}")

    # ------------------------------------------------------------------ collectives
    def all_reduce(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, _ = self._io(collectiveArgs, pair, pairIdx)
        bits = getattr(collectiveArgs, "allreduce_qcomm", 32)
        if bits != 32 and bits > 4 and ip.dtype == torch.float32 and not pair:      # the reference's test (:109-114)
            return self._downcast_reduce(collectiveArgs, ip, bits, lambda q: dist.all_reduce(
                q, op=collectiveArgs.op or dist.ReduceOp.SUM, group=self._group(collectiveArgs),
                async_op=bool(collectiveArgs.asyncOp)), retFlag)
        work = dist.all_reduce(ip, op=collectiveArgs.op or dist.ReduceOp.SUM,
                               group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def reduce(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, _ = self._io(collectiveArgs, pair, pairIdx)
        if getattr(collectiveArgs, "reduce_qcomm", 32) != 32 and not pair:                 # (:166-177)
            assert ip.dtype == torch.float32
            return self._downcast_reduce(collectiveArgs, ip, getattr(collectiveArgs, "allreduce_qcomm", 32), lambda q: dist.reduce(
                q, dst=collectiveArgs.srcOrDst, op=collectiveArgs.op or dist.ReduceOp.SUM,
                group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp)), retFlag)
        work = dist.reduce(ip, dst=collectiveArgs.srcOrDst,
                           op=collectiveArgs.op or dist.ReduceOp.SUM, group=self._group(collectiveArgs),
                           async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_to_all(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        """List-form all_to_all (``opTensor`` / ``ipTensor`` are lists of per-peer tensors).  With batched
        embedding tables attached AND the fused path opted into (``fused_lookup_a2a``; the reference requires its
        ``extend_distributed`` package for this branch, ``pytorch_dist_backend.py:214``) it is the pipelined
        lookup -> pooled all-to-all instead."""
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        # branch order of the reference (pytorch_dist_backend.py:211-214): quantisation first -- for the list form it is
        # armed by the flag alone, no size threshold (:262-272 has one only for all_to_allv) --, then the batched lookup
        if self._wants_quant_a2a(collectiveArgs, ip, op, pair, threshold=False):
            return self._quantized_all_to_all(collectiveArgs, ip, op, None, None, retFlag)
        if (not pair and collectiveArgs.num_emb_tables_batched > 0 and collectiveArgs.emb is not None
                and (self.use_ext_dist or self.fused_lookup_a2a)):
            return self.lookup_all_to_all(collectiveArgs, retFlag)
        if isinstance(op, (list, tuple)):
            if dist.get_backend(self._group(collectiveArgs)) == "gloo":
                # gloo has no list-form alltoall (reference survey probe): flatten to the single-tensor form
                flat_in = torch.cat([t.reshape(-1) for t in ip])
                flat_out = torch.empty(sum(t.numel() for t in op), dtype=flat_in.dtype, device=flat_in.device)
                dist.all_to_all_single(flat_out, flat_in, [t.numel() for t in op], [t.numel() for t in ip],
                                       group=self._group(collectiveArgs))
                o = 0
                for t in op:
                    t.copy_(flat_out[o:o + t.numel()].view_as(t))
                    o += t.numel()
                work = None
            else:
                work = dist.all_to_all(op, ip, group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        else:
            work = dist.all_to_all_single(op, ip, group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_to_allv(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        if pair:
            osp, isp = collectiveArgs.opTensor_split_pair[pairIdx], collectiveArgs.ipTensor_split_pair[pairIdx]
        else:
            osp, isp = collectiveArgs.opTensor_split, collectiveArgs.ipTensor_split
        if self._wants_quant_a2a(collectiveArgs, ip, op, pair):
            return self._quantized_all_to_all(collectiveArgs, ip, op, isp, osp, retFlag)
        work = dist.all_to_all_single(
            op, ip, list(osp) if osp is not None and len(osp) else None, list(isp) if isp is not None and len(isp) else None,
            group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_to_all_single(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        if getattr(collectiveArgs, "all2all_qcomm", None) and int(collectiveArgs.all2all_qcomm) < 32:
            logger.warning("all_to_all_single does not support quantization")     # and does nothing (:325-329)
            return None
        return self.all_to_allv(collectiveArgs, retFlag, pair, pairIdx)

    # The rest of the reference ABC's collective table (pytorch_backend_utils.py:161-180).  Only the all-to-all family is
    # on the DLRM sparse path, but the reference's own drivers need these to run at all: comms.py gathers every rank's
    # latencies through ``all_gather`` after each size (comms.py:951), dlrm.py does the same for its report (:1220, :1283).
    # They are thin c10d calls, as in pytorch_dist_backend.py:359-624.
    def all_gather(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.all_gather(tensor_list=op, tensor=ip, group=self._group(collectiveArgs),
                               async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_gather_base(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.all_gather_into_tensor(output_tensor=op, input_tensor=ip, group=self._group(collectiveArgs),
                                           async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def all_gather_object(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        dist.all_gather_object(object_list=op, obj=ip, group=self._group(collectiveArgs))
        return None

    def gather(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.gather(gather_list=op if collectiveArgs.global_rank == collectiveArgs.srcOrDst else None, tensor=ip,
                           dst=collectiveArgs.srcOrDst, group=self._group(collectiveArgs),
                           async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def scatter(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.scatter(tensor=op, scatter_list=ip if collectiveArgs.global_rank == collectiveArgs.srcOrDst else None,
                            src=collectiveArgs.srcOrDst, group=self._group(collectiveArgs),
                            async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def reduce_scatter(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.reduce_scatter(output=op, input_list=ip, op=collectiveArgs.op or dist.ReduceOp.SUM,
                                   group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def reduce_scatter_base(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        ip, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.reduce_scatter_tensor(output=op, input=ip, op=collectiveArgs.op or dist.ReduceOp.SUM,
                                          group=self._group(collectiveArgs), async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def broadcast(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        _, op = self._io(collectiveArgs, pair, pairIdx)
        work = dist.broadcast(tensor=op, src=collectiveArgs.srcOrDst, group=self._group(collectiveArgs),
                              async_op=bool(collectiveArgs.asyncOp))
        return self._post(collectiveArgs, work, retFlag)

    def broadcast_object_list(self, collectiveArgs, retFlag=False, pair=False, pairIdx=0):
        _, op = self._io(collectiveArgs, pair, pairIdx)
        dist.broadcast_object_list(object_list=op, src=collectiveArgs.srcOrDst, group=self._group(collectiveArgs))
        return None

    def incast(self, collectiveArgs):
        """many-to-one: the root posts one irecv per source rank, the sources isend (pytorch_dist_backend.py:554-573)"""
        if collectiveArgs.global_rank == collectiveArgs.srcOrDst:
            for idx, src in enumerate(collectiveArgs.src_ranks):
                collectiveArgs.waitObj.append(dist.irecv(tensor=collectiveArgs.opTensor[idx], src=src,
                                                         group=self._group(collectiveArgs), tag=0))
        elif collectiveArgs.global_rank in collectiveArgs.src_ranks:
            if collectiveArgs.asyncOp:
                self._isend_to(collectiveArgs, collectiveArgs.srcOrDst)
            else:
                dist.send(tensor=collectiveArgs.ipTensor, dst=collectiveArgs.srcOrDst, group=self._group(collectiveArgs), tag=0)

    def multicast(self, collectiveArgs):
        """one-to-many: the root isends to every destination rank, which recv (pytorch_dist_backend.py:611-624)"""
        if collectiveArgs.global_rank == collectiveArgs.srcOrDst:
            for dst in collectiveArgs.dst_ranks:
                self._isend_to(collectiveArgs, dst)
        elif collectiveArgs.global_rank in collectiveArgs.dst_ranks:
            dist.recv(tensor=collectiveArgs.opTensor, src=collectiveArgs.srcOrDst, group=self._group(collectiveArgs), tag=0)

    # point to point (comms.py:573-700 pt2pt benchmarks, trace replay)
    def send(self, collectiveArgs, retFlag=False, tag=0):
        dist.send(tensor=collectiveArgs.ipTensor, dst=collectiveArgs.dst_rank, group=self._group(collectiveArgs), tag=tag)

    def recv(self, collectiveArgs, retFlag=False, tag=0):
        dist.recv(tensor=collectiveArgs.opTensor, src=collectiveArgs.src_rank, group=self._group(collectiveArgs), tag=tag)

    def _isend_to(self, collectiveArgs, dst, tag=0):
        work = dist.isend(tensor=collectiveArgs.ipTensor, dst=dst, group=self._group(collectiveArgs), tag=tag)
        collectiveArgs.waitObj.append(work)
        return work

    def isend(self, collectiveArgs, retFlag=False, tag=0):
        work = self._isend_to(collectiveArgs, collectiveArgs.dst_rank, tag)
        return work if retFlag else None

    def irecv(self, collectiveArgs, retFlag=False, tag=0):
        work = dist.irecv(tensor=collectiveArgs.opTensor, src=collectiveArgs.src_rank, group=self._group(collectiveArgs), tag=tag)
        collectiveArgs.waitObj.append(work)
        return work if retFlag else None

    def P2POp(self, collectiveArgs, retFlag=False, tag=0):
        if collectiveArgs.collective in ("send", "isend"):
            op, tensor, peer = dist.isend, collectiveArgs.ipTensor, collectiveArgs.dst_rank
        elif collectiveArgs.collective in ("recv", "irecv"):
            op, tensor, peer = dist.irecv, collectiveArgs.opTensor, collectiveArgs.src_rank
        else:
            raise RuntimeError(f"Unknown operation type {collectiveArgs.collective}")
        req = dist.P2POp(op=op, tensor=tensor, peer=peer, group=self._group(collectiveArgs), tag=tag)
        collectiveArgs.p2pOps.append(req)
        return req if retFlag else None

    def batch_isend_irecv(self, collectiveArgs, retFlag=False):
        if not collectiveArgs.p2pOps:
            return
        reqs = dist.batch_isend_irecv(collectiveArgs.p2pOps)
        collectiveArgs.p2pOps.clear()
        collectiveArgs.waitObj.extend(reqs)

    def wait(self, collectiveArgs, retFlag=False):
        """With request ids recorded (trace replay: ``waitObjIds[req] = work``) wait on the request named by
        ``collectiveArgs.collectiveId``; otherwise on the FIRST outstanding request only (reference
        ``wait`` / ``complete_single_op``, pytorch_dist_backend.py:724-744)."""
        if collectiveArgs.waitObjIds:
            self._wait_on(collectiveArgs.waitObjIds.get(collectiveArgs.collectiveId))
            return
        if collectiveArgs.waitObj:
            self._wait_on(collectiveArgs.waitObj.pop(0))
            self.device_sync(collectiveArgs)

    complete_single_op = wait   # the reference's name for the second form (pytorch_dist_backend.py:724)

    @staticmethod
    def _wait_on(w):
        """a work handle, or the list of handles one pipelined lookup -> all-to-all call returns"""
        if w is None:
            return
        for x in (w if isinstance(w, (list, tuple)) else (w,)):
            if x is not None:
                x.wait()

    def complete_accel_ops(self, collectiveArgs, devSync=True):
        for w in collectiveArgs.waitObj:
            self._wait_on(w)
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
        forward = one HIP launch per op over all of its tables; backward = the fused backward + exact row-wise
        Adagrad step the reference's ``LookupOut.backward`` runs through fbgemm (its TBE ops are built with
        ``OptimType.EXACT_ROWWISE_ADAGRAD``, comms_utils.py:2014).  Skipped -- as in the reference -- when the
        lookup is pooled into the all-to-all instead (``all_to_all`` with the fused path on)."""
        if (collectiveArgs.collective == "all_to_all" and collectiveArgs.num_emb_tables_batched != -1
                and (self.use_ext_dist or self.fused_lookup_a2a)):
            return
        if collectiveArgs.direction == "forward":
            for i, (indices, offsets, weights) in enumerate(collectiveArgs.embRequests):
                collectiveArgs.LookupOut = collectiveArgs.emb[i].forward(indices, offsets, weights)
        else:
            for i, (indices, offsets, weights) in enumerate(collectiveArgs.embRequests):
                collectiveArgs.emb[i].optimizer_step_(collectiveArgs.grad_output, indices, offsets,
                                                      per_sample_weights=weights)

    def lookup_all_to_all(self, collectiveArgs, retFlag=False):
        """Pipelined lookup -> pooled all-to-all (reference ``:214-234``): op i's pooled embeddings
        ``[global_batch, tables_i * D]`` leave through RCCL while op i+1 is being looked up.  Blocking mode waits
        for every exchange (as the reference does); with ``asyncOp`` the individual work handles join
        ``collectiveArgs.waitObj``.  Returns the list of handles iff ``retFlag``."""
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
        if collectiveArgs.asyncOp:
            collectiveArgs.waitObj.extend(works)
        else:
            for w in works:
                w.wait()
        return works if retFlag else None

    # ------------------------------------------------------------------ memory
    def get_mem_size(self, collectiveArgs, pair=False, pairIdx=0) -> int:
        """algBW numerator (reference ``:860-897``): bytes of the OUTPUT tensor(s); of the input list for reduce_scatter
        and of the input tensor for reduce_scatter_base / reduce_scatter_v; pair mode: of the pair's output"""
        def nbytes(x):
            if isinstance(x, (list, tuple)):
                return sum(t.nelement() * t.element_size() for t in x)
            return x.nelement() * x.element_size()

        if pair:
            return nbytes(collectiveArgs.opTensor_pair[pairIdx])
        ip, op = collectiveArgs.ipTensor, collectiveArgs.opTensor
        if isinstance(op, (list, tuple)):
            return nbytes(op)
        if isinstance(ip, (list, tuple)) or collectiveArgs.collective in ("reduce_scatter_v", "reduce_scatter_base"):
            return nbytes(ip)
        return nbytes(op)

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

        # optimizer: what the reference builds its TBE ops with (comms_utils.py:2014)
        return BatchedEmbeddingBagMI355(rows, dim, dtype=dtype, device=curRankDevice, layout=layout,
                                        init="uniform_dlrm", seed=int(np.random.randint(0, 2**31 - 1)),
                                        optimizer="rowwise_adagrad")

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

    def get_next_group(self):
        return next(self.round_robin_group) if self.round_robin_group is not None else self.get_default_group()

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

    def _pg_backend(self, backend: str) -> str:
        """c10d process-group backend behind ``--backend rccl_xgmi``: RCCL ("nccl" IS RCCL on ROCm) for device
        tensors; gloo when the driver was started with ``--device cpu`` (host tensors cannot travel over RCCL: this
        is how the plug-in runs under the reference's unmodified drivers on a box without GPUs -- the committed
        fixture tests/golden/ref_plugin_rows.json was made that way).  ``PARAM_AMD_PG_BACKEND`` overrides."""
        forced = os.environ.get("PARAM_AMD_PG_BACKEND")
        if forced:
            return forced
        if backend == BACKEND_NAME or backend in ("rccl", "rocm"):
            return "nccl" if self._is_gpu() else "gloo"
        return backend

    def initialize_backend(self, master_ip, master_port, backend="nccl", eager_mode=False) -> None:
        """TCPStore rendezvous + init_process_group, as the reference (``:1145-1200``); RCCL needs
        HSA_ENABLE_IPC_MODE_LEGACY=0 for dmabuf IPC (set if absent)."""
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.set_device(self.bootstrap_info.local_rank, self.bootstrap_info.global_rank)
        rank, world = self.bootstrap_info.global_rank, self.bootstrap_info.world_size
        backend = self._pg_backend(backend)
        if not dist.is_initialized():
            self.initialize_tcpstore(master_ip, master_port)
            init_method = self._cp("init_method", None)          # --init-method URL replaces the store (reference :1182-1183)
            dist.init_process_group(backend, rank=rank, world_size=world, store=self.tcp_store if init_method is None else None,
                                    init_method=init_method,
                                    device_id=self.get_device() if (eager_mode and self._is_gpu()) else None)
        self.groups = {0: self.get_default_group()}
        self.num_pgs = 1
        self.round_robin_group = cycle(list(self.groups.values()))     # reference :1200

    def initialize_groups(self, groupRanks=None, backend="nccl", force_new_group=False):
        groups = {}
        if groupRanks is None:          # the reference's driver leaves the table on the object first (comms.py:1455-1456)
            groupRanks = getattr(self, "groupRanks", None)
        for pg_id, ranks in (groupRanks or {}).items():
            if len(ranks) == self.get_world_size() and not force_new_group:
                groups[pg_id] = self.get_default_group()
            else:
                groups[pg_id] = dist.new_group(ranks=ranks, backend=self._pg_backend(backend))
        if groups:
            self.groups = groups
        self.num_pgs = len(self.groups)
        self.round_robin_group = cycle(list(self.groups.values()))     # reference :1251: get_next_group() walks the groups

    def get_new_pg(self, group_ranks, backend="nccl", pg_desc=""):
        """a new process group over ``group_ranks`` (every rank of the job calls it; reference ``:1136-1138``)"""
        return dist.new_group(ranks=group_ranks, backend=self._pg_backend(backend))

    def initialize_tcpstore(self, master_ip, master_port) -> None:
        """the rendezvous store by itself (reference ``:1145-1154``): ``store_set`` / ``store_get`` work before any group exists"""
        if self.tcp_store is None:
            # under torch.distributed.run the launcher's agent already serves a store on MASTER_PORT
            # (TORCHELASTIC_USE_AGENT_STORE=True): a driver started without --master-port of its own (its default IS MASTER_PORT)
            # joins that store as a client instead of failing to bind the port -- rank 0 included
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True" and str(master_port) == os.environ.get("MASTER_PORT")
            self.tcp_store = dist.TCPStore(master_ip, int(master_port), self.bootstrap_info.world_size,
                                           is_master=(self.bootstrap_info.global_rank == 0 and not agent), use_libuv=True)

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
