"""dlrm.py -- DLRM sparse-feature communication benchmark on the MI355X path.

Own restatement of the hot path of reference ``train/comms/pt/dlrm.py`` (per-iteration flow
``benchTime`` ``:1200-1323``):

  SparseFeatures            offsets -> lengths on the host, H2D            (:226-277)
  SparseDataDist            all_to_all(lengths) -> all_to_all(indices)     (:744-855)
  splitPerTable             regroup [rank][table][batch] -> per table      (:430-504)
  apply_emb                 local tables x GLOBAL batch                     (:363-388)
  alltoallv (fwd / bwd)     pooled embeddings <-> batch-parallel layout     (:86-218, :858-878)
  MLP all_reduces           weight-shaped random tensors                    (:1266-1282, :1303-1317)

What is different, on purpose (MI355X-first):
  * apply_emb is ONE batched HIP launch over all local tables that writes the all-to-all send
    layout ``[N_global, sum E_local]`` directly (the reference loops over tables in Python,
    ``torch.stack``s and then ``torch.cat``s again, ``:371-387,97``);
  * the regroup after the index exchange is a single device-side gather built from block prefix
    sums (the reference does O(world x tables) Python slicing/``cat`` with ``.item()``/``.numpy()``
    host syncs per batch, ``:448-504,801-811``); only the split sizes an all_to_all needs on the
    host are read back;
  * backward applies the fused in-place update through the sorted (deterministic) scatter-add
    instead of building a sparse COO gradient that nothing consumes (``:1296``);
  * the CLI registers ``--use-device-time`` (reference bug R1 crashes without it).
The timed regions keep the reference's names (``initTimers`` ``:961-1009``); ``--print-comms``
writes the same per-rank JSON records (``:1393-1402``).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import comms_utils
from .mi355_backend import BACKEND_NAME, MI355XBackend
from .pytorch_backend_utils import collectiveArgsHolder

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------------
# pure helpers (pinned by tests/golden/comms_pure.json against the reference)
def get_split_lengths_by_len(n: int, global_rank: int, world_size: int) -> Tuple[int, List[int]]:
    """contiguous table partition: first ``n mod W`` ranks get one more (dlrm.py:390-398)"""
    k, m = divmod(n, world_size)
    splits = [k + 1 if i < m else k for i in range(world_size)]
    return (k if m == 0 else splits[global_rank]), splits


def get_slice_sparse(global_rank: int, num_emb_per_rank: Sequence[int], world_size: int) -> slice:
    start = sum(num_emb_per_rank[:global_rank])
    return slice(start, start + num_emb_per_rank[global_rank], 1)


def lengthsToOffsets(lengths: torch.Tensor, curDevice=None) -> torch.Tensor:
    """exclusive prefix sum, one entry per bag (no trailing end; dlrm.py:245-251)"""
    out = torch.zeros_like(lengths)
    if lengths.numel() > 1:
        out[1:] = torch.cumsum(lengths[:-1], dim=0)
    return out


def calculateLengths(feature_count: int, offsets: Sequence[torch.Tensor], indices: Sequence[torch.Tensor]):
    """per-feature offsets -> lengths, features concatenated (dlrm.py:226-242)"""
    lens = []
    for f in range(feature_count):
        o, n = offsets[f], len(indices[f])
        if len(o) > 0:
            ln = torch.empty_like(o)
            ln[:-1] = o[1:] - o[:-1]
            ln[-1] = n - o[-1]
        else:
            ln = o
        lens.append(ln)
    return torch.cat(lens), torch.cat(list(indices))


def regroup_per_table(lengths: torch.Tensor, indices: torch.Tensor, batch_size: int, num_my_features: int,
                      world_size: int):
    """Received layout: lengths ``[rank][table][batch]``, indices concatenated block by block in the
    same (rank, table) order.  Returns the TBE request for the batched kernel:
    ``indices_tbe`` (table-major, within a table rank-major = global sample order) and
    ``offsets_tbe`` ``[F * world * batch + 1]``.  One gather, no per-block Python loop, no host sync.
    Same regrouping as the reference's splitPerTable (dlrm.py:430-504)."""
    W, F, B = world_size, num_my_features, batch_size
    dev = lengths.device
    if lengths.is_cuda:
        # product path: two HIP launches (block sums; scan + gather), no host sync -- pm_dlrm_regroup
        from ... import _lib

        lengths = lengths.to(torch.int64).contiguous()
        indices = indices.to(torch.int64).contiguous()
        out_idx = torch.empty_like(indices)
        out_off = torch.empty(F * W * B + 1, dtype=torch.int64, device=dev)
        scratch = torch.empty(W * F, dtype=torch.int64, device=dev)
        _lib.check(_lib.load().pm_dlrm_regroup(lengths.data_ptr(), indices.data_ptr(), W, F, B, out_idx.data_ptr(),
                                               out_off.data_ptr(), scratch.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
        return out_idx, out_off
    # host tensors = the driver's `--device cpu` mode (the reference has the same mode; here it exists for the world-2
    # gloo tests of the plumbing).  Not a fallback: device tensors never reach this branch, and the branch above raises if
    # libparam_amd.so is missing.  Same regrouping, written with torch index ops.
    l3 = lengths.view(W, F, B).to(torch.int64)
    block = l3.sum(dim=2)                                  # [W, F] indices per (rank, table) block
    src_start = torch.cumsum(block.reshape(-1), 0) - block.reshape(-1)       # received order (r, f)
    blk_t = block.t().contiguous().reshape(-1)             # destination order (f, r)
    src_start_t = src_start.view(W, F).t().contiguous().reshape(-1)
    dst_start_t = torch.cumsum(blk_t, 0) - blk_t
    n = indices.numel()
    shift = torch.repeat_interleave(src_start_t - dst_start_t, blk_t, output_size=n)
    gather = torch.arange(n, device=dev) + shift
    indices_tbe = indices[gather]
    lens_tbe = l3.permute(1, 0, 2).reshape(-1)             # [F, W, B] flattened
    offsets_tbe = torch.zeros(F * W * B + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens_tbe, 0, out=offsets_tbe[1:])
    return indices_tbe, offsets_tbe


def splitPerTable(lengths, indices, batch_size, num_my_features, world_size, global_rank=0, curDevice=None):
    """Reference-shaped result (list of per-table offsets without the trailing end, list of per-table
    index tensors) derived from :func:`regroup_per_table`."""
    idx, off = regroup_per_table(lengths, indices, batch_size, num_my_features, world_size)
    n = world_size * batch_size
    offsets, inds = [], []
    for f in range(num_my_features):
        s, e = int(off[f * n]), int(off[(f + 1) * n])
        offsets.append(off[f * n:(f + 1) * n] - s)
        inds.append(idx[s:e])
    return offsets, inds


def generate_sparse_batch(rows: Sequence[int], n: int, num_indices_per_lookup: int, fixed: bool, device,
                          generator: Optional[torch.Generator] = None):
    """Synthetic sparse input of one rank, all tables, feature-major: ``lengths [T*n]`` and the
    concatenated indices.  Same distribution as the reference's generate_uniform_input_batch
    (dlrm_data.py:151-194: bag size fixed, or round(max(1, r * min(size, L))); indices
    round(r * (size-1)), sorted and de-duplicated per bag) but vectorised."""
    lens_all, idx_all = [], []
    L = num_indices_per_lookup
    for size in rows:
        if fixed:
            want = torch.full((n,), L, dtype=torch.int64, device=device)
        else:
            r = torch.rand(n, device=device, generator=generator, dtype=torch.float64)
            want = torch.clamp(torch.round(r * min(size, L)), min=1).to(torch.int64)
        cand = torch.round(torch.rand((n, L), device=device, generator=generator, dtype=torch.float64) * (size - 1)).to(torch.int64)
        live = torch.arange(L, device=device).unsqueeze(0) < want.unsqueeze(1)
        cand = torch.where(live, cand, torch.full_like(cand, size))      # dead slots sort last
        cand, _ = torch.sort(cand, dim=1)
        keep = cand < size
        keep[:, 1:] &= cand[:, 1:] != cand[:, :-1]                         # np.unique: sorted, duplicates dropped
        lens_all.append(keep.sum(dim=1))
        idx_all.append(cand[keep])
    return torch.cat(lens_all), torch.cat(idx_all)


def generate_sparse_batch_reference(m_den: int, rows: Sequence[int], n: int, num_indices_per_lookup: int, fixed: bool,
                                    device, num_targets: int = 1):
    """The reference's ``--data-generation random`` batch, draw for draw, from NumPy's GLOBAL generator
    (dlrm_data.py:151-194 + 140-148): one ``rand(1, m_den)`` for the dense row, then per table and per sample
    [``random(1)`` for the bag size unless fixed,] ``random(size)`` for the ids -> ``unique(round(r * (rows-1)))``,
    and finally ``rand(n, num_targets)`` for the click targets.  The caller seeds NumPy with the rank before the
    first batch (the reference overwrites ``--numpy-rand-seed`` with the rank, dlrm.py:629, and re-seeds on access
    to batch 0).  Host-side and O(T * n) Python: for parity runs and fixtures (tests/golden/dlrm_np2), not for
    the 26 x 8192 benchmark shape, where ``generate_sparse_batch`` draws the same distribution on the device."""
    ra = np.random
    ra.rand(1, m_den)
    lens, idx = [], []
    for size in rows:
        for _ in range(n):
            if fixed:
                want = int(num_indices_per_lookup)
            else:
                want = int(np.round(max(1.0, ra.random(1)[0] * min(size, num_indices_per_lookup))))
            group = np.unique(np.round(ra.random(want) * (size - 1)).astype(np.int64))
            lens.append(group.size)
            idx.append(group)
    ra.rand(n, num_targets)
    return (torch.tensor(lens, dtype=torch.int64, device=device),
            torch.from_numpy(np.concatenate(idx) if idx else np.zeros(0, np.int64)).to(device))


# ------------------------------------------------------------------------------------------------
class DLRMSparsePath:
    """One rank's sparse-feature path: redistribute inputs, look up, exchange pooled embeddings,
    exchange gradients back, update.  ``lookup(indices, offsets, out)`` and
    ``update(grad, indices, offsets)`` are injected: the product passes the HIP kernels."""

    def __init__(self, backend, collectiveArgs, n_emb_per_rank: Sequence[int], emb_dim: int, local_batch: int,
                 lookup: Callable, update: Optional[Callable] = None):
        self.bf, self.ca = backend, collectiveArgs
        self.world, self.rank = collectiveArgs.world_size, collectiveArgs.global_rank
        self.n_emb_per_rank = list(n_emb_per_rank)
        self.my_features = self.n_emb_per_rank[self.rank]
        self.D, self.B = emb_dim, local_batch
        self.N = self.world * local_batch
        self.lookup, self.update = lookup, update
        self.commDetails: List[dict] = []
        self.dims_sum_per_rank = [k * emb_dim for k in self.n_emb_per_rank]
        self.timers = collectiveArgs.timers

    def _a2a(self, out, inp, out_split, in_split, async_op=False):
        ca = self.ca
        ca.opTensor, ca.ipTensor, ca.opTensor_split, ca.ipTensor_split, ca.asyncOp = out, inp, out_split, in_split, async_op
        return self.bf.all_to_allv(ca, retFlag=True)

    def sparse_data_dist(self, lengths: torch.Tensor, indices: torch.Tensor):
        """lengths ``[T_total * B]`` feature-major (this rank's local batch, ALL tables) ->
        TBE request of this rank's tables for the GLOBAL batch (dlrm.py:744-855)."""
        W, B, dev = self.world, self.B, lengths.device
        T_total = sum(self.n_emb_per_rank)
        out_lengths = torch.empty(self.my_features * B * W, dtype=lengths.dtype, device=dev)
        out_splits = [self.my_features * B] * W
        in_splits = [k * B for k in self.n_emb_per_rank]
        self.bf.sync_barrier(self.ca)
        self.timers["offset_xchg_start"] = time.monotonic()
        self._a2a(out_lengths, lengths, out_splits, in_splits)
        self.bf.complete_accel_ops(self.ca)
        self.timers["offset_xchg_end"] = time.monotonic()
        self.commDetails.append({"comms": "all_to_all", "msg_size": out_lengths.numel() * out_lengths.element_size(),
                                 "in_split": in_splits, "out_split": out_splits, "dtype": str(lengths.dtype)})
        # the index exchange needs its split sizes on the host: ONE small D2H of 2*W numbers
        per_feature = lengths.view(T_total, B).sum(dim=1)
        bounds = np.cumsum([0] + self.n_emb_per_rank)
        send_per_rank = torch.stack([per_feature[bounds[j]:bounds[j + 1]].sum() for j in range(W)])
        recv_per_rank = out_lengths.view(W, -1).sum(dim=1)
        sizes = torch.stack([send_per_rank, recv_per_rank]).cpu().numpy()
        in_idx_splits, out_idx_splits = sizes[0].tolist(), sizes[1].tolist()
        out_indices = torch.empty(int(sum(out_idx_splits)), dtype=torch.int64, device=dev)
        self.bf.sync_barrier(self.ca)
        self.timers["idx_xchg_start"] = time.monotonic()
        self._a2a(out_indices, indices, out_idx_splits, in_idx_splits)
        self.bf.complete_accel_ops(self.ca)
        self.timers["idx_xchg_end"] = time.monotonic()
        self.commDetails.append({"comms": "all_to_all", "msg_size": out_indices.numel() * out_indices.element_size(),
                                 "in_split": [int(x) for x in in_idx_splits], "out_split": [int(x) for x in out_idx_splits],
                                 "dtype": str(indices.dtype)})
        return regroup_per_table(out_lengths, out_indices, B, self.my_features, W)

    def apply_emb(self, indices_tbe, offsets_tbe, out=None):
        """ONE batched launch -> ``[N_global, my_features * D]`` (the all-to-all send layout)."""
        if out is None:
            out = torch.empty((self.N, self.my_features * self.D), dtype=torch.float32, device=indices_tbe.device)
        self.lookup(indices_tbe, offsets_tbe, out)
        return out

    def alltoallv_fwd(self, ly: torch.Tensor):
        """pooled embeddings -> batch-parallel: returns ``[B, sum_j E_j]`` assembled from the per-source
        ``[B, E_j]`` blocks (what All2Allv_Wait.forward + torch.cat(dim=1) produce, dlrm.py:173-175,1253)."""
        E_loc = self.my_features * self.D
        in_split = [self.B * E_loc] * self.world
        out_split = [self.B * e for e in self.dims_sum_per_rank]
        recv = torch.empty(sum(out_split), dtype=ly.dtype, device=ly.device)
        self.bf.sync_barrier(self.ca)
        self.timers["fwd_a2a_start"] = time.monotonic()
        req = self._a2a(recv, ly.view(-1), out_split, in_split, async_op=True)
        if req is not None:
            req.wait()
        self.bf.complete_accel_ops(self.ca)
        self.timers["fwd_a2a_end"] = time.monotonic()
        self.commDetails.append({"comms": "all_to_all", "msg_size": ly.numel() * ly.element_size(),
                                 "in_split": in_split, "out_split": out_split, "dtype": str(ly.dtype)})
        blocks = [b.view(self.B, -1) for b in recv.split(out_split)]
        return torch.cat(blocks, dim=1), out_split, in_split

    def alltoallv_bwd(self, grad_full: torch.Tensor, out_split, in_split):
        """gradient of the concatenated pooled embeddings ``[B, sum_j E_j]`` -> ``[N_global, E_local]``"""
        gblocks = [g.contiguous().view(-1) for g in grad_full.split(self.dims_sum_per_rank, dim=1)]
        send = torch.cat(gblocks)
        recv = torch.empty(self.N * self.my_features * self.D, dtype=grad_full.dtype, device=grad_full.device)
        self.bf.sync_barrier(self.ca)
        self.timers["bwd_a2a_start"] = time.monotonic()
        req = self._a2a(recv, send, in_split, out_split, async_op=True)
        if req is not None:
            req.wait()
        self.bf.complete_accel_ops(self.ca)
        self.timers["bwd_a2a_end"] = time.monotonic()
        self.commDetails.append({"comms": "all_to_all", "msg_size": recv.numel() * recv.element_size(),
                                 "in_split": out_split, "out_split": in_split, "dtype": str(grad_full.dtype)})
        return recv.view(self.N, self.my_features * self.D)


# ------------------------------------------------------------------------------------------------
REGIONS = [  # (name, start timer, end timer): reference initTimers, dlrm.py:961-1009
    ("intermed_calc_length", "iter_start", "length_calc_end"), ("mem_push_idx", "length_calc_end", "mem_push_idx_end"),
    ("intermed_bef_offset_xchg", "mem_push_idx_end", "offset_xchg_start"), ("offset_xchg", "offset_xchg_start", "offset_xchg_end"),
    ("intermed_btw_offset_idx_xchg", "offset_xchg_end", "idx_xchg_start"), ("idx_xchg", "idx_xchg_start", "idx_xchg_end"),
    ("intermed_post_idx_xchg_sparse_dist", "idx_xchg_end", "bef_emb_lookup"),
    ("intermed_emb_lookup_to_a2a_start", "bef_emb_lookup", "fwd_a2a_start"), ("fwd_a2a", "fwd_a2a_start", "fwd_a2a_end"),
    ("intermed_fwd_a2a_grad_push", "fwd_a2a_end", "grad_push_start"), ("mem_push_gradients", "grad_push_start", "bwd_top_ar_start"),
    ("bwd_top_ar", "bwd_top_ar_start", "bwd_top_ar_end"), ("intermed_top_ar_end_to_bwd_a2a_start", "bwd_top_ar_end", "bwd_a2a_start"),
    ("bwd_a2a", "bwd_a2a_start", "bwd_a2a_end"), ("intermed_bwd_a2a_bot_ar", "bwd_a2a_end", "bwd_bot_ar_start"),
    ("bwd_bot_ar", "bwd_bot_ar_start", "bwd_bot_ar_end"), ("iter_time", "iter_start", "bwd_bot_ar_end"),
    ("iter_data_prep", "iter_start", "bef_emb_lookup"), ("iter_fwd_a2a", "iter_start", "grad_push_start"),
    ("iter_bwd_top_ar", "iter_start", "bwd_top_ar_end"), ("iter_bwd_a2a", "iter_start", "bwd_bot_ar_start"),
]


class commsDLRMBench:
    def __init__(self):
        self.collectiveArgs = collectiveArgsHolder()
        self.measured = {name: [] for name, _, _ in REGIONS}
        self.memory = {name: [] for name, _, _ in REGIONS}

    def readArgs(self, parser, argv=None):
        parser.add_argument("--master-ip", type=str, default="127.0.0.1")
        parser.add_argument("--master-port", type=str, default="29500")
        parser.add_argument("--backend", type=str, default=BACKEND_NAME)
        parser.add_argument("--device", type=str, default="rocm", choices=["cuda", "rocm", "cpu"])
        parser.add_argument("--arch-sparse-feature-size", type=int, default=4)
        parser.add_argument("--arch-embedding-size", type=str, default="4-3-2")
        parser.add_argument("--arch-mlp-bot", type=str, default="4-3-2")
        parser.add_argument("--arch-mlp-top", type=str, default="4-2-1")
        parser.add_argument("--mini-batch-size", type=int, default=1)
        parser.add_argument("--num-batches", type=int, default=10)
        parser.add_argument("--warmup-batches", type=int, default=2)
        parser.add_argument("--num-indices-per-lookup", type=int, default=10)
        # the reference declares this one ``type=bool`` (dlrm.py:705): it takes a value and any non-empty string means True;
        # here the value is optional, so both ``--num-indices-per-lookup-fixed`` and ``--num-indices-per-lookup-fixed True`` parse
        parser.add_argument("--num-indices-per-lookup-fixed", nargs="?", const=True, default=False, type=bool)
        parser.add_argument("--numpy-rand-seed", type=int, default=123)
        parser.add_argument("--embed-data-type", type=str, default="float32", choices=["float32", "bfloat16", "float16"])
        parser.add_argument("--learning-rate", type=float, default=0.01)
        parser.add_argument("--print-comms", action="store_true")
        parser.add_argument("--data-generation", type=str, default="device", choices=["device", "random"],
                            help="device: vectorised generator on the GPU; random: the reference's NumPy sequence, "
                                 "seeded with the rank (dlrm.py:629) -- identical batches, host-side")
        parser.add_argument("--use-device-time", action="store_true", default=False)  # reference bug R1: never registered there
        parser.add_argument("--log", type=str, default="ERROR")
        # sizing of the top MLP's first layer, i.e. of the first dense all_reduce (dlrm.py:583-601)
        parser.add_argument("--arch-interaction-op", type=str, default="dot", help="dot | cat")
        parser.add_argument("--arch-interaction-itself", action="store_true", default=False)
        parser.add_argument("--arch-project-size", type=int, default=0, help="project size for the interaction features")
        parser.add_argument("--model", type=str, default="dlrm", help="Model to be benchmarked (dlrm)")
        parser.add_argument("--perf-debug", action="store_true", help="extra barriers between the regions of the backward (dlrm.py:1261-1299)")
        # parsed by the reference for the DLRM data loader it borrows, without effect on the comms benchmark's batches
        # (dlrm_data.py reads only --data-generation / --data-size / --round-targets): accepted so that a reference
        # command line runs unchanged
        parser.add_argument("--num-workers", type=int, default=0)
        parser.add_argument("--data-size", type=int, default=1)
        parser.add_argument("--synthetic-data-folder", type=str, default="./synthetic_data/syn_data_bs65536/")
        parser.add_argument("--round-targets", type=bool, default=False)
        parser.add_argument("--rand-data-dist", type=str, default="uniform")
        parser.add_argument("--rand-data-min", type=float, default=0)
        parser.add_argument("--rand-data-max", type=float, default=1)
        parser.add_argument("--rand-data-mu", type=float, default=-1)
        parser.add_argument("--rand-data-sigma", type=float, default=1)
        parser.add_argument("--data-trace-file", type=str, default="./input/dist_emb_j.log")
        parser.add_argument("--data-trace-enable-padding", type=bool, default=False)
        return parser.parse_args(argv)

    @staticmethod
    def top_mlp_input_size(args, n_tables: int) -> int:
        """width of the interaction output that feeds the top MLP (dlrm.py:575-601): (tables + 1) features; ``dot`` = unique
        pairs (with the diagonal under --arch-interaction-itself) + the bottom MLP's output, ``cat`` = every feature at the
        bottom MLP's output width; --arch-project-size replaces either by features x project size + bottom output"""
        num_fea = n_tables + 1
        m_den_out = int(args.arch_mlp_bot.split("-")[-1])
        if args.arch_interaction_op == "dot":
            num_int = (num_fea * (num_fea + 1 if args.arch_interaction_itself else num_fea - 1)) // 2 + m_den_out
        elif args.arch_interaction_op == "cat":
            num_int = num_fea * m_den_out
        else:
            sys.exit("ERROR: --arch-interaction-op=" + args.arch_interaction_op + " is not supported")
        if args.arch_project_size > 0:
            num_int = num_fea * args.arch_project_size + m_den_out
        return num_int

    def run(self, args, lookup_factory=None):
        """``lookup_factory(rows, dim, device, dtype) -> (lookup, update)`` overrides the HIP kernels
        (tests only; the product path builds BatchedEmbeddingBagMI355)."""
        env = comms_utils.read_comms_env_vars()
        if env["world_size"] < 1:
            env = {"world_size": 1, "local_size": 1, "global_rank": 0, "local_rank": 0}
        if getattr(args, "model", "dlrm") != "dlrm":                       # dlrm.py:510-555
            print("Model " + args.model + " not supported...Abort!")
            sys.exit(1)
        info = comms_utils.bootstrap_info_holder(args.master_ip, args.master_port, 0, env)
        device = "cuda" if args.device == "rocm" else args.device
        bf = MI355XBackend(info, {"device": device, "backend": args.backend})
        bf.initialize_backend(args.master_ip, args.master_port,
                              backend="gloo" if device == "cpu" else ("nccl" if args.backend == BACKEND_NAME else args.backend))
        ca = self.collectiveArgs
        ca.device, ca.world_size, ca.global_rank, ca.group = bf.get_device(), bf.get_world_size(), bf.get_global_rank(), bf.get_default_group()
        W, rank, dev = ca.world_size, ca.global_rank, ca.device
        ln_emb = [int(x) for x in args.arch_embedding_size.split("-")]
        if len(ln_emb) < W:
            raise ValueError("Embedding size should match process count, please fix '--arch-embedding-size' and try again")
        _, n_emb_per_rank = get_split_lengths_by_len(len(ln_emb), rank, W)
        my_rows = ln_emb[get_slice_sparse(rank, n_emb_per_rank, W)]
        D, B = args.arch_sparse_feature_size, args.mini_batch_size
        dtype = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}[args.embed_data_type]
        if lookup_factory is None:
            from ... import BatchedEmbeddingBagMI355  # HIP kernels: raises if libparam_amd.so is missing

            np.random.seed(args.numpy_rand_seed + rank)
            emb = BatchedEmbeddingBagMI355(my_rows, D, dtype=dtype, device=dev, init="uniform_dlrm",
                                           seed=args.numpy_rand_seed + rank, learning_rate=args.learning_rate)
            n_glob = W * B
            lookup = lambda i, o, out: emb.lookup(i, o, out=out, batch=n_glob)                    # noqa: E731
            update = lambda g, i, o: emb.scatter_add_(g, i, o, alpha=-args.learning_rate, batch=n_glob)  # noqa: E731
        else:
            lookup, update = lookup_factory(my_rows, D, dev, dtype)
        path = DLRMSparsePath(bf, ca, n_emb_per_rank, D, B, lookup, update)
        # MLP "gradients": weight-shaped random tensors (dlrm.py:307-361)
        def layers(spec):
            dims = [int(x) for x in spec.split("-")]
            return [torch.rand(dims[i + 1], dims[i], device=dev) for i in range(len(dims) - 1)]
        num_int = self.top_mlp_input_size(args, len(ln_emb))
        top, bot = layers(f"{num_int}-{args.arch_mlp_top}"), layers(args.arch_mlp_bot)
        if rank == 0:          # the configuration lines of the reference's set-up (dlrm.py:545-612, 1421-1428)
            dims_sum = [n * D for n in n_emb_per_rank]
            print("\t mpi-params: %s" % (env,))
            print("\t rank: %s args.model: %s model: %s " % (rank, args.model, args.model))
            print("\tdims_sum_per_rank: %s " % (dims_sum,))
            print("\t ipConfig['num_sparse_fea']: %s " % (len(ln_emb),))
            print("\t ipConfig['n_emb_per_rank']: %s " % (list(n_emb_per_rank),))
            print("\t ipConfig['dims_sum_per_rank']: %s " % (dims_sum,))
            print("\t ipConfig['local_emb_dims']: %s " % ([D] * len(my_rows),))
            print("\t ln_top: %s \n\t ln_bot: %s \n\t n_emb: %s " % (
                np.array([num_int] + [int(x) for x in args.arch_mlp_top.split("-")]),
                np.array([int(x) for x in args.arch_mlp_bot.split("-")]), len(ln_emb)))
        print("\n\t ****** Rank: G: %d L: %d host: %s starting new epoch, model: %s ***** \n"
              % (rank, bf.get_local_rank(), os.uname()[1], args.model))
        gen = torch.Generator(device=dev)
        gen.manual_seed(args.numpy_rand_seed + rank)
        timers = ca.timers
        m_den = int(args.arch_mlp_bot.split("-")[0])
        n_records = 0
        for batch in range(args.num_batches):
            timers["iter_start"] = time.monotonic()
            if args.data_generation == "random":
                if batch == 0:
                    np.random.seed(rank)
                lengths, indices = generate_sparse_batch_reference(m_den, ln_emb, B, args.num_indices_per_lookup,
                                                                   args.num_indices_per_lookup_fixed, dev)
            else:
                lengths, indices = generate_sparse_batch(ln_emb, B, args.num_indices_per_lookup,
                                                         args.num_indices_per_lookup_fixed, dev, gen)
            timers["length_calc_end"] = timers["mem_push_idx_end"] = time.monotonic()   # generated on the device
            idx_tbe, off_tbe = path.sparse_data_dist(lengths, indices)
            bf.sync_barrier(ca)
            timers["bef_emb_lookup"] = time.monotonic()
            ly = path.apply_emb(idx_tbe, off_tbe)
            pooled, out_split, in_split = path.alltoallv_fwd(ly)
            timers["grad_push_start"] = time.monotonic()
            bf.sync_barrier(ca)
            if args.perf_debug:
                bf.sync_barrier(ca)
            timers["bwd_top_ar_start"] = time.monotonic()
            self._all_reduce_layers(bf, ca, top, path)
            timers["bwd_top_ar_end"] = time.monotonic()
            if args.perf_debug:
                bf.sync_barrier(ca)
            grad_ly = path.alltoallv_bwd(pooled, out_split, in_split)       # C = tempB (dlrm.py:1255)
            if path.update is not None:
                path.update(grad_ly, idx_tbe, off_tbe)
            if args.perf_debug:
                bf.sync_barrier(ca)
            timers["bwd_bot_ar_start"] = time.monotonic()
            self._all_reduce_layers(bf, ca, bot, path)
            timers["bwd_bot_ar_end"] = time.monotonic()
            # bytes moved per region in this batch (the report's "memory (B)" column, dlrm.py:101,193,788,834,1318): the records the
            # batch appended are, in order, lengths a2a, indices a2a, pooled a2a, top all_reduces, gradient a2a, bottom all_reduces
            recs = path.commDetails[n_records:]
            n_records = len(path.commDetails)
            a2a = [r["msg_size"] for r in recs if r["comms"] == "all_to_all"]
            k_bwd = max((k for k, r in enumerate(recs) if r["comms"] == "all_to_all"), default=0)
            mem = dict(zip(("offset_xchg", "idx_xchg", "fwd_a2a", "bwd_a2a"), a2a))
            mem["bwd_top_ar"] = sum(r["msg_size"] for r in recs[:k_bwd] if r["comms"] == "all_reduce")
            mem["bwd_bot_ar"] = sum(r["msg_size"] for r in recs[k_bwd:] if r["comms"] == "all_reduce")
            for name, _, _ in REGIONS:
                self.memory[name].append(mem.get(name, 0))
            if batch >= args.warmup_batches:
                for name, s, e in REGIONS:
                    self.measured[name].append((timers.get(e, 0.0) - timers.get(s, 0.0)) * 1e6)
        report = self.report(bf, ca, path, args)
        if args.print_comms:
            folder = os.path.join(os.getcwd(), f"dlrm_np{W}")
            os.makedirs(folder, exist_ok=True)
            with open(os.path.join(folder, f"rank{rank}.json"), "w") as f:   # every batch, like dlrm.py:1393-1402
                json.dump(path.commDetails, f)
        bf.shutdown()
        return report

    @staticmethod
    def _all_reduce_layers(bf, ca, layers, path):
        for t in layers:
            ca.ipTensor, ca.asyncOp, ca.op = t, True, bf.get_reduce_op("sum")
            bf.all_reduce(ca)
            path.commDetails.append({"comms": "all_reduce", "msg_size": t.nelement() * t.element_size(), "dtype": str(t.dtype)})
        bf.sync_barrier(ca)

    def report(self, bf, ca, path, args):
        """The reference's report (dlrm.py:1011-1198): every rank's per-iteration samples of the 21 regions are all-gathered; rank 0
        prints two tables with the columns iters / region / memory (B) / Latency(us):min / p50 / p75 / p95 / sum(p50) -- the first
        over all samples of all ranks, the second over the ranks' MEAN latencies -- each closed by a ``total_time`` row (sum of the
        p50 of the regions that are not ``iter_*`` roll-ups).  ``memory`` is the p50 over the measured iterations of the bytes the
        region moved (one deviation: ``bwd_top_ar`` / ``bwd_bot_ar`` show the all_reduces' bytes; the reference prints 0 there
        because its ``getMemSizes`` reads a stale output tensor, dlrm.py:936-950).  Returns the first table as a dict (+ the pooled
        all-to-all's bandwidth) on every rank."""
        names = [n for n, _, _ in REGIONS]
        measured = max((len(self.measured[n]) for n in names), default=0)
        out = {}
        if measured == 0:
            return out
        lat = torch.tensor([self.measured[n] for n in names], dtype=torch.float64, device=ca.device)          # [regions, iters]
        mem = torch.tensor([self.memory[n][args.warmup_batches:] for n in names], dtype=torch.float64, device=ca.device)
        all_lat = [torch.zeros_like(lat) for _ in range(ca.world_size)]
        all_mem = [torch.zeros_like(mem) for _ in range(ca.world_size)]
        dist.all_gather(all_lat, lat, group=bf.get_default_group())
        dist.all_gather(all_mem, mem, group=bf.get_default_group())
        lat_r = torch.stack(all_lat).cpu().numpy()          # [ranks, regions, iters]
        mem_r = torch.stack(all_mem).cpu().numpy()
        fmt = "\t%d\t%36s\t%12s\t%12s\t%12s\t%12s\t%12s\t%12s"
        # the "iters" column is the reference's label: (--num-batches + --warmup-batches) - --warmup-batches (comms_utils.py:845,
        # dlrm.py:1403) = --num-batches, although --num-batches - --warmup-batches batches are measured (as here)
        measured = args.num_batches
        lines_all, lines_mean = [], []
        sum_all = sum_mean = 0.0
        for k, n in enumerate(names):
            samples = lat_r[:, k, :].reshape(-1)
            means = lat_r[:, k, :].mean(axis=1)
            mem_p50 = np.percentile(mem_r[:, k, :].reshape(-1), 50) if mem_r.shape[2] else 0
            p50, p75, p95 = (float(np.percentile(samples, q)) for q in (50, 75, 95))
            m50, m75, m95 = (float(np.percentile(means, q)) for q in (50, 75, 95))
            if "iter" not in n:
                sum_all += p50
                sum_mean += m50
            out[n] = {"min": float(samples.min()), "p50": p50, "p75": p75, "p95": p95, "max": float(samples.max()),
                      "memory": float(mem_p50), "mean_p50": m50}
            lines_all.append(fmt % (measured, n, "%d" % mem_p50, "%.3f" % samples.min(), "%.3f" % p50, "%.3f" % p75, "%.3f" % p95,
                                    "%.3f" % sum_all))
            lines_mean.append(fmt % (measured, n, "%d" % mem_p50, "%.3f" % means.min(), "%.3f" % m50, "%.3f" % m75, "%.3f" % m95,
                                     "%.3f" % sum_mean))
        if ca.global_rank == 0:
            rule = "\n\n " + "-" * 125 + "\n\n"
            print("\t{}\t{:>36}\t{:>12}\t{:>12}\t{:>12}\t{:>12}\t{:>12}\t{:>12}".format(
                "iters", "region", "memory (B)", "Latency(us):min", "p50", "p75", "p95", "sum(p50)"))
            for lines, total in ((lines_all, sum_all), (lines_mean, sum_mean)):
                for ln in lines:
                    if "iter_time" in ln:
                        print("\n")
                    print(ln)
                print("\t%d\t%36s\t%12s\t%12s\t%12s" % (measured, "total_time", "N/A", "N/A", "%.3f" % total))
                print(rule)
        a2a_bytes = path.B * sum(path.dims_sum_per_rank) * 4
        if out["fwd_a2a"]["p50"] > 0:
            alg = a2a_bytes / (out["fwd_a2a"]["p50"] * 1e3)
            out["fwd_a2a_bw"] = {"bytes_per_rank": a2a_bytes, "algBW_GBps": alg,
                                 "busBW_GBps": bf.getBusBW("all_to_allv", alg, ca)}
        return out


def main(argv=None):
    bench = commsDLRMBench()
    args = bench.readArgs(argparse.ArgumentParser(description="DLRM sparse-feature comms benchmark (MI355X build)"), argv)
    comms_utils.init_logging(args.log)
    return bench.run(args)


if __name__ == "__main__":
    main()  # pragma: no cover
