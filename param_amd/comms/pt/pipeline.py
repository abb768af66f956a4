"""Table-wise sharded lookup + pooled-embedding all-to-all, pipelined per table group.

The DLRM sparse-feature exchange (reference ``train/comms/pt/dlrm.py:858-878`` C3, and its
per-op pipelining site ``pytorch_dist_backend.py:214-234`` C9): every rank owns ``T_loc`` whole
tables and looks up the GLOBAL batch (``world * B_local`` bags) for them; one all-to-all then
returns the pooled embeddings to batch-parallel layout.  Here the local tables are cut into
``groups`` table groups; group g's all-to-all is issued (asynchronously, on the process group's
own HIP stream) right after its lookup and overlaps the lookup of group g+1 on the compute stream.

Layouts (all fp32, row-major):
  send[g]  [world * B_local, Tg * D]   lookup output; rows [j*B_local, (j+1)*B_local) go to rank j
  recv[g]  [world,  B_local, Tg * D]   block i = this rank's local-batch rows of rank i's group-g tables
so ``recv[g][i, b, t*D:(t+1)*D]`` is the pooled embedding of local sample b for table
``i*T_loc + g*Tg + t`` -- the same values the reference's All2Allv_Wait views as per-source
``[lN, E_rank_i]`` blocks (dlrm.py:173-175), without the ``cat``/``contiguous`` copies.

The lookup is injected (``lookup(group, indices, offsets, out)``): bench.py passes the HIP kernel
(:func:`param_amd.embedding_bag._fwd`); tests/ pass a torch stub to check the plumbing on gloo.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def split_request_by_group(indices: torch.Tensor, offsets: torch.Tensor, tables: int, groups: int, batch: int
                           ) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Cut a TBE request (indices table-major, offsets [T*batch+1]) into ``groups`` requests of
    ``tables // groups`` consecutive tables each, offsets rebased to start at 0."""
    if tables % groups:
        raise ValueError(f"{tables} tables do not divide into {groups} groups")
    tg = tables // groups
    out = []
    for g in range(groups):
        lo, hi = g * tg * batch, (g + 1) * tg * batch
        start, end = int(offsets[lo]), int(offsets[hi])
        out.append((indices[start:end].contiguous(), (offsets[lo:hi + 1] - start).contiguous()))
    return out


class LookupAllToAll:
    def __init__(self, lookup: Callable, world: int, local_batch: int, group_dims: Sequence[int], device,
                 group=None, dtype=torch.float32, depth: int = 1):
        """``group_dims[g]`` = sum of embedding dims of table group g (``Tg * D``).  ``depth`` = number of steps in
        flight: 1 = every exchange is waited for inside its own step; 2 = the exchanges of step k complete under the
        lookups of step k+1 (double-buffered send / receive tensors) -- what a training loop does, and the only way to hide
        the exchange of the LAST table group."""
        self.lookup, self.world, self.local_batch, self.pg = lookup, world, local_batch, group
        self.groups = len(group_dims)
        self.depth = max(1, int(depth))
        n = world * local_batch
        self._send = [[torch.empty((n, d), dtype=dtype, device=device) for d in group_dims] for _ in range(self.depth)]
        self._recv = [[torch.empty((world, local_batch, d), dtype=dtype, device=device) for d in group_dims]
                      for _ in range(self.depth)]
        self._pending: List[list] = [[] for _ in range(self.depth)]
        self._k = 0
        self.send, self.recv = self._send[0], self._recv[0]

    def lookups_only(self, requests) -> None:
        for g in range(self.groups):
            self.lookup(g, requests[g][0], requests[g][1], self.send[g])

    def all_to_all_only(self) -> None:
        works = [dist.all_to_all_single(self.recv[g].view(-1, self.recv[g].shape[-1]), self.send[g],
                                        group=self.pg, async_op=True) for g in range(self.groups)]
        for w in works:
            w.wait()

    @staticmethod
    def _wait(works) -> None:
        for w in works:
            if w is not None:
                w.wait()
        works.clear()

    def step(self, requests) -> List[torch.Tensor]:
        """lookup(g) -> async all_to_all(g) for every group.  depth 1: waits for all exchanges and returns this step's
        receive tensors.  depth d: waits only for the step issued d steps ago (whose buffers are reused now) and returns
        the receive tensors of the step just issued -- valid after :meth:`flush` or after d more steps."""
        slot = self._k % self.depth
        self._wait(self._pending[slot])                      # the exchange still reading / writing this slot's tensors
        send, recv = self._send[slot], self._recv[slot]
        for g in range(self.groups):
            self.lookup(g, requests[g][0], requests[g][1], send[g])
            # equal splits: B_local rows to / from every rank.  The collective runs on the process group's stream after
            # the lookup above, under the next group's lookup (and, with depth > 1, under the next step's lookups).
            self._pending[slot].append(dist.all_to_all_single(recv[g].view(-1, recv[g].shape[-1]), send[g],
                                                              group=self.pg, async_op=True))
        if self.depth == 1:
            self._wait(self._pending[slot])
        self._k += 1
        self.send, self.recv = send, recv
        return recv

    def flush(self) -> None:
        """wait for every exchange still in flight (makes the current stream wait; no host sync)"""
        for works in self._pending:
            self._wait(works)

    def bytes_per_rank(self) -> int:
        """output-tensor bytes per rank: the reference's ``memSize`` for algBW (pytorch_dist_backend.py:860-897)"""
        return sum(r.numel() * r.element_size() for r in self.recv)


def table_split(num_tables: int, world: int) -> List[int]:
    """tables per rank, the reference's contiguous partition (dlrm.py:390-398): the first ``T mod W`` ranks hold one
    more -- 26 tables over 8 ranks -> [4, 4, 3, 3, 3, 3, 3, 3]"""
    k, m = divmod(num_tables, world)
    return [k + 1 if r < m else k for r in range(world)]


class ShardedEmbeddingExchange:
    """Training-step form of the sparse path for ANY table count (uneven shards, mixed dims): forward lookup ->
    pooled all-to-all (dlrm.py:858-878 ``fwd_a2a``), and the way back: gradient all-to-all (``bwd_a2a``, dlrm.py:204-214:
    the same exchange with send / receive splits swapped) -> fused backward on the owner.

    Rank r owns the tables ``[sum(split[:r]), sum(split[:r+1]))`` whose dims add up to ``widths[r]`` and serves the GLOBAL
    batch (``world * B_local`` bags) for them.  Layouts (fp32, contiguous):
      pooled   [world * B_local, widths[me]]            lookup output = forward send buffer; rows [j*B_local, ...) go to rank j
      recv     flat: block i = [B_local, widths[i]]     my samples' pooled embeddings of rank i's tables (forward receive)
      grad_in  flat, same blocks as recv                d(loss)/d(recv): backward send buffer, block i goes back to rank i
      grad     [world * B_local, widths[me]]            backward receive = the lookup-layout gradient the backward kernel reads
    Forward: equal send splits (``B_local * widths[me]``), receive split i = ``B_local * widths[i]``; backward: swapped.

    Three batches in flight (``step``): the compute stream runs lookup(k) and backward(k-2), the process group's stream
    runs fwd_a2a(k) and bwd_a2a(k-1) -- every collective has a whole lookup + backward to hide under, and the tables a
    lookup reads are two updates stale (pipelined-training semantics; the arithmetic of each piece is unchanged).
    ``lookup(indices, offsets, out)`` and ``backward(grad, indices, offsets)`` are injected: bench.py passes the HIP
    kernels, the gloo tests a torch stand-in.  ``make_grad(recv, grad_in)`` stands for the dense part of the model (the
    reference's benchmark has none either: it sends a gradient of the received shape back, dlrm.py:1268-1281); default:
    the received embeddings themselves."""

    def __init__(self, lookup: Callable, backward: Callable, world: int, rank: int, local_batch: int,
                 widths: Sequence[int], device, group=None, make_grad: Callable = None, slots: int = 3,
                 quant: "RowQuant | None" = None, layout: str = "bd", dim: int = 0):
        self.lookup, self.backward, self.make_grad = lookup, backward, make_grad
        self.world, self.rank, self.local_batch, self.pg = world, rank, local_batch, group
        self.widths = [int(w) for w in widths]
        assert len(self.widths) == world
        # layout of a peer's chunk inside the send / receive buffers: "bd" = [B_local, widths[r]] (the default: measured fastest at the
        # N = 8 rank shape, profiles/r05_blocked_layout_probe.jsonl) or "blocked" = [tables of r, B_local, dim] (include/param_amd.h,
        # ABI v6; one common `dim`, B_local a power of two).  Chunk sizes, splits and every collective are the same either way: the
        # injected lookup / backward write and read the buffers in the layout they were built for.
        assert layout in ("bd", "blocked")
        if layout == "blocked" and (dim <= 0 or any(w % dim for w in self.widths) or quant is not None):
            raise ValueError("blocked layout: one common embedding dim dividing every rank's width, fp32 payloads")
        self.layout, self.dim = layout, int(dim)
        wme, n = self.widths[rank], world * local_batch
        self.fwd_recv_splits = [local_batch * w for w in self.widths]
        self.fwd_send_splits = [local_batch * wme] * world
        total = sum(self.fwd_recv_splits)
        self.slots = slots
        mk = lambda *shape: [torch.empty(shape, dtype=torch.float32, device=device) for _ in range(slots)]  # noqa: E731
        self.pooled, self.recv = mk(n, wme), mk(total)
        self.grad_in, self.grad = (mk(total) if make_grad is not None else self.recv), mk(n, wme)
        self._req = [None] * slots
        self._fwd_work = [None] * slots
        self._bwd_work = [None] * slots
        self._k = 0
        # quantised exchange (the reference's --bitwidth): byte buffers beside the fp32 ones, splits in bytes
        self.quant = quant
        if quant is not None:
            if any(w % quant.dim for w in self.widths):
                raise ValueError(f"quantised exchange: every rank's width must be a whole number of {quant.dim}-element rows")
            mkq = lambda nbytes: [torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(slots)]  # noqa: E731
            fb, gb = quant.row_bytes(quant.fwd_bits), quant.row_bytes(quant.bwd_bits)
            rows = lambda elems: elems // quant.dim                                                             # noqa: E731
            if quant.fwd_bits:
                self.q_pooled, self.q_recv = mkq(rows(n * wme) * fb), mkq(rows(total) * fb)
                self.qf_send = [rows(x) * fb for x in self.fwd_send_splits]
                self.qf_recv = [rows(x) * fb for x in self.fwd_recv_splits]
            if quant.bwd_bits:
                self.q_grad_in, self.q_grad = mkq(rows(total) * gb), mkq(rows(n * wme) * gb)
                self.qb_send = [rows(x) * gb for x in self.fwd_recv_splits]
                self.qb_recv = [rows(x) * gb for x in self.fwd_send_splits]

    # -- pieces (also timed on their own by bench.py) ---------------------------------------------
    def _lookup(self, slot: int, indices, offsets) -> None:
        q = self.quant
        if q is not None and q.fwd_bits:
            if q.lookup_quantized is not None:           # the lookup kernel writes the quantised rows itself
                q.lookup_quantized(indices, offsets, self.q_pooled[slot])
            else:
                self.lookup(indices, offsets, self.pooled[slot])
                q.quantize(self.pooled[slot], q.fwd_bits, self.q_pooled[slot])
        else:
            self.lookup(indices, offsets, self.pooled[slot])

    def fwd_a2a(self, slot: int, async_op: bool = True):
        if self.quant is not None and self.quant.fwd_bits:
            return dist.all_to_all_single(self.q_recv[slot], self.q_pooled[slot], self.qf_recv, self.qf_send,
                                          group=self.pg, async_op=async_op)
        return dist.all_to_all_single(self.recv[slot], self.pooled[slot].view(-1), self.fwd_recv_splits, self.fwd_send_splits,
                                      group=self.pg, async_op=async_op)

    def _fwd_arrived(self, slot: int) -> None:
        """after the forward exchange has been waited for: restore fp32 (quantised exchange), build the gradient to send"""
        q = self.quant
        if q is not None and q.fwd_bits:
            q.dequantize(self.q_recv[slot], q.fwd_bits, self.recv[slot])
        if self.make_grad is not None:
            self.make_grad(self.recv[slot], self.grad_in[slot])
        if q is not None and q.bwd_bits:
            q.quantize(self.grad_in[slot], q.bwd_bits, self.q_grad_in[slot])

    def bwd_a2a(self, slot: int, async_op: bool = True):
        if self.quant is not None and self.quant.bwd_bits:
            return dist.all_to_all_single(self.q_grad[slot], self.q_grad_in[slot], self.qb_recv, self.qb_send,
                                          group=self.pg, async_op=async_op)
        return dist.all_to_all_single(self.grad[slot].view(-1), self.grad_in[slot], self.fwd_send_splits, self.fwd_recv_splits,
                                      group=self.pg, async_op=async_op)

    def _backward(self, slot: int) -> None:
        q = self.quant
        if q is not None and q.bwd_bits:
            q.dequantize(self.q_grad[slot], q.bwd_bits, self.grad[slot].view(-1))
        self.backward(self.grad[slot], *self._req[slot])

    def recv_block(self, slot: int, src: int) -> torch.Tensor:
        """[B_local, widths[src]] view of what rank ``src`` sent me in the forward exchange ([tables of src, B_local, dim] for the
        blocked layout)"""
        o = sum(self.fwd_recv_splits[:src])
        flat = self.recv[slot][o:o + self.fwd_recv_splits[src]]
        if self.layout == "blocked":
            return flat.view(self.widths[src] // self.dim, self.local_batch, self.dim)
        return flat.view(self.local_batch, self.widths[src])

    def selfcheck(self, slot: int, peer_block: Callable, exact: bool = True, rtol: float = 1e-5, local_only: bool = False) -> dict:
        """Verify the payload of the forward exchange held in ``slot`` (already waited for) -- the reference's ``--c 1``
        idea (comms_utils.py:997-1055: a collective's result is compared with what it must be), applied to the pooled
        all-to-all: ``peer_block(src)`` returns what rank ``src`` must have sent THIS rank, ``[B_local, k]`` = the first k
        columns of its block, recomputed locally from rank ``src``'s seeds (or None to skip that peer).  Every rank checks
        its own receive buffer; the verdicts are combined with one MIN all-reduce, so every rank returns the same dict.
        ``exact``: bit for bit (same kernel on both sides); otherwise ``rtol`` relative to the block's magnitude.
        ``local_only``: no collective here -- the caller combines the ranks' verdicts itself (:meth:`combine_selfcheck`), so that
        a rank whose recomputation fails cannot leave the others waiting in an all-reduce."""
        bad, checked, worst = [], 0, 0.0
        for src in range(self.world):
            exp = peer_block(src)
            if exp is None:
                continue
            checked += 1
            blk = self.recv_block(slot, src)
            got = (blk[0] if self.layout == "blocked" else blk)[:, :exp.shape[1]]      # table 0 of the source's chunk
            if exact:
                same = bool(torch.equal(got, exp))
            else:
                same = bool(((got - exp).abs() <= rtol * exp.abs().max().clamp_min(1e-30)).all())
            if not same:
                bad.append(src)
                worst = max(worst, float((got - exp).abs().max()))
        local = {"this_rank_ok": not bad, "peers_checked": checked, "mismatched_sources_on_this_rank": bad, "max_abs_diff": worst}
        return local if local_only else self.combine_selfcheck(local)

    def combine_selfcheck(self, local: dict) -> dict:
        """one MIN all-reduce over the ranks' local verdicts (every rank must call it; a rank whose check could not run passes
        ``{"error": ...}`` and turns the global verdict into "not run on every rank")"""
        ran = 0 if "error" in local else 1
        ok = 1 if (ran and local.get("this_rank_ok")) else 0
        flag = torch.tensor([ok, ran, int(local.get("peers_checked", 0))], dtype=torch.int64, device=self.recv[0].device)
        if dist.is_initialized():
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        ok_all, ran_all, min_checked = bool(int(flag[0])), bool(int(flag[1])), int(flag[2])
        verdict = "ok" if ok_all else ("not run on every rank" if not ran_all else "MISMATCH")
        return dict(local, a2a_selfcheck=verdict, min_peers_checked_over_ranks=min_checked, ranks=self.world)

    def bytes_per_rank(self) -> int:
        """output-tensor bytes of ONE exchange per rank (the reference's ``memSize``, which stays the fp32 size under
        ``--bitwidth``: the report scales busBW by bitwidth / 32 instead, comms.py:1149); forward and backward are equal
        in total over the ranks, not per rank, when the shards are uneven"""
        return self.recv[0].numel() * 4

    def wire_bytes_per_rank(self):
        """bytes this rank RECEIVES per (forward, gradient) exchange as actually sent: quantised rows where enabled"""
        q = self.quant
        f = self.q_recv[0].numel() if q is not None and q.fwd_bits else self.recv[0].numel() * 4
        b = self.q_grad[0].numel() if q is not None and q.bwd_bits else self.grad[0].numel() * 4
        return f, b

    # -- the pipelined step ---------------------------------------------------------------------------
    def step(self, indices, offsets) -> None:
        k, s = self._k, self._k % self.slots
        # slot s was last used by batch k - slots, whose backward ran at step k - slots + 2 on this stream: free.
        self._lookup(s, indices, offsets)
        self._req[s] = (indices, offsets)
        self._fwd_work[s] = self.fwd_a2a(s)                      # on the pg stream, after the lookup; under what follows
        if k >= 1:                                               # batch k-1: its pooled embeddings have arrived -> gradient back
            p = (k - 1) % self.slots
            self._fwd_work[p].wait()
            self._fwd_arrived(p)
            self._bwd_work[p] = self.bwd_a2a(p)
        if k >= 2:                                               # batch k-2: its gradient is home -> fused backward
            q = (k - 2) % self.slots
            self._bwd_work[q].wait()
            self._backward(q)
        self._k += 1

    def drain(self) -> None:
        """finish the batches still in flight (the last two): their gradient exchange and backward"""
        k = self._k
        for b in (k - 1, k - 2):
            if b < 0:
                continue
            p = b % self.slots
            if b == k - 1:
                self._fwd_work[p].wait()
                self._fwd_arrived(p)
                self._bwd_work[p] = self.bwd_a2a(p)
        for b in (k - 2, k - 1):
            if b < 0:
                continue
            q = b % self.slots
            self._bwd_work[q].wait()
            self._backward(q)
        self._k = 0

    def step_serial(self, indices, offsets) -> None:
        """the same batch with nothing overlapped: lookup, exchange, gradient exchange, backward, each waited for
        (what the reference does: barrier after every region, dlrm.py:119-123,1196-1290)"""
        self._lookup(0, indices, offsets)
        self._req[0] = (indices, offsets)
        self.fwd_a2a(0).wait()
        self._fwd_arrived(0)
        self.bwd_a2a(0).wait()
        self._backward(0)


class RowQuant:
    """How a :class:`ShardedEmbeddingExchange` quantises its payloads (the reference's ``--bitwidth`` for the pooled
    all-to-all): rows of ``dim`` values, ``fwd_bits`` / ``bwd_bits`` in {0 (fp32), 16, 8, 4, 2} for the forward and the
    gradient exchange.  ``quantize(src_f32, bits, out_u8)`` / ``dequantize(src_u8, bits, out_f32)`` default to the HIP row
    quantisers (:mod:`param_amd.quant`, GPU tensors only); ``lookup_quantized(indices, offsets, out_u8)`` -- optional --
    is a lookup that writes quantised rows itself (``BatchedEmbeddingBagMI355.lookup_quantized``), which makes the
    forward payload free.  The gloo tests inject numpy stand-ins."""

    def __init__(self, dim: int, fwd_bits: int = 0, bwd_bits: int = 0, quantize: Callable = None, dequantize: Callable = None,
                 lookup_quantized: Callable = None):
        self.dim, self.fwd_bits, self.bwd_bits = int(dim), int(fwd_bits), int(bwd_bits)
        for b in (self.fwd_bits, self.bwd_bits):
            if b not in (0, 16, 8, 4, 2):
                raise ValueError(f"bit width must be 0 (fp32), 16, 8, 4 or 2, got {b}")
        self.lookup_quantized = lookup_quantized
        if quantize is None or dequantize is None:
            from ... import quant as _q
            quantize = quantize or (lambda src, bits, out: _q.quantize_rows(src, self.dim, bits, out=out))
            dequantize = dequantize or (lambda src, bits, out: _q.dequantize_rows(src, self.dim, bits, out=out))
        self.quantize, self.dequantize = quantize, dequantize

    def row_bytes(self, bits: int) -> int:
        return 0 if not bits else 2 * self.dim if bits == 16 else self.dim + 8 if bits == 8 else self.dim * bits // 8 + 4
