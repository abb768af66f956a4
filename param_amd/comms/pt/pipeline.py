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
