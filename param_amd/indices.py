"""Synthetic index generation for the EmbeddingBag benchmarks.

Two forms of the reference's generator (``init_indices``, reference
``train/compute/pt/pytorch_emb.py:138-160``):

* :func:`init_indices` -- same results as the reference for the same torch / numpy seeds
  (alpha == 0: ``torch.randint``; alpha > 0: ``np.random.choice`` over the pmf
  ``(i+1)^-alpha`` with 2*nnz draws per bag, first nnz distinct kept).  Host-side, python
  loop over bags: small problems and parity tests only.  Differences from the reference,
  on purpose: ``alpha`` given as a string is accepted (reference bug R2, driver.py:43-45),
  and an under-filled bag raises a clear ``ValueError`` instead of a broadcast error (R4).
* :func:`zipf_indices` -- the same distribution (same pmf, hot rows = low row ids, per-bag
  de-duplication of 2*nnz draws) vectorised with torch ops on any device, for the
  64 x 10M-row configurations where the python loop is prohibitive.

Plus the TBE request builder (reference ``generate_requests``,
``train/compute/python/workloads/pytorch/split_table_batched_embeddings_ops.py:93-135``).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch


def init_indices(alpha, features: int, batch: int, nnz: int) -> torch.Tensor:
    """Reference-compatible index generator (int64 ``[batch*nnz]`` on the host)."""
    alpha = float(alpha)
    if alpha == 0.0:
        return torch.randint(0, features, (batch * nnz,))
    pmf = np.arange(1, features + 1, dtype=np.float64) ** (-alpha)
    pmf /= pmf.sum()
    draws = np.random.choice(features, size=(batch, 2 * nnz), replace=True, p=pmf)
    picked = np.empty((batch, nnz), dtype=draws.dtype)
    for b in range(batch):
        first = list(dict.fromkeys(draws[b].tolist()))[:nnz]  # first nnz distinct, draw order
        if len(first) < nnz:
            raise ValueError(
                f"bag {b}: only {len(first)} distinct rows in {2 * nnz} Zipf(alpha={alpha}) draws "
                f"over {features} rows (need nnz={nnz}); the reference fails here too (broadcast error)")
        # the reference stores list(set): reproduce the set's iteration order exactly
        picked[b] = list(set(np.asarray(first, dtype=draws.dtype)))
    return torch.from_numpy(picked.reshape(-1)).to(torch.int64)


def fixed_offsets(batch: int, nnz: int, device=None, dtype=torch.int64, include_last: bool = False) -> torch.Tensor:
    """offsets[i] = i*nnz (pytorch_emb.py:172-174), vectorised."""
    n = batch + 1 if include_last else batch
    return torch.arange(n, dtype=dtype, device=device) * nnz


_cdf_cache: dict = {}


def _zipf_cdf(alpha: float, features: int, device) -> torch.Tensor:
    key = (float(alpha), int(features), str(device))
    if key not in _cdf_cache:
        _cdf_cache.clear()  # one 8*features-byte table at a time
        pmf = torch.arange(1, features + 1, dtype=torch.float64, device=device).pow_(-float(alpha))
        cdf = torch.cumsum(pmf, 0)
        cdf /= cdf[-1].clone()
        _cdf_cache[key] = cdf
    return _cdf_cache[key]


def zipf_indices(alpha: float, features: int, batch: int, nnz: int, device=None,
                 generator: Optional[torch.Generator] = None, dedupe: bool = True) -> torch.Tensor:
    """Vectorised Zipf(alpha) indices, int64 ``[batch*nnz]`` on ``device``.

    Same sampling scheme as the reference (inverse-CDF over pmf (i+1)^-alpha, 2*nnz draws per
    bag, first nnz distinct in draw order); within-bag order is draw order (the reference's is
    a python-set order: irrelevant to a sum).  Bags that under-fill are redrawn.
    """
    alpha = float(alpha)
    if alpha == 0.0:
        return torch.randint(0, features, (batch * nnz,), device=device, generator=generator)
    cdf = _zipf_cdf(alpha, features, device)
    width = 2 * nnz if dedupe else nnz

    def draw(n_rows: int) -> torch.Tensor:
        u = torch.rand((n_rows, width), dtype=torch.float64, device=device, generator=generator)
        return torch.searchsorted(cdf, u, right=True).clamp_(max=features - 1)

    cand = draw(batch)
    if not dedupe:
        return cand.reshape(-1)
    out = torch.empty((batch, nnz), dtype=torch.int64, device=device)
    todo = torch.arange(batch, device=device)
    for _ in range(64):
        # dup[b,i] = candidate i equals an earlier candidate of the same bag
        eq = cand.unsqueeze(2) == cand.unsqueeze(1)                      # [n, w, w]
        earlier = torch.ones(width, width, dtype=torch.bool, device=device).tril_(-1)
        keep = ~(eq & earlier).any(dim=2)                                # first occurrences
        rank = torch.cumsum(keep, dim=1) - 1
        ok = keep.sum(dim=1) >= nnz
        sel = keep & (rank < nnz)
        rows_ok = torch.nonzero(ok).squeeze(1)
        if rows_ok.numel():
            c, s, r = cand[rows_ok], sel[rows_ok], rank[rows_ok]
            tmp = torch.empty((rows_ok.numel(), nnz), dtype=torch.int64, device=device)
            bi = torch.arange(rows_ok.numel(), device=device).unsqueeze(1).expand_as(c)
            tmp[bi[s], r[s]] = c[s]
            out[todo[rows_ok]] = tmp
        todo = todo[~ok]
        if todo.numel() == 0:
            break
        cand = draw(todo.numel())
    else:
        raise ValueError(f"could not draw {nnz} distinct Zipf(alpha={alpha}) rows out of {features}")
    return out.reshape(-1)


def tbe_request(rows: Sequence[int], batch: int, pooling, alpha: float = 0.0, device=None,
                seed: int = 0, index_dtype=torch.int64):
    """A batched request in the TBE layout the reference's data generator emits
    (split_table_batched_embeddings_ops.py:191-208): indices = per-table lists concatenated
    table-major, offsets = ``[0, L, 2L, ...]`` running on across tables, length ``T*B+1``.
    alpha == 0 -> uniform ``randint``; alpha > 0 -> :func:`zipf_indices` (pmf of
    pytorch_emb.py:143, the benchmark's skew model).  ``pooling`` is one bag size for every table or a
    per-table list (multi-hot sizes of a Criteo-style model); a table with fewer rows than its
    pooling factor is sampled with replacement."""
    gen = torch.Generator(device=device if device is not None else "cpu")
    gen.manual_seed(seed)
    pools = [int(pooling)] * len(rows) if isinstance(pooling, int) else [int(x) for x in pooling]
    assert len(pools) == len(rows)
    parts = []
    for r, L in zip(rows, pools):
        if alpha == 0.0 or int(r) < 4 * L:
            parts.append(torch.randint(0, int(r), (batch * L,), device=device, generator=gen))
        else:
            parts.append(zipf_indices(alpha, int(r), batch, L, device=device, generator=gen))
    indices = torch.cat(parts).to(index_dtype)
    if len(set(pools)) == 1:
        offsets = torch.arange(len(rows) * batch + 1, dtype=index_dtype, device=device) * pools[0]
    else:
        lens = torch.cat([torch.full((batch,), L, dtype=torch.int64, device=device) for L in pools])
        offsets = torch.zeros(len(rows) * batch + 1, dtype=torch.int64, device=device)
        torch.cumsum(lens, 0, out=offsets[1:])
        offsets = offsets.to(index_dtype)
    return indices, offsets
