#!/usr/bin/env python3
"""Debug aid: reproduce the mid-size sorted-backward case and print the rows that deviate."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.embbag_oracle import COracle, bag_bounds
from param_amd import EmbeddingBagMI355
from param_amd.embedding_bag import _bwd
from param_amd.indices import zipf_indices

DEV = "cuda:0"
orc = COracle()
rng = np.random.default_rng(21)
D = 128
R, B = 20000, 700
lens = rng.integers(0, 60, B)
lens[3] = 5000
off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
n = int(lens.sum())
idx = zipf_indices(1.2, R, n, 1, dedupe=False, generator=torch.Generator().manual_seed(D)).numpy()
idx[off[3]:off[3] + 5000] = 17
grad = rng.standard_normal((B, D)).astype(np.float32)
W = rng.standard_normal((R, D)).astype(np.float32)
exp = orc.bwd_f32(W.copy(), idx, off, grad, None, alpha=-0.03)
# fp64 truth
start, end = bag_bounds(off, B, n)
bag_of = np.repeat(np.arange(B), end - start)
truth = W.astype(np.float64).copy()
np.add.at(truth, idx, -0.03 * grad.astype(np.float64)[bag_of])
m = EmbeddingBagMI355(R, D, _weight=torch.from_numpy(W).to(DEV))
ts = m._tables()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
_bwd(ts, t(grad), t(idx), t(off), B, ts.d_ptrs, torch.float32, -0.03, None)
got = m.weight.data.cpu().numpy()
cnt = np.bincount(idx, minlength=R)
err_o = np.abs(got.astype(np.float64) - exp).max(axis=1)
err_t = np.abs(got.astype(np.float64) - truth).max(axis=1)
err_ot = np.abs(exp.astype(np.float64) - truth).max(axis=1)
order = np.argsort(-err_o)[:15]
print("n", n, "rows>256:", int((cnt > 256).sum()))
for r in order:
    print(f"row {r:6d} count {cnt[r]:6d}  |got-oracle| {err_o[r]:.3e}  |got-truth| {err_t[r]:.3e}  |oracle-truth| {err_ot[r]:.3e}")
# where do the sorted positions of the worst row sit?
key_order = np.argsort(idx, kind="stable")
sorted_rows = idx[key_order]
for r in order[:3]:
    pos = np.nonzero(sorted_rows == r)[0]
    print(f"row {r}: sorted positions {pos[0]}..{pos[-1]} (chunks of 128: {pos[0] // 128}..{pos[-1] // 128})")
