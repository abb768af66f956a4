#!/usr/bin/env python3
"""Backward: streaming (nt) destination-row accesses on/off; kBatch-independent."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.embedding_bag import _TableSet, _bwd, _sort_indices
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
def timeit(fn, steps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps
bb = T * B * L * (2 * D * 4 + 8) + T * B * (D * 4 + 8)
ts = _TableSet([m.table(t) for t in range(T)], "bd")
grad = torch.randn((B, T * D), device=dev)
for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=2)
    _sort_indices(ts, idx, off, B)
    for rep in range(2):
        for nt in (0, 1):
            param_amd.set_tuning(0, 0, -1, nt)
            s = timeit(lambda: _bwd(ts, grad, idx, off, B, ts.d_ptrs, torch.float32, -1e-6, presorted=True))
            print(json.dumps({"alpha": alpha, "nt_rows": nt, "bwd_apply_ms": s * 1e3, "frac": bb / s / 8e12}), flush=True)
