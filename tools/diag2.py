#!/usr/bin/env python3
"""Backward diagnostics: does the gradient layout ([B, T*D] vs [T, B, D]) change the apply kernel's
L2 behaviour?  (strided 512-B slices of a [B, T*D] row map onto few L2 sets)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.embedding_bag import _TableSet, _bwd, _fwd, _sort_indices
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
for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=2)
    for layout in ("bd", "tbd"):
        ts = _TableSet([m.table(t) for t in range(T)], layout)
        shape = (B, T * D) if layout == "bd" else (T, B, D)
        grad = torch.randn(shape, device=dev)
        _sort_indices(ts, idx, off, B)
        s_apply = timeit(lambda: _bwd(ts, grad, idx, off, B, ts.d_ptrs, torch.float32, -1e-6, presorted=True))
        s_all = timeit(lambda: _bwd(ts, grad, idx, off, B, ts.d_ptrs, torch.float32, -1e-6))
        out = torch.empty(shape, device=dev)
        s_fwd = timeit(lambda: _fwd(ts, idx, off, B, out=out))
        print(json.dumps({"alpha": alpha, "layout": layout, "bwd_apply_ms": s_apply * 1e3, "bwd_total_ms": s_all * 1e3,
                          "bwd_apply_frac": bb / s_apply / 8e12, "bwd_total_frac": bb / s_all / 8e12, "fwd_ms": s_fwd * 1e3}), flush=True)
        del grad, out
