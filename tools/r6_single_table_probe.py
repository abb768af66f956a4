#!/usr/bin/env python3
"""Round 6: what the hybrid backward would do for ONE large multi-hot table (the 40 M-row Criteo tables, pooling 100 / 27 / 12 / 7 / 3,
batch 8192) -- a single-table request is even, so the library offers it the hybrid path by itself: pm_set_hybrid_tuning 0 / 1 taking turns,
uniform rows.  One JSON line per (pooling, setting, round)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

dev = torch.device("cuda:0")
rows, D, B = 40_000_000, 128, 8192
m = param_amd.BatchedEmbeddingBagMI355([rows], D, dtype=torch.float32, device=dev, init="normal", layout="bd", seed=1, fused_update=False)
grad = torch.randn(B, D, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for L in (100, 27, 12, 7, 3):
    idx, off = tbe_request([rows], B, [L], alpha=0.0, device=dev, seed=3)
    n = B * L
    bwd_bytes = n * (2 * D * 4 + 8) + B * (D * 4 + 8)
    for rnd in range(2):
        for hyb in (0, 1):
            param_amd.set_hybrid_tuning(hyb)
            s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B))
            st = m.sort_status(idx, off, batch=B)
            print(json.dumps({"exp": "single_table_bwd", "pooling": L, "lookups": n, "hybrid": hyb, "round": rnd, "us": round(s * 1e6, 2),
                              "alg_frac": round(bwd_bytes / s / 8e12, 4), **st}), flush=True)
param_amd.set_hybrid_tuning()
