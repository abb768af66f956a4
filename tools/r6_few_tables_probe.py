#!/usr/bin/env python3
"""Round 6: from how many bag-major tiles on does the hybrid backward pay?  T tables of 10 M rows x 128 fp32, batch B, pooling 20, uniform
rows; pm_set_hybrid_tuning 0 / 1 taking turns (an even request is offered the hybrid path by the library itself).  The bag-major kernel
tiles 128 bags: T * B / 128 workgroups.  One JSON line per (T, B, setting, round)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

dev = torch.device("cuda:0")
D, L, R = 128, 20, 10_000_000


def timed(fn, n=20):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for T, B in ((1, 8192), (2, 8192), (4, 8192), (8, 8192), (12, 8192), (16, 8192), (24, 8192), (32, 8192), (1, 65536), (2, 65536), (4, 65536), (8, 2048), (32, 2048)):
    m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=torch.float32, device=dev, init="normal", layout="bd", seed=1, fused_update=False)
    grad = torch.randn(B, T * D, device=dev)
    idx, off = tbe_request([R] * T, B, [L] * T, alpha=0.0, device=dev, seed=3)
    n = T * B * L
    bwd_bytes = n * (2 * D * 4 + 8) + T * B * (D * 4 + 8)
    for rnd in range(2):
        for hyb in (0, 1):
            param_amd.set_hybrid_tuning(hyb)
            s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B))
            st = m.sort_status(idx, off, batch=B)
            print(json.dumps({"exp": "few_tables_bwd", "T": T, "B": B, "tiles": T * B // 128, "hybrid": hyb, "round": rnd, "us": round(s * 1e6, 2),
                              "alg_frac": round(bwd_bytes / s / 8e12, 4), "hybrid_tables": st["hybrid_tables"], "lds_pairs": st["lds_pairs"],
                              "pairs_sorted": st["pairs_sorted"]}), flush=True)
    del m, grad, idx, off
    torch.cuda.empty_cache()
param_amd.set_hybrid_tuning()
