#!/usr/bin/env python3
"""Round 5: the backward of ONE rank of an N-GPU table-wise sharded step on one GPU -- 64 / N tables x 10 M x 128 fp32, global batch
N x 8192, L = 20, uniform indices -- with the hybrid path off and on (its dup maps now grow with the table's lookups).

    PROBE_TABLES=8|12|24 python tools/r5_rank_shape_probe.py        (N = 8 | 4 | 2; 12 tables stand for N = 4's 16: 16 x 5.12 GB + work space fits too)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import param_amd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

T = int(os.environ.get("PROBE_TABLES", "8"))
W = {8: 8, 12: 4, 16: 4, 24: 2, 32: 2}[T]
dev = torch.device("cuda", 0)
R, D, Bl, L = int(os.environ.get("PROBE_ROWS", "10000000")), 128, 8192, 20
B = int(os.environ.get("PROBE_BATCH", str(Bl * W)))      # (PROBE_BATCH / PROBE_ROWS: shapes off the N-GPU line, to separate bags per table from tables)
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", layout="bd", seed=7, fused_update=False)
grad = torch.randn(B, T * D, device=dev)
n = T * B * L
bwd_bytes = n * (2 * D * 4 + 8) + T * B * (D * 4 + 8)


def timed(fn, iters=15, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha=alpha, device=dev, seed=5)
    for en in (0, 1):
        param_amd.set_hybrid_tuning(en)
        b = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B))
        st = m.sort_status(idx, off, batch=B)
        print(json.dumps({"n_gpus_shape": W, "tables": T, "global_batch": B, "lookups_per_table": B * L, "indices": "uniform" if alpha == 0.0 else "zipf1.05",
                          "hybrid": en, "bwd_us": round(b * 1e6, 1), "alg_frac": round(bwd_bytes / b / 8e12, 4), **st}), flush=True)
param_amd.set_hybrid_tuning()
