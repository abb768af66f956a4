#!/usr/bin/env python3
"""The key sort alone (no apply): `sort_indices` with the hybrid path off, benchmark-shaped request (48 x 10 M rows, B 8192, L 20),
narrow tables (the sort reads row counts only).  Run under `rocprofv3 --kernel-trace` by tools/r4_timeline.sh-style wrappers:
experiment builds of the sort kernels whose results are not fit to be applied can be timed here.

    python tools/r4_sort_probe.py [--requests uniform,zipf1.05] [--iters 6]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import param_amd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--requests", default="uniform,zipf1.05")
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--tables", type=int, default=48)
a = ap.parse_args()
dev = torch.device("cuda", 0)
T, R, B, L = a.tables, 10_000_000, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, 8, dtype=torch.float32, device=dev, init="normal", layout="tbd", seed=1, fused_update=False)
param_amd.set_hybrid_tuning(0)
for rq in a.requests.split(","):
    alpha = 0.0 if rq == "uniform" else float(rq[4:])
    idx, off = tbe_request([R] * T, B, [L] * T, alpha=alpha, device=dev, seed=3)
    for _ in range(3):
        m.sort_indices(idx, off)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        m.sort_indices(idx, off)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"exp": "sort_only", "indices": rq, "sort_ms": round(e0.elapsed_time(e1) / a.iters, 4), "lib": os.environ.get("PARAM_AMD_LIB", "default")}), flush=True)
