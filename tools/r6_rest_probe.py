#!/usr/bin/env python3
"""Round 6: same-box A/B of the hybrid backward's left-over kernel (pm_set_hybrid_rest 0 / 1 taking turns) at benchmark size:
one fused backward call per step, uniform (hybrid) and Zipf (sorted) requests.  One JSON line per (setting, request, repeat)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=48)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--requests", default="uniform,zipf1.05")
ap.add_argument("--settings", default="0,1", help="pm_set_hybrid_rest values, taking turns")
ap.add_argument("--workload", default="tables")
ap.add_argument("--layout", default="tbd", choices=["bd", "tbd"])
a = ap.parse_args()
dev = torch.device("cuda:0")
D, B, L = 128, a.batch, 20
if a.workload == "criteo":
    from param_amd.compute.pt import dataset as ds

    rows, pools = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot)
else:
    rows, pools = [a.rows] * a.tables, [L] * a.tables
T = len(rows)
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
es = 4 if a.dtype == "fp32" else 2
m = param_amd.BatchedEmbeddingBagMI355(rows, D, dtype=dt, device=dev, init="normal", seed=1, fused_update=False, layout=a.layout)
grad = torch.randn((B, T * D) if a.layout == "bd" else (T, B, D), device=dev)
reqs = {"uniform": tbe_request(rows, B, pools, 0.0, device=dev, seed=2), "zipf1.05": tbe_request(rows, B, pools, 1.05, device=dev, seed=1)}
reqs = {k: v for k, v in reqs.items() if k in a.requests.split(",")}
n_lookups = B * sum(pools)
bwd_bytes = n_lookups * (2 * D * es + 8) + T * B * (D * 4 + 8)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for idx, off in reqs.values():
    for _ in range(10):
        m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
torch.cuda.synchronize()
for rep in range(a.reps):
    for rest in [int(x) for x in a.settings.split(",")]:
        param_amd.set_hybrid_rest(rest)
        for name, (idx, off) in reqs.items():
            s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B), a.iters)
            st = m.sort_status(idx, off, batch=B)
            print(json.dumps({"exp": "hyb_rest_ab", "rest": rest, "rep": rep, "indices": name, "dtype": a.dtype, "tables": T,
                              "workload": a.workload, "ms": round(s * 1e3, 4), "alg_frac": round(bwd_bytes / s / 8e12, 4), **st}), flush=True)
param_amd.set_hybrid_rest()
