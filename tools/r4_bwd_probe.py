#!/usr/bin/env python3
"""Round 4: the backward at benchmark size (48 x 10 M x 128 fp32, or 64 bf16 tables; B 8192, L 20) with the hybrid path
off / on / on with the second stream, uniform and Zipf requests: sort alone, apply alone (pre-sorted), sort + apply, and what
the sort left on the device (pairs sorted, hybrid tables).  One JSON line per setting."""
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
ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--settings", default="0,2", help="hybrid enable values")
ap.add_argument("--requests", default="uniform,zipf1.05")
ap.add_argument("--workload", default="tables")
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--layout", default="bd", choices=["bd", "tbd"])
ap.add_argument("--row-policies", default="-1", help="comma-separated pm_set_tuning nt_loads values (destination-row cache policy; -1 = the default)")
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
    for _ in range(15):
        m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
torch.cuda.synchronize()
from param_amd import _lib as _pm_lib  # noqa: E402

for setting, pol in [(s_, int(p_)) for s_ in a.settings.split(",") for p_ in a.row_policies.split(",")]:
    en = int(setting.split(":")[0])
    if hasattr(param_amd, "set_hybrid_tuning"):
        param_amd.set_hybrid_tuning(en)
    _pm_lib.set_tuning(nt_loads=pol)
    for name, (idx, off) in reqs.items():
        sort_s = timed(lambda: m.sort_indices(idx, off, batch=B), a.iters)
        apply_s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True), a.iters)
        both_s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B), a.iters)
        st = m.sort_status(idx, off, batch=B) if hasattr(m, "sort_status") else {}
        print(json.dumps({"exp": "bwd_hybrid", "enable": en, "indices": name, "dtype": a.dtype, "tables": T,
                          "workload": a.workload, "layout": a.layout, "row_policy": pol, "sort_call_ms": round(sort_s * 1e3, 4), "apply_call_ms": round(apply_s * 1e3, 4),
                          "total_ms": round(both_s * 1e3, 4), "alg_frac_total": round(bwd_bytes / both_s / 8e12, 4),
                          "alg_frac_apply": round(bwd_bytes / apply_s / 8e12, 4), **st}), flush=True)
if hasattr(param_amd, "set_hybrid_tuning"):
    param_amd.set_hybrid_tuning()
