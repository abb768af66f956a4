#!/usr/bin/env python3
"""Round-3 apply-kernel experiments at benchmark size (48 x 10 M x 128 fp32 or 64 bf16 tables, B 8192, L 20), one JSON line each:
  * destination-row cache policy 0 (plain) / 1 (nt) / 3 (plain load, sc1 store: line dropped from L2) / 4 (nt load, sc1 store)
  * sorted order: ascending rows vs the lowest row digit sorted last (PARAM_AMD_EXP_DIGIT_ROT=1: neighbours in the sorted
    array are 256 rows x stride apart -- what the ascending order is worth to address translation)."""
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
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--policies", default="0,1,3,4")
ap.add_argument("--runs", default="legacy,rot,seg0,seg1,seg2", help="which sorts to run")
ap.add_argument("--requests", default="uniform,zipf1.05")
a = ap.parse_args()
dev = torch.device("cuda:0")
T, R, D, B, L = a.tables, a.rows, 128, 8192, 20
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
es = 4 if a.dtype == "fp32" else 2
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", seed=1, fused_update=False)
grad = torch.randn((B, T * D), device=dev)
reqs = {"uniform": tbe_request([R] * T, B, L, 0.0, device=dev, seed=2), "zipf1.05": tbe_request([R] * T, B, L, 1.05, device=dev, seed=1)}
reqs = {k: v for k, v in reqs.items() if k in a.requests.split(",")}
runs = a.runs.split(",")
bwd_bytes = T * B * L * (2 * D * es + 8) + T * B * (D * 4 + 8)


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


def run(tag, pol):
    param_amd.set_tuning(nt_loads=pol)
    for name, (idx, off) in reqs.items():
        sort_s = timed(lambda: m.sort_indices(idx, off, batch=B), a.iters)
        apply_s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True), a.iters)
        both_s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B), a.iters)
        print(json.dumps(dict(tag, exp="bwd", row_policy=pol, indices=name, dtype=a.dtype, tables=T, sort_ms=sort_s * 1e3,
                              apply_ms=apply_s * 1e3, total_ms=both_s * 1e3, alg_frac_total=bwd_bytes / both_s / 8e12,
                              alg_frac_apply=bwd_bytes / apply_s / 8e12)), flush=True)


pols = [int(x) for x in a.policies.split(",")]
# ~50 ms of the same work before anything is timed: the first block after the set-up phase rides a clock / power transient
for idx, off in reqs.values():
    for _ in range(15):
        m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
torch.cuda.synchronize()
# round 2's sort (host-side plan, fixed pooling): ascending rows vs lowest digit sorted last
param_amd.set_backward_tuning(sort_impl=2)
for rot in ("0", "1"):
    if ("legacy", "rot")[int(rot)] not in runs:
        continue
    os.environ["PARAM_AMD_EXP_DIGIT_ROT"] = rot
    for pol in (pols if rot == "0" else pols[:1]):
        run({"sort": "legacy", "digit_rot": int(rot)}, pol)
os.environ["PARAM_AMD_EXP_DIGIT_ROT"] = "0"
if "ph2" in runs:     # round 2's two-phase apply (lower / upper half of the bags in two launches), now with the sc1 row stores
    param_amd.set_backward_tuning(sort_impl=2, max_phases=2)
    for pol in pols:
        run({"sort": "legacy", "phases": 2}, pol)
# the segmented sort: LSD passes / low-digit partition + local / top-digit partition + local
param_amd.set_backward_tuning(sort_impl=0)
for mode in (0, 1, 2):
    if f"seg{mode}" not in runs:
        continue
    param_amd.set_sort_tuning(mode)
    for pol in pols:
        run({"sort": "seg", "mode": mode}, pol)
param_amd.set_sort_tuning()
param_amd.set_backward_tuning()
param_amd.set_tuning()
