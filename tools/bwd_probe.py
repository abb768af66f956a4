#!/usr/bin/env python3
"""Sorted backward at benchmark size (48 x 10 M x 128 fp32, B 8192, L 20) under every backward tuning: own vs rocPRIM sort,
(row, table) vs (table, row) order, XCD-affine apply tiles, streaming row accesses.  One JSON line per setting and index
distribution: sort / apply / total times and the algorithmic fraction (1058 B per lookup)."""
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
ap.add_argument("--layout", default="bd")
ap.add_argument("--nt", type=int, default=0, help="row cache policy of the apply kernel: 0 default, 1 nt, 2 system scope")
ap.add_argument("--configs", default="", help="semicolon-separated sort_impl,order,xcd,nt tuples (default: the full matrix)")
a = ap.parse_args()
dev = torch.device("cuda:0")
T, R, D, B, L = a.tables, a.rows, 128, 8192, 20
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
es = 4 if a.dtype == "fp32" else 2
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", seed=1, fused_update=False, layout=a.layout)
grad = torch.randn((B, T * D) if a.layout == "bd" else (T, B, D), device=dev)
reqs = {"uniform": tbe_request([R] * T, B, L, 0.0, device=dev, seed=2), "zipf1.05": tbe_request([R] * T, B, L, 1.05, device=dev, seed=1)}
bwd_bytes = T * B * L * (2 * D * es + 8) + T * B * (D * 4 + 8)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


# (sort_impl, order, xcd, max_phases)
CONFIGS = [(1, 1, 0, 1), (0, 0, 0, 1), (0, 1, 0, 1), (0, 1, 1, 1), (0, 1, 1, 2), (0, 1, 0, 2), (1, 1, 1, 2)]
if a.configs:
    CONFIGS = [tuple(int(x) for x in c.split(",")) for c in a.configs.split(";")]
# (round 6: sort_impl 1 / 2 -- rocPRIM, round 2's sort -- exist in the alternates build only; the whole matrix runs on it)
from param_amd import _lib  # noqa: E402

_alt = _lib.use_alternates()
_alt.__enter__()
for sort_impl, order, xcd, ph in CONFIGS:
    param_amd.set_backward_tuning(sort_impl, order, xcd, ph)
    param_amd.set_tuning(nt_loads=a.nt)
    for name, (idx, off) in reqs.items():
        sort_s = timed(lambda: m.sort_indices(idx, off, batch=B), a.iters)
        apply_s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True), a.iters)
        both_s = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B), a.iters)
        print(json.dumps({"sort": ["own", "rocprim"][sort_impl], "order": ["row,table", "table,row"][order], "xcd": xcd, "max_phases": ph, "row_policy": a.nt,
                          "indices": name, "dtype": a.dtype, "layout": a.layout, "sort_ms": sort_s * 1e3, "apply_ms": apply_s * 1e3, "total_ms": both_s * 1e3,
                          "alg_frac_total": bwd_bytes / both_s / 8e12, "alg_frac_apply": bwd_bytes / apply_s / 8e12}), flush=True)
param_amd.set_backward_tuning()
