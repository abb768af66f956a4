#!/usr/bin/env python3
"""Workload for rocprofv3 passes: a few launches of each kernel of interest with KNOWN byte
counts, so the PMC rows (FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS) can be calibrated:

  calib_write : pm_fill_random over 8 GiB            (writes exactly 8 GiB, reads ~0)
  calib_read  : torch sum over the same 8 GiB fp32   (reads exactly 8 GiB)
  fwd uniform : embbag_fwd_kernel, alpha=0           (no reuse: HBM bytes ~ algorithmic bytes)
  fwd zipf    : embbag_fwd_kernel, alpha=1.05        (hot rows hit L2/MALL: HBM bytes < algorithmic)
  bwd uniform / zipf : sort + bwd_sorted_main_kernel + bwd_sorted_fixup_kernel (the deterministic backward)

Launch order is fixed and printed, so dispatches can be matched by order in the CSV.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--tables", type=int, default=48)
p.add_argument("--rows", type=int, default=10_000_000)
p.add_argument("--dim", type=int, default=128)
p.add_argument("--batch", type=int, default=8192)
p.add_argument("--pooling", type=int, default=20)
p.add_argument("--reps", type=int, default=3)
p.add_argument("--bwd", action="store_true")
p.add_argument("--manifest", default="")
a = p.parse_args()

dev = torch.device("cuda:0")
T, R, D, B, L = a.tables, a.rows, a.dim, a.batch, a.pooling
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
out = torch.empty((B, T * D), dtype=torch.float32, device=dev)
iu, ou = tbe_request([R] * T, B, L, alpha=0.0, device=dev, seed=1)
iz, oz = tbe_request([R] * T, B, L, alpha=1.05, device=dev, seed=1)
torch.cuda.synchronize()

manifest = {"alg_bytes_fwd": algorithmic_bytes(T, B, L, D, 4), "T": T, "R": R, "D": D, "B": B, "L": L,
            "lookups": T * B * L, "calib_bytes": 8 << 30, "reps": a.reps, "order": []}

calib = torch.empty(2 << 30, dtype=torch.float32, device=dev)  # 8 GiB
for _ in range(a.reps):
    param_amd.fill_random_(calib, "uniform", 0.0, 1.0, seed=3)
    manifest["order"].append("calib_write")
torch.cuda.synchronize()
for _ in range(a.reps):
    calib.sum()
    manifest["order"].append("calib_read")
torch.cuda.synchronize()
for tag, (i, o) in (("fwd_uniform", (iu, ou)), ("fwd_zipf", (iz, oz))):
    for _ in range(a.reps + 1):  # first one is a warm-up for the caches
        m.lookup(i, o, out=out, batch=B)
        manifest["order"].append(tag)
    torch.cuda.synchronize()
if a.bwd:
    grad = torch.randn((B, T * D), device=dev)
    manifest["alg_bytes_bwd"] = T * B * L * (2 * D * 4 + 8) + T * B * (D * 4 + 8)
    for _ in range(a.reps):
        m.scatter_add_(grad, iu, ou, alpha=-1e-6, batch=B)
        manifest["order"].append("bwd_uniform")
    torch.cuda.synchronize()
    for _ in range(a.reps):
        m.scatter_add_(grad, iz, oz, alpha=-1e-6, batch=B)
        manifest["order"].append("bwd_zipf")
    torch.cuda.synchronize()
if a.manifest:
    json.dump(manifest, open(a.manifest, "w"), indent=1)
print(json.dumps(manifest))
