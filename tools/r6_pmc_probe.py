#!/usr/bin/env python3
"""Round 6 workload for the rocprofv3 --pmc / --kernel-trace --stats passes (round 5's probe + request rotation + the mixed-dim Criteo block): ONE phase of ONE workload per process, so that every library kernel a pass
sees belongs to that phase (no labelling by launch order):

    python tools/r6_pmc_probe.py --workload {fp32,bf16,criteo,mixed} --phase {calib,fwd_uniform,fwd_zipf,bwd_uniform,bwd_zipf} [--iters 4] [--rotate 4]

--rotate K (default 4): K distinct requests (bench.py's seeds: 1 / 2 + 1000 k) take turns, as in the bench line's windows since round 6 -- a
replayed request leaves up to 256 MB of its rows in the memory-side cache; the counters of a rotated run see none of that reuse.

fp32 = 48 x 10 M x 128 fp32 (the N = 1 benchmark), bf16 = all 64 tables in bf16, criteo = the 26 MLPerf DLRM-v2 tables (fp32,
[B, sum D] output), mixed = the same tables with dims by table size (dataset.criteo_v2_mixed_dims).  calib: pm_fill_random over 8 GiB (writes exactly 8 GiB) and a torch sum over it (reads exactly 8 GiB), the
known byte counts FETCH_SIZE / WRITE_SIZE are scaled on (MI355X_MICROARCH.md, HBM section).  Prints one JSON line: the phase's
algorithmic bytes per step and the number of steps, which tools/r5_parse_pmc.py divides the counter sums by.
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

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="fp32")
ap.add_argument("--phase", default="fwd_uniform")
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--rotate", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.phase == "calib":
    calib = torch.empty(2 << 30, dtype=torch.float32, device=dev)   # 8 GiB
    for _ in range(a.iters):
        param_amd.fill_random_(calib, "uniform", 0.0, 1.0, seed=3)
    torch.cuda.synchronize()
    for _ in range(a.iters):
        calib.sum()
    torch.cuda.synchronize()
    print(json.dumps({"phase": "calib", "bytes": 8 << 30, "iters": a.iters}))
    sys.exit(0)
D, B = 128, 8192
if a.workload in ("criteo", "mixed"):
    from param_amd.compute.pt import dataset as ds
    rows, pools, dt, layout = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot), torch.float32, "bd"
elif a.workload == "bf16":
    rows, pools, dt, layout = [10_000_000] * 64, [20] * 64, torch.bfloat16, "tbd"
else:
    rows, pools, dt, layout = [10_000_000] * 48, [20] * 48, torch.float32, "tbd"
T = len(rows)
es = 4 if dt == torch.float32 else 2
dims = ds.criteo_v2_mixed_dims(rows) if a.workload == "mixed" else [D] * T
m = param_amd.BatchedEmbeddingBagMI355(rows, dims, dtype=dt, device=dev, init="normal", layout=layout, seed=1000, fused_update=False)
alpha = 0.0 if a.phase.endswith("uniform") else 1.05
reqs = [tbe_request(rows, B, pools, alpha=alpha, device=dev, seed=(2 if alpha == 0.0 else 1) + 1000 * k) for k in range(max(1, a.rotate))]
shape = (B, sum(dims)) if layout == "bd" else (T, B, D)
n = B * sum(pools)
fwd_bytes = sum(algorithmic_bytes(1, B, Lt, Dt, es) for Lt, Dt in zip(pools, dims))
bwd_bytes = sum(B * Lt * (2 * Dt * es + 8) + B * (Dt * 4 + 8) for Lt, Dt in zip(pools, dims))
torch.cuda.synchronize()
if a.phase.startswith("fwd"):
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    for it in range(a.iters):
        idx, off = reqs[it % len(reqs)]
        m.lookup(idx, off, out=out, batch=B)
    alg = fwd_bytes
else:
    grad = torch.randn(shape, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for it in range(a.iters):
        idx, off = reqs[it % len(reqs)]
        m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
    alg = bwd_bytes
torch.cuda.synchronize()
print(json.dumps({"workload": a.workload, "phase": a.phase, "iters": a.iters, "algorithmic_bytes_per_step": alg, "lookups_per_step": n,
                  "tables": T, "dtype": str(dt), "requests_rotated": len(reqs)}))
