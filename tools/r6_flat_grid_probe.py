#!/usr/bin/env python3
"""Round 6: launch shape of the flat-walk forward, IN ONE PROCESS (pm_set_forward_tuning(flat_grid) values taking turns: between processes
the same launch differs by up to 7 % on one box -- where the 100 GB of tables land physically).  Criteo tables, all D = 128 and mixed
dims, batch 8192, four requests rotating.  One JSON line per (workload, indices, grid, round)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.compute.pt import dataset as ds  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grids", default="0,1,1024,1536,2048,3072,4096,6144,8192")
ap.add_argument("--workloads", default="all128,mixed")
ap.add_argument("--targets", default="", help="lookups per tile (pm_set_forward_tuning(flat_target)) taking turns at the library's grid, instead of the grids")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--batch", type=int, default=8192)
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.batch
rows, pools = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot)


def timed(fn, n):
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


for wl in a.workloads.split(","):
    dims = [128] * 26 if wl == "all128" else ds.criteo_v2_mixed_dims(rows)
    m = param_amd.BatchedEmbeddingBagMI355(rows, dims, dtype=torch.float32, device=dev, init="normal", layout="bd", seed=1, fused_update=False)
    out = torch.empty(B, sum(dims), device=dev)
    fwd_bytes = sum(B * L * (D * 4 + 8) + B * (8 + D * 4) for L, D in zip(pools, dims))
    for name, alpha in (("uniform", 0.0), ("zipf1.05", 1.05)):
        reqs = [tbe_request(rows, B, pools, alpha=alpha, device=dev, seed=2 + 1000 * k) for k in range(4)]
        k = [0]

        def f():
            i, o = reqs[k[0] % 4]
            k[0] += 1
            m.lookup(i, o, out=out, batch=B)

        for rnd in range(a.rounds):
            for val in [int(x) for x in (a.targets or a.grids).split(",")]:
                grid, target = (1, val) if a.targets else (val, -1)
                param_amd.set_forward_tuning(flat_grid=grid, flat_target=target)
                s = timed(f, a.iters)
                print(json.dumps({"exp": "flat_grid", "workload": wl, "indices": name, "grid": grid, "target": target, "round": rnd, "us": round(s * 1e6, 2),
                                  "alg_frac": round(fwd_bytes / s / 8e12, 4)}), flush=True)
    param_amd.set_forward_tuning()
    del m, out
    torch.cuda.empty_cache()
