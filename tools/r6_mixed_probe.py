#!/usr/bin/env python3
"""Round 6: where the mixed-dim Criteo forward's time goes.  The 26 Criteo tables with dims by table size (dataset.criteo_v2_mixed_dims),
batch 8192: the forward of the whole request and of sub-requests (tables of one width only), with the per-table lane-group hint
(pm_embbag_batch.min_dim) and without, uniform and Zipf rows.  PARAM_AMD_FLAT_TARGET / _BAGS are read once per process: sweep them
by running the script again.  One JSON line per (subset, hint, request)."""
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
ap.add_argument("--subsets", default="all,128,64,32,16,narrow")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--backward", action="store_true")
ap.add_argument("--hybrid", default="", help="comma-separated pm_set_hybrid_tuning values for the backward, taking turns (default: the library's)")
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.batch
rows_all, pools_all = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot)
dims_all = ds.criteo_v2_mixed_dims(rows_all)


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        s = e0.elapsed_time(e1) * 1e-3 / n
        best = s if best is None else min(best, s)
    return best


for subset in a.subsets.split(","):
    if subset in ("all", "all128"):
        sel = list(range(26))
    elif subset == "narrow":
        sel = [t for t in range(26) if dims_all[t] < 128]
    else:
        sel = [t for t in range(26) if dims_all[t] == int(subset)]
    rows, pools, dims = [rows_all[t] for t in sel], [pools_all[t] for t in sel], [dims_all[t] for t in sel]
    if subset == "all128":
        dims = [128] * 26
    m = param_amd.BatchedEmbeddingBagMI355(rows, dims, dtype=torch.float32, device=dev, init="normal", layout="bd", seed=1, fused_update=False)
    out = torch.empty(B, sum(dims), device=dev)
    grad = torch.randn(B, sum(dims), device=dev)
    n = B * sum(pools)
    fwd_bytes = sum(B * L * (D * 4 + 8) + B * (8 + D * 4) for L, D in zip(pools, dims))
    bwd_bytes = sum(B * L * (2 * D * 4 + 8) + B * (D * 4 + 8) for L, D in zip(pools, dims))
    for name, alpha in (("uniform", 0.0), ("zipf1.05", 1.05)):
        reqs = [tbe_request(rows, B, pools, alpha=alpha, device=dev, seed=2 + 1000 * k) for k in range(4)]
        for hint in (1, 0):
            ts = m._tables()
            ts.min_dim = min(dims) if hint else 0
            ts._req = {}
            k = [0]

            def f():
                i, o = reqs[k[0] % 4]
                k[0] += 1
                m.lookup(i, o, out=out, batch=B)

            s = timed(f, a.iters)
            rec = {"exp": "mixed_fwd", "subset": subset, "tables": len(sel), "hint": hint, "indices": name, "lookups": n,
                   "us": round(s * 1e6, 2), "alg_frac": round(fwd_bytes / s / 8e12, 4), "alg_MB": round(fwd_bytes / 1e6, 1),
                   "flat_target": os.environ.get("PARAM_AMD_FLAT_TARGET"), "flat_bags": os.environ.get("PARAM_AMD_FLAT_BAGS"), "compact": os.environ.get("PARAM_AMD_FLAT_COMPACT")}
            print(json.dumps(rec), flush=True)
        if a.backward:
            i, o = reqs[0]
            for hyb in ([int(x) for x in a.hybrid.split(",")] if a.hybrid else [-1]):
                param_amd.set_hybrid_tuning(hyb)
                s = timed(lambda: m.scatter_add_(grad, i, o, alpha=-1e-6, batch=B), max(10, a.iters // 2))
                st = m.sort_status(i, o, batch=B)
                print(json.dumps({"exp": "mixed_bwd", "subset": subset, "tables": len(sel), "indices": name, "lookups": n, "hybrid": hyb,
                                  "us": round(s * 1e6, 2), "alg_frac": round(bwd_bytes / s / 8e12, 4), "alg_MB": round(bwd_bytes / 1e6, 1), **st}), flush=True)
            param_amd.set_hybrid_tuning()
    del m, out, grad
    torch.cuda.empty_cache()
