#!/usr/bin/env python3
"""Round 4: forward with the XCD-contiguous tile order for table counts that are not a multiple of 8 (26 tables: BASELINE
configs[3]; the Criteo tables: configs[4]) against the plain order, one process, [B, sum D] output.  One JSON line each."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.compute.pt import dataset as ds  # noqa: E402
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

dev = torch.device("cuda:0")
B, D = 8192, 128
for name, rows, pools in (("26x10M", [10_000_000] * 26, [20] * 26), ("criteo", list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot))):
    m = param_amd.BatchedEmbeddingBagMI355(rows, D, device=dev, init="normal", layout="bd", seed=3, fused_update=False)
    out = torch.empty((B, len(rows) * D), device=dev)
    alg = sum(algorithmic_bytes(1, B, L, D, 4) for L in pools)
    reqs = {"uniform": tbe_request(rows, B, pools, 0.0, device=dev, seed=2), "zipf": tbe_request(rows, B, pools, 1.05, device=dev, seed=1)}
    for rnd in range(2):
        for xa in (0, -1):
            param_amd.set_tuning(xcd_affine=xa)
            for dist, (i, o) in reqs.items():
                for _ in range(15):
                    m.lookup(i, o, out=out, batch=B)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(40):
                    m.lookup(i, o, out=out, batch=B)
                e1.record()
                torch.cuda.synchronize()
                s = e0.elapsed_time(e1) * 1e-3 / 40
                print(json.dumps({"workload": name, "round": rnd, "xcd_order": "plain" if xa == 0 else "contiguous eighths", "indices": dist,
                                  "avg_launch_us": round(s * 1e6, 2), "G_lookups_per_s": round(B * sum(pools) / s / 1e9, 3),
                                  "alg_frac_of_8TBps": round(alg / s / 8e12, 4)}), flush=True)
    param_amd.set_tuning()
    del m, out
    torch.cuda.empty_cache()
