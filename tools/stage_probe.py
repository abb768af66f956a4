#!/usr/bin/env python3
"""forward with / without the LDS-staged output burst (PARAM_AMD_FWD_STAGE is read once per process: run twice)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.indices import tbe_request
from param_amd.embedding_bag import _TableSet, _fwd
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
alg = T * B * L * (D * 4 + 8) + T * B * (D * 4 + 8)
bpb = int(os.environ.get("BPB", "0"))
unroll = int(os.environ.get("UNROLL", "0"))
param_amd.set_tuning(unroll=unroll, bags_per_block=bpb)
for layout in ("tbd",):
    ts = _TableSet([m.table(t) for t in range(T)], layout)
    out = torch.empty((T, B, D) if layout == "tbd" else (B, T * D), device=dev)
    for name, alpha in (("uniform", 0.0), ("zipf", 1.05)):
        idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=3)
        best = 1e9
        for rep in range(3):
            for _ in range(3): _fwd(ts, idx, off, B, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): _fwd(ts, idx, off, B, out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / 20)
        print(json.dumps({"stage": os.environ.get("PARAM_AMD_FWD_STAGE", "0"), "bpb": bpb, "unroll": unroll, "layout": layout, "indices": name, "ms": best * 1e3,
                          "alg_frac": alg / best / 8e12, "checksum": float(out.double().sum())}), flush=True)
