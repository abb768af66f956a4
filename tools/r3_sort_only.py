#!/usr/bin/env python3
"""stand-alone key sort at benchmark size (48 x 10 M, B 8192, L 20), uniform and Zipf indices: ms per sort (HIP events, 20 sorts)"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, B, L = 48, 10_000_000, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, 4, device=dev, init=None, fused_update=False)
for name, a, seed in (("uniform", 0.0, 2), ("zipf1.05", 1.05, 1)):
    idx, off = tbe_request([R] * T, B, L, a, device=dev, seed=seed)
    for _ in range(5):
        m.sort_indices(idx, off, batch=B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.sort_indices(idx, off, batch=B)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"indices": name, "sort_ms": e0.elapsed_time(e1) / 20, "lib": os.environ.get("PARAM_AMD_LIB", "default")}), flush=True)
