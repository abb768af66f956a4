#!/usr/bin/env python3
"""rocprofv3 workload (round 2): the forward kernel on uniform indices over the full 245 GB footprint, (A) with row-by-row
output stores (round 1's kernel), (B) with the LDS-staged output burst, (C) burst + rows confined to 1/64 of every table;
1 warm-up + 3 measured launches each, in that order."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False, layout="tbd")
out = torch.empty((T, B, D), device=dev)
idx, off = tbe_request([R] * T, B, L, 0.0, device=dev, seed=2)
idc, ofc = tbe_request([R // 64] * T, B, L, 0.0, device=dev, seed=2)
for stage, (i, o) in ((0, (idx, off)), (1, (idx, off)), (1, (idc, ofc))):
    param_amd.set_forward_tuning(stage)
    for _ in range(4):
        m.lookup(i, o, out=out, batch=B)
    torch.cuda.synchronize()
param_amd.set_forward_tuning()
