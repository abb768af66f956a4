#!/usr/bin/env python3
"""How fast can the chip read random 512-byte rows out of 245 GB at all?  The same 7.86 M uniform lookups per launch with
the per-bag work (offset staging, output writes) taken to the limit: pooling 20 / 100 / 1024 (163 840 lookups per table
in every case).  At pooling 1024 writes and staging are < 0.1 % of the bytes: what remains is the random-row read rate."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D = 48, 10_000_000, 128
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
for L, B in ((20, 8192), (100, 1638), (1024, 160)):
    idx, off = tbe_request([R] * T, B, L, 0.0, device=dev, seed=3)
    out = torch.empty((B, T * D), device=dev)
    for split in (False, True):
        if split and L < 1024:
            continue
        for _ in range(3): m.lookup(idx, off, out=out, batch=B, split_bags=split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.lookup(idx, off, out=out, batch=B, split_bags=split)
        e1.record(); torch.cuda.synchronize()
        s = e0.elapsed_time(e1) * 1e-3 / 20
        n = T * B * L
        print(json.dumps({"pooling": L, "bags_per_table": B, "split_kernel": split, "ms": round(s * 1e3, 4),
                          "row_read_TBps": round(n * 512 / s / 1e12, 3), "alg_frac": round((n * 520 + T * B * 520) / s / 8e12, 4)}), flush=True)
