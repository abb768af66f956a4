#!/usr/bin/env python3
"""forward: output layout [B, T*D] (bd, the all-to-all send layout) vs [T, B, D] (tbd) -- how much do the strided
512-byte output pieces of the bd layout cost?"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes
from param_amd.embedding_bag import _TableSet, _fwd
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
alg = algorithmic_bytes(T, B, L, D, 4)
for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=3)
    for rep in range(2):
        for layout in ("bd", "tbd"):
            ts = _TableSet([m.table(t) for t in range(T)], layout)
            out = torch.empty((B, T * D) if layout == "bd" else (T, B, D), device=dev)
            for _ in range(5): _fwd(ts, idx, off, B, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): _fwd(ts, idx, off, B, out=out)
            e1.record(); torch.cuda.synchronize()
            s = e0.elapsed_time(e1) * 1e-3 / 30
            print(json.dumps({"alpha": alpha, "layout": layout, "ms": round(s * 1e3, 4), "frac": round(alg / s / 8e12, 4)}), flush=True)
