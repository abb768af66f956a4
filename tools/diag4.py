#!/usr/bin/env python3
"""Forward under RAGGED bags (DLRM default: bag sizes uniform in 1..2L-1, same mean as the fixed-L run)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
out = torch.empty((B, T * D), device=dev)
def timeit(fn, steps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps
g = torch.Generator(device=dev).manual_seed(1)
for name, lens in [("fixed20", torch.full((T * B,), L, device=dev)),
                   ("ragged_1_39", torch.randint(1, 2 * L, (T * B,), device=dev, generator=g)),
                   ("ragged_0_100_heavy_tail", (torch.rand(T * B, device=dev, generator=g).pow(4) * 100).long())]:
    off = torch.zeros(T * B + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=off[1:])
    n = int(off[-1])
    idx = torch.randint(0, R, (n,), device=dev, generator=g)
    for bpb in (0, 8):
        param_amd.set_tuning(0, bpb, -1, -1)
        s = timeit(lambda: m.lookup(idx, off, out=out, batch=B))
        alg = n * (D * 4 + 8) + T * B * (D * 4 + 8)
        print(json.dumps({"bags": name, "bpb": bpb, "lookups": n, "ms": s * 1e3, "Glookups_s": n / s / 1e9, "frac": alg / s / 8e12}), flush=True)
