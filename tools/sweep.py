#!/usr/bin/env python3
"""Tuning sweep of the forward (and backward) kernel on one GPU: allocate the tables once,
then time launch variants.  Prints one JSON line per variant (to stdout / --out)."""
import argparse, itertools, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes
from param_amd.indices import tbe_request

p = argparse.ArgumentParser()
p.add_argument("--tables", type=int, default=48)
p.add_argument("--rows", type=int, default=10_000_000)
p.add_argument("--dim", type=int, default=128)
p.add_argument("--batch", type=int, default=8192)
p.add_argument("--pooling", type=int, default=20)
p.add_argument("--dtype", default="fp32")
p.add_argument("--alphas", default="0,1.05")
p.add_argument("--unrolls", default="2,4,8")
p.add_argument("--bpbs", default="0,16,64,128")
p.add_argument("--xcds", default="0,1")
p.add_argument("--nts", default="0,1")
p.add_argument("--steps", type=int, default=20)
p.add_argument("--bwd", action="store_true")
p.add_argument("--out", default="")
a = p.parse_args()
dev = torch.device("cuda:0")
dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
es = torch.empty(0, dtype=dt).element_size()
T, R, D, B, L = a.tables, a.rows, a.dim, a.batch, a.pooling
t0 = time.time()
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", seed=1, fused_update=False)
torch.cuda.synchronize()
print(f"# alloc+init {T}x{R}x{D} {a.dtype}: {time.time()-t0:.1f}s, free={torch.cuda.mem_get_info()[0]/2**30:.1f} GiB", flush=True)
out = torch.empty((B, T * D), dtype=torch.float32, device=dev)
alg = algorithmic_bytes(T, B, L, D, es)
fh = open(a.out, "a") if a.out else None

def timeit(fn, steps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps

for alpha in [float(x) for x in a.alphas.split(",")]:
    t0 = time.time()
    idx, off = tbe_request([R] * T, B, L, alpha=alpha, device=dev, seed=1)
    torch.cuda.synchronize()
    print(f"# indices alpha={alpha}: {time.time()-t0:.1f}s", flush=True)
    for unroll, bpb, xcd, nt in itertools.product([int(x) for x in a.unrolls.split(",")], [int(x) for x in a.bpbs.split(",")],
                                                  [int(x) for x in a.xcds.split(",")], [int(x) for x in a.nts.split(",")]):
        param_amd.set_tuning(unroll, bpb, xcd, nt)
        s = timeit(lambda: m.lookup(idx, off, out=out, batch=B), a.steps)
        rec = {"op": "fwd", "alpha": alpha, "unroll": unroll, "bpb": bpb, "xcd": xcd, "nt": nt, "ms": s * 1e3,
               "Glookups_s": T * B * L / s / 1e9, "alg_GBps": alg / s / 1e9, "frac": alg / s / 1e9 / 8000}
        line = json.dumps(rec); print(line, flush=True)
        if fh: fh.write(line + "\n"); fh.flush()
    if a.bwd:
        grad = torch.randn((B, T * D), device=dev)
        bb = T * B * L * (2 * D * es + 8) + T * B * (D * 4 + 8)
        for bpb, xcd in itertools.product([0, 64], [0, 1]):
            param_amd.set_tuning(0, bpb, xcd, 0)
            s = timeit(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B), max(5, a.steps // 2))
            rec = {"op": "bwd", "alpha": alpha, "bpb": bpb, "xcd": xcd, "ms": s * 1e3, "Glookups_s": T * B * L / s / 1e9,
                   "alg_GBps": bb / s / 1e9, "frac": bb / s / 1e9 / 8000}
            line = json.dumps(rec); print(line, flush=True)
            if fh: fh.write(line + "\n"); fh.flush()
        del grad
