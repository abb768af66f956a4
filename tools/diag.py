#!/usr/bin/env python3
"""Diagnostics on one GPU: where does the forward's bandwidth ceiling come from?
 (a) footprint scaling (tables in play 8..56): TLB / page-walk reach
 (b) sequential rows (streaming) vs random rows
 (c) batch sweep (dataset.py batches) and pooling sweep
Prints JSON lines."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes
from param_amd.embedding_bag import _TableSet, _fwd
from param_amd.indices import tbe_request

p = argparse.ArgumentParser()
p.add_argument("--tables", type=int, default=56)
p.add_argument("--rows", type=int, default=10_000_000)
p.add_argument("--dim", type=int, default=128)
p.add_argument("--dtype", default="fp32")
p.add_argument("--steps", type=int, default=20)
p.add_argument("--skip", default="")
a = p.parse_args()
dev = torch.device("cuda:0")
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
es = 4 if dt == torch.float32 else 2
T, R, D = a.tables, a.rows, a.dim
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", seed=1, fused_update=False)

def timeit(fn, steps=a.steps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps

def emit(**kw):
    print(json.dumps(kw), flush=True)

def run_fwd(tag, Tn, B, L, idx, off, **extra):
    ts = _TableSet([m.table(t) for t in range(Tn)], "bd")
    out = torch.empty((B, Tn * D), dtype=torch.float32, device=dev)
    s = timeit(lambda: _fwd(ts, idx, off, B, out=out))
    alg = algorithmic_bytes(Tn, B, L, D, es)
    emit(test=tag, tables=Tn, batch=B, pooling=L, ms=s * 1e3, Glookups_s=Tn * B * L / s / 1e9,
         alg_GBps=alg / s / 1e9, frac=alg / s / 1e9 / 8000, **extra)

B, L = 8192, 20
if "a" not in a.skip:
    for Tn in (8, 16, 32, 48, T):
        if Tn > T: continue
        idx, off = tbe_request([R] * Tn, B, L, 0.0, device=dev, seed=2)
        run_fwd("footprint_uniform", Tn, B, L, idx, off, footprint_GB=Tn * R * D * es / 1e9)
    # same lookups count but confined to the first 1/64 of each table (TLB-friendly, still > MALL)
    Tn = min(48, T)
    idx, off = tbe_request([R // 64] * Tn, B, L, 0.0, device=dev, seed=2)
    run_fwd("confined_rows_1_64", Tn, B, L, idx, off, footprint_GB=Tn * (R // 64) * D * es / 1e9)
if "b" not in a.skip:
    Tn = min(48, T)
    n = Tn * B * L
    seq = (torch.arange(n, device=dev) % (B * L)) + 12345  # each table: B*L consecutive rows
    off = torch.arange(Tn * B + 1, device=dev) * L
    run_fwd("sequential_rows", Tn, B, L, seq, off)
    perm = torch.cat([torch.randperm(B * L, device=dev) + 12345 for _ in range(Tn)])
    run_fwd("permuted_dense_rows", Tn, B, L, perm, off)
if "c" not in a.skip:
    Tn = min(48, T)
    for Bc in (512, 2048, 8192, 32768):
        idx, off = tbe_request([R] * Tn, Bc, L, 0.0, device=dev, seed=2)
        run_fwd("batch_sweep_uniform", Tn, Bc, L, idx, off)
    for Lc in (1, 5, 30, 100):
        idx, off = tbe_request([R] * Tn, B, Lc, 0.0, device=dev, seed=2)
        run_fwd("pooling_sweep_uniform", Tn, B, Lc, idx, off)
    idx, off = tbe_request([R] * Tn, B, L, 0.0, device=dev, seed=2, index_dtype=torch.int32)
    run_fwd("int32_indices_uniform", Tn, B, L, idx, off)
