#!/usr/bin/env python3
"""single 14 M x 128 table (PARAM dataset A): per-call host overhead at tiny batches and tile size at large ones"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
dev = torch.device("cuda:0")
emb = param_amd.EmbeddingBagMI355(14_000_000, 128).to(dev)
emb.weight.requires_grad_(False)
def run(batch, nnz=30, steps=200):
    idx = torch.randint(0, 14_000_000, (batch * nnz,), device=dev)
    off = torch.arange(batch, device=dev) * nnz
    for _ in range(10): emb(idx, off)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): emb(idx, off)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6
for b in (8, 512, 2048):
    h, tot = run(b)
    print(json.dumps({"batch": b, "host_us_per_call": round(h, 2), "us_per_step": round(tot, 2)}), flush=True)
for bpb in (0, 8, 16, 32, 64):
    param_amd.set_tuning(0, bpb, -1, -1)
    for b in (16384, 65536):
        h, tot = run(b, steps=50)
        print(json.dumps({"bags_per_block": bpb, "batch": b, "us_per_step": round(tot, 2), "param_GBps": round(b * 30 * 512 / tot / 1e3, 1)}), flush=True)
