#!/usr/bin/env python3
"""Round 6: host time of ONE EmbeddingBagMI355 call (the reference's benchmark loop, pytorch_emb.py:48-69, is bound by it below batch
~2048: 14 M x 128, nnz 30).  Times the module call at batch 512 .. 4096 (async loop + one sync, as the driver does) and the pieces of
the call on their own (request key, out_desc, torch.empty, stream accessor, the ctypes call).  One JSON line."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd import _lib, embedding_bag as eb  # noqa: E402
from param_amd.indices import fixed_offsets, init_indices  # noqa: E402

dev = torch.device("cuda:0")
features, D, nnz = 14_000_000, 128, 30
emb = param_amd.EmbeddingBagMI355(features, D, mode="sum", device=dev)
emb.weight.requires_grad_(False)
rec = {}
for batch in (512, 1024, 2048, 4096):
    idx = init_indices(0.0, features, batch, nnz).to(dev)
    off = fixed_offsets(batch, nnz, device=dev)
    for _ in range(20):
        emb(idx, off)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        emb(idx, off)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    rec[f"batch{batch}"] = {"us_per_call_issue": round(t_issue / n * 1e6, 2), "us_per_step": round(t_all / n * 1e6, 2)}
ts = emb._tables()
B = off.numel()


def per(fn, n=20000):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return round((time.perf_counter() - t0) / n * 1e6, 3)


op = ts.request(idx, off, B, None, 0, None, forward=True)
out = torch.empty((B, D), device=dev)
L = _lib.load()
rec["pieces_us"] = {
    "request(cached)": per(lambda: ts.request(idx, off, B, None, 0, None, forward=True)),
    "out_desc": per(lambda: ts.out_desc(B)),
    "torch.empty": per(lambda: torch.empty((B, D), dtype=torch.float32, device=dev)),
    "stream_ptr": per(eb._stream_ptr),
    "out.data_ptr": per(out.data_ptr),
    "module.__call__ overhead (nn.Module)": None,
}
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3000
sp = eb._stream_ptr()
for _ in range(n):
    L.pm_embbag_fwd(ctypes.byref(op), out.data_ptr(), sp)
rec["pieces_us"]["ctypes pm_embbag_fwd (issue)"] = round((time.perf_counter() - t0) / n * 1e6, 3)
torch.cuda.synchronize()
print(json.dumps(rec))
