#!/usr/bin/env python3
"""What the quantised output burst of the forward buys at benchmark size (48 x 10 M x 128 fp32, batch 8192, pool 20):
  fp32      : lookup (the headline kernel)
  two-pass  : lookup + pm_rows_quantize on its output            (what a quantised exchange costs without the fusion)
  fused     : pm_embbag_fwd_quantized                            (the pooled fp32 rows never reach HBM)
for bit widths 16 / 8 / 4 and both index distributions.  One JSON line per case."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd import quant
from param_amd.indices import tbe_request
from param_amd.embedding_bag import _TableSet, _fwd, _fwd_quantized

dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
read_bytes = T * B * L * (D * 4 + 8) + T * B * 8


def timed(fn, iters=20, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / iters)
    return best


for layout in ("tbd", "bd"):
    ts = _TableSet([m.table(t) for t in range(T)], layout)
    out = torch.empty((T, B, D) if layout == "tbd" else (B, T * D), device=dev)
    for name, alpha in (("uniform", 0.0), ("zipf", 1.05)):
        idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=3)
        t32 = timed(lambda: _fwd(ts, idx, off, B, out=out))
        print(json.dumps({"layout": layout, "indices": name, "case": "fp32", "ms": round(t32 * 1e3, 4),
                          "alg_frac": round((read_bytes + T * B * D * 4) / t32 / 8e12, 4)}), flush=True)
        for bits in (16, 8, 4):
            rb = quant.host_row_bytes(D, bits)
            q = torch.empty(T * B * rb, dtype=torch.uint8, device=dev)
            t2 = timed(lambda: (_fwd(ts, idx, off, B, out=out), quant.quantize_rows(out, D, bits, out=q)))
            q2 = torch.empty_like(q)
            tf = timed(lambda: _fwd_quantized(ts, idx, off, B, bits, out=q2))
            same = bool(torch.equal(q, q2))
            print(json.dumps({"layout": layout, "indices": name, "case": f"bits{bits}", "two_pass_ms": round(t2 * 1e3, 4),
                              "fused_ms": round(tf * 1e3, 4), "fused_vs_fp32": round(tf / t32, 4),
                              "fused_alg_frac": round((read_bytes + T * B * rb) / tf / 8e12, 4), "bytes_equal": same}), flush=True)
