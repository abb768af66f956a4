#!/usr/bin/env python3
"""Round 5: the lookup of ONE rank of an 8-GPU table-wise sharded exchange on one GPU -- 8 tables x 10 M x 128 fp32, global batch
8 x 8192, L = 20 -- writing its send buffer in the three layouts: [B, sum D] (what the exchange used so far), [T, B, D] (no exchange
can send it: a peer's chunk is not contiguous) and the blocked [W][T][B_local][D]; forward and backward (sort + apply of the
gradient in the same layout), uniform and Zipf.  One JSON line per (layout, distribution)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import param_amd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

dev = torch.device("cuda", 0)
T, R, D, Bl, W, L = int(os.environ.get("PROBE_TABLES", "8")), 10_000_000, 128, 8192, 8, 20
B = Bl * W
rows = [R] * T
models = {lay: param_amd.BatchedEmbeddingBagMI355(rows, D, device=dev, init="normal", layout=lay, seed=7, fused_update=False,
                                                  block_bags=Bl if lay == "blocked" else None) for lay in ("bd", "tbd", "blocked")}
n = T * B * L
fwd_bytes = n * (D * 4 + 8) + T * B * (8 + D * 4)
bwd_bytes = n * (2 * D * 4 + 8) + T * B * (D * 4 + 8)


def timed(fn, iters=30, warm=8):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for rnd in range(2):
    for dist, alpha in (("uniform", 0.0), ("zipf", 1.05)):
        idx, off = tbe_request(rows, B, L, alpha=alpha, device=dev, seed=5)
        for lay, m in models.items():
            shape = {"bd": (B, T * D), "tbd": (T, B, D), "blocked": (W, T, Bl, D)}[lay]
            out = torch.empty(shape, device=dev)
            grad = torch.randn(shape, device=dev)
            f = timed(lambda: m.lookup(idx, off, out=out, batch=B))
            b = timed(lambda: m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B), iters=15, warm=5)
            print(json.dumps({"round": rnd, "layout": lay, "indices": dist, "tables": T, "global_batch": B, "fwd_us": round(f * 1e6, 1),
                              "fwd_alg_frac": round(fwd_bytes / f / 8e12, 4), "bwd_us": round(b * 1e6, 1), "bwd_alg_frac": round(bwd_bytes / b / 8e12, 4)}), flush=True)
            del out, grad
