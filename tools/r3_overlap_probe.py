#!/usr/bin/env python3
"""fwd + bwd step at benchmark size with the key sort on a second HIP stream under the lookup: stream priorities (0 default, -1 high)
for the sort's stream -- does a latency-bound chain of small kernels get through next to a chip-filling lookup?"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
grad = torch.randn((B, T * D), device=dev)
out = torch.empty((T, B, D), device=dev) if False else None
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, a, seed in (("zipf1.05", 1.05, 1), ("uniform", 0.0, 2)):
    idx, off = tbe_request([R] * T, B, L, a, device=dev, seed=seed)
    o = m.lookup(idx, off, batch=B)
    def serial():
        m.lookup(idx, off, out=o, batch=B)
        m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
    res = {"indices": name, "serial_ms": timed(serial)}
    for pr in (0, -1):
        side = torch.cuda.Stream(device=dev, priority=pr)
        ev = torch.cuda.Event()
        def aside():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                m.sort_indices(idx, off, batch=B)
                ev.record(side)
            m.lookup(idx, off, out=o, batch=B)
            main.wait_event(ev)
            m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True)
        res[f"sort_aside_prio{pr}_ms"] = timed(aside)
        def aside_after():      # lookup launched first, then the sort on the side stream
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            m.lookup(idx, off, out=o, batch=B)
            with torch.cuda.stream(side):
                m.sort_indices(idx, off, batch=B)
                ev.record(side)
            main.wait_event(ev)
            m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True)
        res[f"lookup_first_prio{pr}_ms"] = timed(aside_after)
    print(json.dumps(res), flush=True)
