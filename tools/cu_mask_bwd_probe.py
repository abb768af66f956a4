#!/usr/bin/env python3
"""Is the sorted backward's apply bound by the CUs (latency / occupancy) or by the memory system?  The same presorted
apply on HIP streams masked to fewer CUs (hipExtStreamCreateWithCUMask), uniform and Zipf indices."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
grad = torch.randn(B, T * D, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
bwd_bytes = T * B * L * (2 * D * 4 + 8) + T * B * (D * 4 + 8)

def masked_stream(n_cus):
    words = (ctypes.c_uint32 * 8)()
    for i in range(n_cus):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words) == 0
    return s.value

for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=3)
    m.sort_indices(idx, off, batch=B)
    torch.cuda.synchronize()
    for n_cus in (256, 224, 192, 160, 128, 64):
        st = torch.cuda.ExternalStream(masked_stream(n_cus), device=dev)
        with torch.cuda.stream(st):
            for _ in range(2):
                m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B, presorted=True)
            e1.record(st)
        st.synchronize()
        s = e0.elapsed_time(e1) * 1e-3 / 10
        print(json.dumps({"alpha": alpha, "cus": n_cus, "apply_ms": s * 1e3, "alg_frac": bwd_bytes / s / 8e12}), flush=True)
