#!/usr/bin/env python3
"""Where is the forward's ceiling -- in the CUs or in the memory system?  The same request is launched on HIP streams
restricted to 256 / 192 / 128 / 64 / 32 CUs (hipExtStreamCreateWithCUMask).  If achieved bandwidth falls in proportion
to the CU count the limit is per-CU (outstanding-miss capacity x latency); if it stays flat down to some count the
memory system (HBM / fabric / translation) is the limit and more in-flight loads per CU cannot help."""
import argparse, ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd import _lib
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes
from param_amd.embedding_bag import _TableSet, _fwd
from param_amd.indices import tbe_request

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=48)
ap.add_argument("--rows", type=int, default=10_000_000)
a = ap.parse_args()
dev = torch.device("cuda:0")
T, R, D, B, L = a.tables, a.rows, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
hip = ctypes.CDLL("libamdhip64.so")
lib = _lib.load()


def masked_stream(n_cus):
    words = (ctypes.c_uint32 * 8)()
    for i in range(n_cus):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return s.value


ts = _TableSet([m.table(t) for t in range(T)], "bd")
out = torch.empty((B, T * D), device=dev)
alg = algorithmic_bytes(T, B, L, D, 4)
for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=3)
    for n_cus in (256, 224, 192, 160, 128, 96, 64, 32):
        raw = masked_stream(n_cus)
        st = torch.cuda.ExternalStream(raw, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for _ in range(3):
                _fwd(ts, idx, off, B, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                _fwd(ts, idx, off, B, out=out)
            e1.record(st)
        st.synchronize()
        s = e0.elapsed_time(e1) * 1e-3 / 20
        print(json.dumps({"alpha": alpha, "cus": n_cus, "ms": s * 1e3, "alg_TBps": alg / s / 1e12, "frac": alg / s / 8e12,
                          "GBps_per_cu": alg / s / 1e9 / n_cus}), flush=True)
        hip.hipStreamDestroy(ctypes.c_void_p(raw))
