#!/usr/bin/env python3
"""What do the forward's output writes cost?  The benchmark launch with the output row stride set to 0: every bag of a
table writes the same 512 bytes (garbage result; the writes merge in L2 and never reach HBM), everything else unchanged."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from param_amd import _lib
from param_amd.embedding_bag import _TableSet, _stream_ptr
from param_amd.indices import tbe_request
dev = torch.device("cuda:0")
T, R, D, B, L = 48, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, device=dev, init="normal", seed=1, fused_update=False)
ts = _TableSet([m.table(t) for t in range(T)], "bd")
lib = _lib.load()
out = torch.empty((B, T * D), device=dev)
for alpha in (0.0, 1.05):
    idx, off = tbe_request([R] * T, B, L, alpha, device=dev, seed=3)
    for stride in (T * D, 0, T * D, 0):
        op = ts._build_request(idx, off, B, None, 0, None)
        op.out_stride = stride
        call = lambda: _lib.check(lib.pm_embbag_fwd(ctypes.byref(op), out.data_ptr(), _stream_ptr()))
        for _ in range(5): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): call()
        e1.record(); torch.cuda.synchronize()
        s = e0.elapsed_time(e1) * 1e-3 / 30
        n = T * B * L
        print(json.dumps({"alpha": alpha, "writes_reach_hbm": stride != 0, "ms": round(s * 1e3, 4), "row_read_TBps": round(n * 512 / s / 1e12, 3)}), flush=True)
