#!/usr/bin/env python3
"""Experiment: do PHYSICALLY CONTIGUOUS tables -- hipExtMallocWithFlags(hipDeviceMallocContiguous) --
let amdgpu map the tables with translation fragments larger than 2 MiB and lift the translation-bound forward
(profiles/r01_pmc_translation.md)?  The same request is timed on tables allocated (a) by torch / hipMalloc as one slab,
(b) one hipExtMallocWithFlags(Contiguous) allocation per table, (c) one contiguous allocation for everything.
Prints the amdgpu VM module parameters of the box first (fragment size / block size decide what the driver can do)."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from param_amd import _lib  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=16)
ap.add_argument("--rows", type=int, default=10_000_000)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
L = _lib.load()
T, R, D, B, Lp = a.tables, a.rows, 128, 8192, 20
table_bytes = R * D * 4

params = {}
for name in ("vm_fragment_size", "vm_block_size", "vm_size", "vm_update_mode", "vramlimit", "mes", "sched_policy"):
    try:
        params[name] = open(f"/sys/module/amdgpu/parameters/{name}").read().strip()
    except OSError:
        params[name] = None
print(json.dumps({"amdgpu_parameters": params}), flush=True)


def contiguous(nbytes):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(0x4))  # hipDeviceMallocContiguous
    if rc != 0:
        raise RuntimeError(f"hipExtMallocWithFlags(Contiguous, {nbytes / 2**30:.1f} GiB): hipError {rc}")
    return p.value


def run(table_ptrs, tag, extra):
    ptrs = torch.tensor(table_ptrs, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for t, p in enumerate(table_ptrs):
        _lib.check(L.pm_fill_random(p, R * D, _lib.PM_F32, 1, 0.0, 1.0, 1000 + t, stream))
    rows = torch.tensor([R] * T, dtype=torch.int64, device=dev)
    dims = torch.tensor([D] * T, dtype=torch.int32, device=dev)
    col0 = torch.arange(T, dtype=torch.int64, device=dev) * D
    out = torch.empty((B, T * D), device=dev)
    res = {}
    for alpha in (0.0, 1.05):
        idx, off = tbe_request([R] * T, B, Lp, alpha, device=dev, seed=3)
        op = _lib.pm_embbag_batch()
        op.num_tables, op.weight_dtype, op.index_dtype, op.max_dim = T, _lib.PM_F32, _lib.PM_I64, D
        op.batch, op.num_indices, op.bag_begin, op.bag_count = B, idx.numel(), 0, B
        op.tables, op.rows, op.dims, op.out_offsets = ptrs.data_ptr(), rows.data_ptr(), dims.data_ptr(), col0.data_ptr()
        op.out_stride, op.indices, op.offsets, op.per_sample_weights = T * D, idx.data_ptr(), off.data_ptr(), None

        def fn():
            _lib.check(L.pm_embbag_fwd(ctypes.byref(op), out.data_ptr(), stream))
        best = None
        for _rep in range(3):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            s = e0.elapsed_time(e1) * 1e-3 / 20
            best = s if best is None else min(best, s)
        alg = T * B * Lp * (D * 4 + 8) + T * B * (D * 4 + 8)
        res[f"alpha{alpha}"] = {"ms": best * 1e3, "alg_GBps": alg / best / 1e9, "frac": alg / best / 8e12}
    print(json.dumps({"alloc": tag, **extra, "tables": T, "GB": T * table_bytes / 1e9, **res}), flush=True)


slab = torch.empty(T * table_bytes, dtype=torch.uint8, device=dev)
run([slab.data_ptr() + t * table_bytes for t in range(T)], "torch/hipMalloc slab",
    {"va_mod_1GiB_MiB": (slab.data_ptr() % (1 << 30)) >> 20})
del slab
torch.cuda.empty_cache()

try:
    ptrs = [contiguous(table_bytes) for _ in range(T)]
    run(ptrs, "hipExtMallocWithFlags(Contiguous) per table",
        {"va_mod_1GiB_MiB": [(p % (1 << 30)) >> 20 for p in ptrs[:4]], "va_mod_2MiB": [p % (1 << 21) for p in ptrs[:4]]})
    for p in ptrs:
        hip.hipFree(ctypes.c_void_p(p))
except Exception as e:
    print(json.dumps({"alloc": "contiguous per table", "error": str(e)}), flush=True)

try:
    p = contiguous(T * table_bytes)
    run([p + t * table_bytes for t in range(T)], "hipExtMallocWithFlags(Contiguous) one slab", {"va_mod_1GiB_MiB": (p % (1 << 30)) >> 20})
    hip.hipFree(ctypes.c_void_p(p))
except Exception as e:
    print(json.dumps({"alloc": "contiguous one slab", "error": str(e)}), flush=True)
