#!/usr/bin/env python3
"""Round 6: hyb_mark_kernel's phases per workgroup from an EXPERIMENT build (csrc/pm_experiments.h): 0 start, 1 verdict known, 2 maps zeroed,
3 wave 0's scan of the table's lookups done, 4 every wave's, 5 dup words written (drained).
  make -C param_amd/csrc EXTRA=-DPM_EXPERIMENTS OBJDIR=$PWD/build/csrc_exp OUT=$PWD/build/libparam_amd_exp.so
  PARAM_AMD_LIB=build/libparam_amd_exp.so python tools/r6_mark_trace.py [--dtype bf16 --tables 64]"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import param_amd  # noqa: E402
from param_amd import _lib  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=48)
ap.add_argument("--dtype", default="fp32")
a = ap.parse_args()
dev = torch.device("cuda", 0)
L_ = _lib.load()
fn = L_.pm_experiment_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
T, R, D, B, L = a.tables, 10_000_000, 128, 8192, 20
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", layout="tbd", seed=1, fused_update=False)
grad = torch.randn((T, B, D), device=dev)
SLOTS, WGS, BASE = 8, 1 << 15, 24576
idx, off = tbe_request([R] * T, B, [L] * T, alpha=0.0, device=dev, seed=3)
for _ in range(3):
    m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
fn(None, 0, 1)
m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
buf = np.zeros(WGS * SLOTS, dtype=np.uint64)
assert fn(buf.ctypes.data, buf.size, 0) == 0
tr = buf.reshape(WGS, SLOTS).astype(np.int64)[BASE:]
live = tr[(tr[:, 0] > 0) & (tr[:, 5] > 0)]
t0 = live[:, 0].min()
rel = (live[:, :6] - t0) * 0.01
names = ["start", "verdict", "zeroed", "wave0_scanned", "all_scanned", "written"]
rec = {"dtype": a.dtype, "tables": T, "working_wgs": int(live.shape[0]), "span_us": round(float(rel[:, 5].max()), 2)}
for s, nm in enumerate(names):
    rec[nm + "_at_us_min_p50_p90_max"] = [round(float(x), 2) for x in np.percentile(rel[:, s], [0, 50, 90, 100])]
for s in range(5):
    rec[names[s + 1] + "_len_us_p50_p90_max"] = [round(float(x), 2) for x in np.percentile(rel[:, s + 1] - rel[:, s], [50, 90, 100])]
print(json.dumps(rec), flush=True)
