#!/usr/bin/env python3
"""Round 5: the sorted apply's main kernel on the hybrid path's left-overs (uniform benchmark request: ~225 K flagged pairs) --
per-workgroup stamps from an EXPERIMENT build (csrc/pm_experiments.h): 0 start, 1 window staged (barrier), 2 chunk shape known,
3 lane group 0 done.   PARAM_AMD_LIB=build/libparam_amd_exp.so python tools/r5_apply_trace.py [--requests uniform,zipf1.05]"""
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
ap.add_argument("--requests", default="uniform")
ap.add_argument("--tables", type=int, default=48)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L_ = _lib.load()
fn = L_.pm_experiment_trace_apply_f32
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
T, R, D, B, L = a.tables, 10_000_000, 128, 8192, 20
m = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=torch.float32, device=dev, init="normal", layout="tbd", seed=1, fused_update=False)
grad = torch.randn((T, B, D), device=dev)
SLOTS, WGS = 8, 1 << 15
for rq in a.requests.split(","):
    alpha = 0.0 if rq == "uniform" else float(rq[4:])
    idx, off = tbe_request([R] * T, B, [L] * T, alpha=alpha, device=dev, seed=3)
    for _ in range(3):
        m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
    fn(None, 0, 1)
    m.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B)
    buf = np.zeros(WGS * SLOTS, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size, 0) == 0
    tr = buf.reshape(WGS, SLOTS).astype(np.int64)
    live = tr[(tr[:, 0] > 0) & (tr[:, 3] > 0)]
    early = int(((tr[:, 0] > 0) & (tr[:, 3] == 0)).sum())
    t0 = tr[tr[:, 0] > 0][:, 0].min()
    rel = (live[:, :4] - t0) * 0.01
    rec = {"request": rq, "working_wgs": int(live.shape[0]), "wgs_that_left_early": early, "span_us": round(float(rel[:, 3].max()), 2),
           "status": m.sort_status(idx, off, batch=B)}
    for s, nm in enumerate(["start", "staged", "shape", "done"]):
        rec[nm + "_us_min_p50_p90_max"] = [round(float(x), 2) for x in np.percentile(rel[:, s], [0, 50, 90, 100])]
    for s, nm in enumerate(["stage", "shape", "walk"]):
        rec[nm + "_len_us_p50_p90_max"] = [round(float(x), 2) for x in np.percentile(rel[:, s + 1] - rel[:, s], [50, 90, 100])]
    print(json.dumps(rec), flush=True)
