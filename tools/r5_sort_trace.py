#!/usr/bin/env python3
"""Round 5: where a look-back pass of the key sort spends its time -- per-workgroup phase timestamps from an EXPERIMENT build
(csrc/pm_experiments.h; `make -C param_amd/csrc EXTRA=-DPM_EXPERIMENTS OBJDIR=$PWD/build/csrc_exp OUT=$PWD/build/libparam_amd_exp.so`).

    PARAM_AMD_LIB=build/libparam_amd_exp.so python tools/r5_sort_trace.py [--workload criteo|tables] [--requests uniform,zipf1.05]

Stamps of seg_lookback_pass_kernel (100 MHz clock, 10 ns): 0 start, 1 tile's pairs loaded, 2 ranked (digit counts of the tile known),
3 digit starts scanned + counts published, 4 pairs placed in LDS, 5 walk over the predecessors done, 6 barrier behind it, 7 pairs scattered.
Per pass: the launch's span, and percentiles of every phase's length and of every stamp's offset from the launch's first stamp.
"""
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
ap.add_argument("--workload", default="tables")
ap.add_argument("--tables", type=int, default=48)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L_ = _lib.load()
assert hasattr(L_, "pm_experiment_trace"), "run with PARAM_AMD_LIB=build/libparam_amd_exp.so (an experiment build)"
L_.pm_experiment_trace.restype = ctypes.c_int
L_.pm_experiment_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
if a.workload == "criteo":
    from param_amd.compute.pt import dataset as ds
    rows, pools = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot)
else:
    rows, pools = [10_000_000] * a.tables, [20] * a.tables
B = 8192
m = param_amd.BatchedEmbeddingBagMI355(rows, 8, dtype=torch.float32, device=dev, init="normal", layout="tbd", seed=1, fused_update=False)
SLOTS, WGS = 8, 1 << 15
names = ["load", "rank", "starts+publish", "place", "walk", "barrier", "scatter"]
for rq in a.requests.split(","):
    alpha = 0.0 if rq == "uniform" else float(rq[4:])
    idx, off = tbe_request(rows, B, pools, alpha=alpha, device=dev, seed=3)
    for _ in range(3):
        m.sort_indices(idx, off, batch=B)
    L_.pm_experiment_trace(None, 0, 1)
    m.sort_indices(idx, off, batch=B)
    buf = np.zeros(WGS * SLOTS, dtype=np.uint64)
    rc = L_.pm_experiment_trace(buf.ctypes.data, buf.size, 0)
    assert rc == 0, rc
    tr = buf.reshape(WGS, SLOTS).astype(np.int64)
    for p in range(8):
        blk = tr[p * 4096:(p + 1) * 4096]
        live = blk[blk[:, 0] > 0]
        if live.shape[0] == 0:
            continue
        t0 = live[:, 0].min()
        rel = (live - t0) * 0.01                      # us
        rec = {"request": rq, "workload": a.workload, "pass": p, "tiles": int(live.shape[0]), "span_us": round(float(rel[:, 7].max()), 2)}
        for s in range(8):
            q = np.percentile(rel[:, s], [0, 50, 90, 100])
            rec[f"stamp{s}_us_min_p50_p90_max"] = [round(float(x), 2) for x in q]
        for s in range(7):
            d = rel[:, s + 1] - rel[:, s]
            q = np.percentile(d, [50, 90, 100])
            rec[f"{names[s]}_us_p50_p90_max"] = [round(float(x), 2) for x in q]
        print(json.dumps(rec), flush=True)
