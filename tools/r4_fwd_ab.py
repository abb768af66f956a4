#!/usr/bin/env python3
"""Forward A/B of several builds of libparam_amd.so INSIDE ONE PROCESS: same tables, same allocation, same box, the libraries
taking turns (round 4: where did the forward's uniform-index fraction go between visit v45 and the end of round 3?).

    python tools/r4_fwd_ab.py [--rounds 3] [--tables 48] label=path.so label=path.so ...

Per (library, layout, index distribution): average launch time of 30 launches after 10 warm-ups, one JSON line each.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import param_amd  # noqa: E402
from param_amd import _lib  # noqa: E402
from param_amd.embedding_bag import _TableSet, _fwd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402


def swap(path: str) -> None:
    _lib._lib = None
    _lib.LIB_PATH = path
    _lib.load()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--tables", type=int, default=48)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--env", default="", help="comma-separated KEY=VAL applied before the first library loads")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    for kv in filter(None, a.env.split(",")):
        k, v = kv.split("=", 1)
        os.environ[k] = v
    libs = [x.split("=", 1) for x in a.libs]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
    T, R, D, B, L = a.tables, a.rows, 128, 8192, 20
    swap(libs[0][1])
    model = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", layout="tbd", seed=1000, fused_update=False)
    req = {"zipf": tbe_request([R] * T, B, [L] * T, alpha=1.05, device=dev, seed=1),
           "uniform": tbe_request([R] * T, B, [L] * T, alpha=0.0, device=dev, seed=2)}
    esize = 4 if dt == torch.float32 else 2
    alg = T * B * L * (D * esize + 8) + T * B * (8 + D * 4)
    sets = {lay: _TableSet([model.table(t) for t in range(T)], lay) for lay in ("tbd", "bd")}
    outs = {"tbd": torch.empty((T, B, D), dtype=torch.float32, device=dev), "bd": torch.empty((B, T * D), dtype=torch.float32, device=dev)}
    for rnd in range(a.rounds):
        for label, path in libs:
            swap(path)
            for lay in ("tbd", "bd"):
                for dist in ("uniform", "zipf"):
                    i, o = req[dist]
                    for _ in range(10):
                        _fwd(sets[lay], i, o, B, out=outs[lay])
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.iters):
                        _fwd(sets[lay], i, o, B, out=outs[lay])
                    e1.record()
                    torch.cuda.synchronize()
                    s = e0.elapsed_time(e1) * 1e-3 / a.iters
                    print(json.dumps({"round": rnd, "lib": label, "layout": lay, "indices": dist, "avg_launch_us": round(s * 1e6, 2),
                                      "G_lookups_per_s": round(T * B * L / s / 1e9, 3), "alg_frac_of_8TBps": round(alg / s / 8e12, 4),
                                      "env": a.env}), flush=True)


if __name__ == "__main__":
    main()
