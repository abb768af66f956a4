#!/usr/bin/env python3
"""Round 5: the persistent forward against the classic one, configurations taking turns INSIDE ONE PROCESS on one box (same
tables, same allocation).  Each configuration's output is first compared bit for bit with the classic kernel's.

    python tools/r5_fwd_ab.py [--tables 48] [--dtype fp32] [--rounds 2] [--configs "classic;2,3,1,4,0;2,2,2,4,0"] [--unroll 0]

A configuration is ``classic`` or ``mode,slots,bags_per_group,pool_waves,wgs_per_cu`` (pm_set_forward_persist).  One JSON line per
(round, configuration, layout, index distribution): average launch time of --iters launches after 10 warm-ups (HIP events on
the launch stream), lookups/s, algorithmic fraction of 8 TB/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import param_amd  # noqa: E402
from param_amd.embedding_bag import _TableSet, _fwd  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402


def apply(cfg: str) -> None:
    if cfg == "classic":
        param_amd.set_forward_persist(0)
    else:
        param_amd.set_forward_persist(*[int(x) for x in cfg.split(",")])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--tables", type=int, default=48)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--pooling", type=int, default=20)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--layouts", default="tbd,bd")
    ap.add_argument("--dists", default="uniform,zipf")
    ap.add_argument("--configs", default="classic;2,3,1,4,0;2,4,1,4,0;2,2,2,4,0;2,3,2,4,0;2,2,4,4,0;2,3,1,7,0;2,2,2,7,0;2,3,2,7,0")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
    T, R, D, B, L = a.tables, a.rows, 128, a.batch, a.pooling
    param_amd.set_tuning(unroll=a.unroll)
    model = param_amd.BatchedEmbeddingBagMI355([R] * T, D, dtype=dt, device=dev, init="normal", layout="tbd", seed=1000, fused_update=False)
    req = {"zipf": tbe_request([R] * T, B, [L] * T, alpha=1.05, device=dev, seed=1),
           "uniform": tbe_request([R] * T, B, [L] * T, alpha=0.0, device=dev, seed=2)}
    esize = 4 if dt == torch.float32 else 2
    alg = T * B * L * (D * esize + 8) + T * B * (8 + D * 4)
    lays = a.layouts.split(",")
    dists = a.dists.split(",")
    sets = {lay: _TableSet([model.table(t) for t in range(T)], lay) for lay in lays}
    shape = {"tbd": (T, B, D), "bd": (B, T * D)}
    outs = {lay: torch.empty(shape[lay], dtype=torch.float32, device=dev) for lay in lays}
    cfgs = a.configs.split(";")
    # bit-identity of every configuration with the classic kernel, before anything is timed
    refs = {}
    apply("classic")
    for lay in lays:
        for dist in dists:
            i, o = req[dist]
            refs[lay, dist] = _fwd(sets[lay], i, o, B).clone()
    for cfg in cfgs:
        apply(cfg)
        for lay in lays:
            for dist in dists:
                i, o = req[dist]
                outs[lay].fill_(float("nan"))
                _fwd(sets[lay], i, o, B, out=outs[lay])
                torch.cuda.synchronize()
                same = bool(torch.equal(outs[lay], refs[lay, dist]))
                print(json.dumps({"check": cfg, "layout": lay, "indices": dist, "bit_identical_to_classic": same}), flush=True)
                if not same:
                    print(json.dumps({"error": "configuration differs from the classic kernel", "config": cfg}), flush=True)
                    sys.exit(2)
    del refs
    for rnd in range(a.rounds):
        for cfg in cfgs:
            apply(cfg)
            for lay in lays:
                for dist in dists:
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
                    print(json.dumps({"round": rnd, "config": cfg, "dtype": a.dtype, "tables": T, "unroll": a.unroll, "layout": lay, "indices": dist,
                                      "avg_launch_us": round(s * 1e6, 2), "G_lookups_per_s": round(T * B * L / s / 1e9, 3),
                                      "alg_frac_of_8TBps": round(alg / s / 8e12, 4)}), flush=True)
    param_amd.set_forward_persist()


if __name__ == "__main__":
    main()
