"""JSON-config operator microbenchmark -- the slice of reference ``train/compute/python/pytorch/run_benchmark.py``
(``:24-365``) + ``lib/pytorch/build_executor.py`` / ``op_executor.py`` needed to run the batched EmbeddingBag
operator from the reference's config files (schema of ``examples/pytorch/configs/
split_table_batched_embeddings_ops.json``: op name -> ``config: [{build: [{args, kwargs}], input: [{args}]}]``).

    python -m param_amd.compute.python.run_benchmark -c my_config.json -d cuda -b --warmup 5 --iteration 20

Per (op, build, input) combination one JSON line ``{"op_name", "id", "metric": {"forward": {"gpu.time": [...ms]},
"backward": {...}}, "config"}`` (reference ``output_stats``, ``build_executor.py:511-541``).  Timing = the
reference's ``Timer``: host clock around the call closed by a device synchronize (``lib/pytorch/timer.py:19-27``).
Stated limits: plain ``value`` entries (no ``__range__`` / ``__list__`` macros), no L2-flush option (the
reference's flush table has no gfx950 entry and raises KeyError there), operators registered in this package only.
"""
from __future__ import annotations

import argparse
import json
import sys
import time

import torch

from . import op_map
from .split_table_batched_embeddings_ops import generate_batched_request


def _values(arg_list):
    return [a["value"] for a in arg_list]


def run_op(name: str, op_cfg: dict, device: str, warmup: int, iters: int, backward: bool, out_stream=None, alpha=1.0):
    if name not in op_map:
        raise KeyError(f"operator {name!r} is not registered (registered: {sorted(op_map)})")
    op = op_map[name]
    op.device = device
    out_stream = sys.stdout if out_stream is None else out_stream
    results = []
    for ci, cfg in enumerate(op_cfg["config"]):
        for bi, build in enumerate(cfg["build"]):
            bargs = _values(build["args"])
            bkw = {k: v["value"] for k, v in build.get("kwargs", {}).items()}
            op.cleanup()
            kw = dict(bkw)
            if len(bargs) < 7:
                kw.setdefault("optimizer", "exact_row_wise_adagrad")  # the reference's choice (comms_utils.py:2014)
            op.build(*bargs, **kw)
            num_tables, rows, _dim, _pool, weighted = bargs[0], bargs[1], bargs[2], bargs[3], bargs[4]
            for ii, inp in enumerate(cfg["input"]):
                batch_size, pooling_factor = _values(inp["args"])[:2]
                data = generate_batched_request(num_tables, rows, batch_size, pooling_factor, alpha=alpha,
                                                weighted=weighted, device="cuda" if device.startswith(("cuda", "rocm")) else device)
                metrics = {"forward": {"gpu.time": []}}
                if backward:
                    metrics["backward"] = {"gpu.time": []}
                for it in range(warmup + iters):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    op.forward(*data)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    if it >= warmup:
                        metrics["forward"]["gpu.time"].append((t1 - t0) * 1e3)
                    if backward:
                        op.create_grad()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        op.backward()
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        if it >= warmup:
                            metrics["backward"]["gpu.time"].append((t1 - t0) * 1e3)
                stats = {"op_name": name, "id": f"{ci}:{bi}:{ii}", "metric": metrics,
                         "config": {"build": {"args": bargs, "kwargs": bkw}, "input": {"args": [batch_size, pooling_factor]}}}
                out_stream.write(json.dumps(stats) + "\n")
                out_stream.flush()
                results.append(stats)
    op.cleanup()
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="operator microbenchmark from a JSON config (MI355X build)")
    ap.add_argument("-c", "--config", type=str, required=True)
    ap.add_argument("-d", "--device", type=str, default="cuda")
    ap.add_argument("-w", "--warmup", type=int, default=1)
    ap.add_argument("-i", "--iteration", type=int, default=1)
    ap.add_argument("-b", "--backward", action="store_true")
    ap.add_argument("--alpha", type=float, default=1.0, help="generate_requests distribution switch (reference :93-135)")
    a = ap.parse_args(argv)
    cfg = json.load(open(a.config))
    out = []
    for name, op_cfg in cfg.items():
        out += run_op(name, op_cfg, a.device, a.warmup, a.iteration, a.backward, alpha=a.alpha)
    return out


if __name__ == "__main__":
    main()  # pragma: no cover
