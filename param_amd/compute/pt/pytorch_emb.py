"""EmbeddingBag microbenchmark driver -- same CLI, timing protocol and stdout table as the
reference ``train/compute/pt/pytorch_emb.py`` (run: ``:208-234``, run_single: ``:163-205``,
measure_cpu / measure_gpu: ``:37-69``, CLI: ``:237-270``), with the ``gpu`` device served by
the MI355X HIP kernels (:class:`param_amd.EmbeddingBagMI355`).

Kept from the reference: flag names and defaults, ``torch.manual_seed(randomseed)`` before
index generation, fixed-L offsets, the bytes metric ``batch*nnz*embdim*elem_size``
(``:180``), clock restart after the last warm-up, one ``synchronize`` at the end of the GPU
loop (throughput, not latency), and the row format.

Differences (stated, deliberate):
  * ``--device gpu`` runs the hand-written HIP forward, never torch's kernel, and exits with
    an error if libparam_amd.so is missing (no fallback);
  * ``--device cpu`` is the reference's own CPU path (torch.nn.EmbeddingBag on host cores):
    the baseline the GPU number is printed next to;
  * ``--dtype`` is honoured for the table (float32 / bfloat16 / float16; reference bug R6
    ignores it), ``--alpha`` may be a string (R2), numpy is seeded too (R3);
  * ``--tables T`` (extension) runs the batched multi-table kernel; ``--json`` appends one
    machine-readable line per row with lookups/s, PARAM GB/s and algorithmic GB/s.
"""
from __future__ import annotations

import json
import sys
import time

import numpy as np
import torch
import torch.nn as nn

from ...indices import fixed_offsets, init_indices, zipf_indices

_DTYPES = {"float32": torch.float32, "float": torch.float32, "bfloat16": torch.bfloat16,
           "float16": torch.float16}

HEADER_RULE = "-" * 81
HEADER = "    Features    embdim    nnz     batch      Time(s)/step   Data(MB)   BW(GB/s)"


def measure_cpu(warmups, steps, h_emb, h_indices, h_offsets):
    """warmups+steps calls; the clock restarts after the last warm-up (reference :37-45)."""
    start = time.perf_counter()
    results = None
    for i in range(warmups + steps):
        results = h_emb(h_indices, h_offsets)
        if i < warmups:
            start = time.perf_counter()
    return time.perf_counter() - start, results


def measure_gpu(warmups, steps, g_emb, g_indices, g_offsets):
    """Async launch loop closed by one device synchronize (reference :48-69)."""
    dev = torch.device("cuda:0")
    with torch.cuda.device(dev):
        g_emb = g_emb.to(dev)
        g_indices = g_indices.to(dev)
        g_offsets = g_offsets.to(dev)
        torch.cuda.synchronize()
        start = time.perf_counter()
        results = None
        for i in range(warmups + steps):
            results = g_emb(g_indices, g_offsets)
            if i < warmups:
                torch.cuda.synchronize()
                start = time.perf_counter()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
    return elapsed, results


def algorithmic_bytes(tables: int, batch: int, nnz: int, embdim: int, elem_size: int, index_size: int = 8) -> int:
    """per lookup D*e + idx bytes read; per bag 8 B offset read + D*4 B written."""
    bags = tables * batch
    return bags * nnz * (embdim * elem_size + index_size) + bags * (embdim * 4 + index_size)


def run_single(args, features, embdim, nnz, batch):
    """One row: returns (elapsed seconds for args.steps steps, PARAM bytes per step)."""
    device = args.device
    torch.manual_seed(args.randomseed)
    np.random.seed(args.randomseed)
    dtype = _DTYPES[getattr(args, "dtype", "float32")]
    tables = int(getattr(args, "tables", 1))
    alpha = float(args.alpha)
    elem = torch.empty(0, dtype=dtype).element_size()
    total_bytes = tables * batch * nnz * embdim * elem

    if device == "cpu":
        # the reference's CPU path, unchanged in substance: torch.nn.EmbeddingBag on host cores
        h_indices = init_indices(alpha, features, batch, nnz)
        h_offsets = fixed_offsets(batch, nnz)
        h_emb = nn.EmbeddingBag(features, embdim, mode="sum")
        if dtype != torch.float32:
            h_emb = h_emb.to(dtype)
        if getattr(args, "no_grad", False):
            with torch.no_grad():
                elapsed, _ = measure_cpu(args.warmups, args.steps, h_emb, h_indices, h_offsets)
        else:
            elapsed, _ = measure_cpu(args.warmups, args.steps, h_emb, h_indices, h_offsets)
        return elapsed, batch * nnz * embdim * elem

    if device != "gpu":
        print(f"device '{device}' is not served by the MI355X build (cpu | gpu)")
        sys.exit(1)
    if not torch.cuda.is_available():
        print("ROCm device is not available, could not run on GPU")
        sys.exit(1)

    from ... import BatchedEmbeddingBagMI355, EmbeddingBagMI355  # raises if the .so is missing

    dev = torch.device("cuda:0")
    if tables == 1:
        if alpha == 0.0 or batch * nnz <= 1 << 20:
            g_indices = init_indices(alpha, features, batch, nnz).to(dev)
        else:
            g_indices = zipf_indices(alpha, features, batch, nnz, device=dev)
        g_offsets = fixed_offsets(batch, nnz, device=dev)
        g_emb = EmbeddingBagMI355(features, embdim, mode="sum", dtype=dtype, device=dev)
        g_emb.weight.requires_grad_(False)
        elapsed, _ = measure_gpu(args.warmups, args.steps, g_emb, g_indices, g_offsets)
    else:
        from ...indices import tbe_request
        g_emb = BatchedEmbeddingBagMI355([features] * tables, embdim, dtype=dtype, device=dev,
                                         init="normal", seed=args.randomseed, fused_update=False)
        g_indices, g_offsets = tbe_request([features] * tables, batch, nnz, alpha, device=dev,
                                           seed=args.randomseed)
        out = torch.empty((batch, tables * embdim), dtype=torch.float32, device=dev)
        fn = lambda i, o: g_emb.lookup(i, o, out=out)  # noqa: E731
        torch.cuda.synchronize()
        start = time.perf_counter()
        for i in range(args.warmups + args.steps):
            fn(g_indices, g_offsets)
            if i < args.warmups:
                torch.cuda.synchronize()
                start = time.perf_counter()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
    return elapsed, total_bytes


def format_row(features, embdim, nnz, batch, elap, total_mb):
    """Row format of the reference (:225-233)."""
    return "{:10},  {:6},  {:6},  {:8},    {:10.6f}, {:10.1f},  {:8.3f}".format(
        features, embdim, nnz, batch, elap, total_mb, total_mb / elap / 1.0e3)


def run(args, dataset):
    print(HEADER_RULE)
    print(HEADER)
    print(HEADER_RULE)
    tables = int(getattr(args, "tables", 1))
    dtype = _DTYPES[getattr(args, "dtype", "float32")]
    elem = torch.empty(0, dtype=dtype).element_size()
    for features, embdim, nnz, batch in dataset:
        elap, total_bytes = run_single(args, features, embdim, nnz, batch)
        elap /= args.steps
        print(format_row(features, embdim, nnz, batch, elap, total_bytes / 1.0e6))
        if getattr(args, "json", False):
            lookups = tables * batch * nnz
            alg = algorithmic_bytes(tables, batch, nnz, embdim, elem)
            print(json.dumps({
                "features": features, "embdim": embdim, "nnz": nnz, "batch": batch, "tables": tables,
                "dtype": str(dtype).replace("torch.", ""), "device": args.device, "alpha": float(args.alpha),
                "s_per_step": elap, "lookups_per_s": lookups / elap,
                "param_GBps": total_bytes / elap / 1e9, "algorithmic_GBps": alg / elap / 1e9,
                "hbm_roofline_frac": (alg / elap / 1e9) / 8000.0 if args.device == "gpu" else None,
            }))


def build_parser():
    import argparse

    parser = argparse.ArgumentParser(description="Measure the performance of EmbeddingBag (MI355X build)")
    parser.add_argument("--features", type=int, default=1024)
    parser.add_argument("--embdim", type=int, default=64)
    parser.add_argument("--nnz", type=int, default=10)
    parser.add_argument("--batch", type=int, default=1000)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--warmups", type=int, default=1)
    parser.add_argument("--randomseed", type=int, default=0)
    parser.add_argument("-t", "--dtype", type=str, default="float32", choices=sorted(_DTYPES))
    parser.add_argument("-d", "--device", choices=["cpu", "gpu", "tpu"], type=str, default="cpu")
    parser.add_argument("--usexlabag", action="store_true", help="accepted for CLI compatibility; TPU-only in the reference")
    parser.add_argument("--alpha", type=float, default=0.0, help="Zipf param. Use uniform if == 0.0")
    # extensions
    parser.add_argument("--tables", type=int, default=1, help="number of tables looked up by one batched launch")
    parser.add_argument("--json", action="store_true", help="also print one JSON line per row")
    parser.add_argument("--no-grad", dest="no_grad", action="store_true", help="cpu: run under torch.no_grad()")
    return parser


def main() -> None:
    args = build_parser().parse_args()
    run(args, [(args.features, args.embdim, args.nnz, args.batch)])


if __name__ == "__main__":
    main()  # pragma: no cover
