"""Unified CLI ``driver.py --device ... emb -d {A,B}`` -- the reference's
``train/compute/pt/driver.py:12-87`` restricted to the kernel on the hot path (``emb``).

The ``gemm`` and ``linear`` sub-commands are accepted by the parser so existing command
lines fail with a clear message rather than an argparse error: they are dense-MFMA
benchmarks outside this build's scope.
"""
from __future__ import annotations

import argparse
import sys

from . import dataset
from . import pytorch_emb as kemb


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Measuring the Compute Kernel Performance (MI355X build)")
    parser.add_argument("--warmups", type=int, default=10, help="warmup times")
    parser.add_argument("--steps", type=int, default=100, help="repeat times")
    parser.add_argument("--device", type=str, choices=["cpu", "gpu", "tpu"], required=True, help="valid devices")
    sub = parser.add_subparsers(title="kernels", dest="kernel")
    sub.required = True
    p_emb = sub.add_parser("emb", help="measure EmbeddingBag performance")
    p_emb.add_argument("-d", "--dataset", choices=["A", "B"], default="A")
    p_emb.add_argument("--randomseed", type=int, default=0)
    p_emb.add_argument("--usexlabag", action="store_true", help="accepted for compatibility (TPU only)")
    # type=float: the reference forgets it (bug R2) and crashes on any --alpha value
    p_emb.add_argument("--alpha", type=float, default=0.0, help="Zipf param. Use uniform if == 0.0")
    p_emb.add_argument("-t", "--dtype", type=str, default="float32")
    p_emb.add_argument("--tables", type=int, default=1)
    p_emb.add_argument("--json", action="store_true")
    for name in ("gemm", "linear"):
        sub.add_parser(name, help="not part of the MI355X embedding build")
    return parser


def main(argv=None) -> None:
    args, _ = build_parser().parse_known_args(argv)
    print("Measuring the performance of ", args.kernel, " on device = ", args.device)
    print("Steps = ", args.steps, " warmups = ", args.warmups)
    if args.kernel != "emb":
        print(f"kernel '{args.kernel}' is outside the MI355X embedding hot path; use the reference driver")
        sys.exit(2)
    print("with emb dataset ", args.dataset)
    kemb.run(args, dataset.emb_A if args.dataset == "A" else dataset.emb_B)


if __name__ == "__main__":
    main()  # pragma: no cover
