"""Generates tests/golden/rowquant.npz: outputs of torch's CPU operators for the fused row-wise quantised formats
(quantized::embedding_bag_{byte,4bit,2bit}_prepack / _unpack = fbgemm's published formats) and of the reference's own
16-bit downcast (``input.to(torch.float16)``, reference train/comms/pt/pytorch_dist_backend.py:48-54) on seeded inputs.
They pin oracle/rowquant.py.  Run in the build container:  python tests/golden/gen_rowquant.py
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rowquant.npz")


def inputs():
    g = torch.Generator().manual_seed(20260927)
    cases = {}
    for dim in (32, 64, 128, 256):                                  # the reference's --quant-a2a-embedding-dim choices
        x = torch.randn(40, dim, generator=g) * (torch.rand(40, 1, generator=g) * 10 + 0.01)
        x[0] = 1.5                                                  # constant rows: range 0
        x[1] = 0.0
        x[2] = -7.0
        x[3] = torch.arange(dim, dtype=torch.float32)
        x[4] = torch.randn(dim, generator=g) * 1e-6                 # fp16 scale underflows to 0 -> 1
        x[5] = torch.randn(dim, generator=g) * 1e4
        x[6] = torch.randn(dim, generator=g).abs() * 70000          # beyond fp16 range: inf after the 16-bit downcast
        cases[dim] = x
    cases[8] = torch.randn(5, 8, generator=g)                       # smallest row the kernels take
    cases[96] = torch.randn(7, 96, generator=g) * 3                 # lane group not full (12 of 16 lanes)
    return cases


def main():
    ops = {8: (torch.ops.quantized.embedding_bag_byte_prepack, torch.ops.quantized.embedding_bag_byte_unpack),
           4: (torch.ops.quantized.embedding_bag_4bit_prepack, torch.ops.quantized.embedding_bag_4bit_unpack),
           2: (torch.ops.quantized.embedding_bag_2bit_prepack, torch.ops.quantized.embedding_bag_2bit_unpack)}
    out = {}
    for dim, x in inputs().items():
        out[f"x_{dim}"] = x.numpy()
        h = x.to(torch.float16)
        out[f"q16_{dim}"] = h.numpy().view(np.uint8).reshape(x.shape[0], -1)
        out[f"d16_{dim}"] = h.to(torch.float32).numpy()
        for bits, (pack, unpack) in ops.items():
            q = pack(x)
            out[f"q{bits}_{dim}"] = q.numpy()
            out[f"d{bits}_{dim}"] = unpack(q).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; torch", torch.__version__)


if __name__ == "__main__":
    main()
