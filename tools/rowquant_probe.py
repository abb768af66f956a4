"""Throughput of the row quantisers (csrc/rowquant.hip) against the HBM roofline: one rank's forward payload of the 8-GPU
DLRM exchange (8 x 8192 bags x 4 tables x 128 fp32 = 134 MB) and a 1 GiB payload.  Algorithmic bytes = fp32 rows + quantised
rows (each read or written once).  Prints one JSON line per (payload, bitwidth, direction)."""
import json
import sys

import torch

sys.path.insert(0, ".")
from param_amd import quant  # noqa: E402


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dim = 128
    for n in (8 * 8192 * 4, (1 << 30) // (dim * 4)):
        x = torch.randn(n, dim, device="cuda")
        for bits in (16, 8, 4, 2, 4, 8, 16):
            q = quant.quantize_rows(x, dim, bits)
            d = torch.empty_like(x)
            nbytes = x.numel() * 4 + q.numel()
            tq = timed(lambda: quant.quantize_rows(x, dim, bits, out=q))
            td = timed(lambda: quant.dequantize_rows(q, dim, bits, out=d))
            for name, t in (("quantize", tq), ("dequantize", td)):
                print(json.dumps({"rows": n, "dim": dim, "bitwidth": bits, "kernel": name, "us": round(t * 1e6, 1),
                                  "alg_GBps": round(nbytes / t / 1e9, 1), "frac_of_8TBps": round(nbytes / t / 8e12, 3)}))
        # yardsticks on the same buffers: torch's cast, a device copy, a fill
        d = torch.empty_like(x)
        h = torch.empty(n, dim, dtype=torch.float16, device="cuda")
        for name, fn, nbytes in (("torch copy_ f32->f16", lambda: h.copy_(x), x.numel() * 6),
                                 ("torch copy_ f16->f32", lambda: d.copy_(h), x.numel() * 6),
                                 ("torch copy_ f32->f32", lambda: d.copy_(x), x.numel() * 8),
                                 ("torch fill_ f32", lambda: d.fill_(1.0), x.numel() * 4)):
            t = timed(fn)
            print(json.dumps({"rows": n, "dim": dim, "kernel": name, "us": round(t * 1e6, 1),
                              "alg_GBps": round(nbytes / t / 1e9, 1), "frac_of_8TBps": round(nbytes / t / 8e12, 3)}))


if __name__ == "__main__":
    main()
