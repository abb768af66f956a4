"""Row-wise quantisation of pooled fp32 embeddings on the GPU (``csrc/rowquant.hip`` through the C ABI).

What the reference's comms drivers switch on with ``--bitwidth {2,4,8,16}`` / ``--quant-a2a-embedding-dim`` /
``--quant-threshold`` (reference train/comms/pt/comms_utils.py:1788-1806): a float32 payload is downcast before the
collective and restored after it (pytorch_dist_backend.py:48-76; the all-to-all variant, ``all_to_allv_internal`` at :273,
is not published).  Row formats: fp16 for 16 bits, fbgemm's fused row-wise formats (scale and bias stored behind each
row's codes) for 8 / 4 / 2 bits -- the same bytes as torch's ``quantized::embedding_bag_*_prepack`` operators.

There is no CPU path in this module: tensors must live on the GPU, and a GPU tensor is only ever quantised by the HIP
kernels.  (The comms plug-in refuses HOST tensors too unless a test injects a codec: ``MI355XBackend.host_row_codec``, set by the gloo
workers of ``tests/dist_workers.py`` to torch's own ``quantized::embedding_bag_*_prepack`` operators -- the operators
``oracle/rowquant.py`` is pinned to.  Round 6 moved them out of the product backend.)
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

BITWIDTHS = (16, 8, 4, 2)


def host_row_bytes(dim: int, bitwidth: int) -> int:
    """bytes of one quantised row, computed on the host (same table as ``pm_rows_quantized_bytes``)"""
    if bitwidth not in BITWIDTHS:
        raise ValueError(f"bitwidth must be one of {BITWIDTHS}, got {bitwidth}")
    return 2 * dim if bitwidth == 16 else dim + 8 if bitwidth == 8 else dim * bitwidth // 8 + 4


def row_bytes(dim: int, bitwidth: int) -> int:
    """bytes of one quantised row of ``dim`` fp32 values"""
    n = _lib.load().pm_rows_quantized_bytes(1, int(dim), int(bitwidth))
    if n < 0:
        _lib.check(int(n))
    return int(n)


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check_src(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: tensor must be on the GPU (there is no CPU path)")
    if not x.is_contiguous():
        raise ValueError(f"{what}: tensor must be contiguous")


def quantize_rows(x: torch.Tensor, dim: int, bitwidth: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """``x``: float32, ``numel % dim == 0``, read as rows of ``dim`` values.  Returns a uint8 tensor ``[n_rows, row_bytes]``."""
    _check_src(x, "quantize_rows")
    if x.dtype != torch.float32:
        raise TypeError(f"quantize_rows: float32 expected, got {x.dtype}")      # as the reference: quantisation is fp32-only
    if dim <= 0 or x.numel() % dim != 0:
        raise ValueError(f"quantize_rows: {x.numel()} elements are not a whole number of rows of {dim}")
    n = x.numel() // dim
    rb = row_bytes(dim, bitwidth)
    if out is None:
        out = torch.empty((n, rb), dtype=torch.uint8, device=x.device)
    elif out.dtype != torch.uint8 or out.numel() != n * rb or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"quantize_rows: out must be a contiguous uint8 tensor of {n * rb} bytes on {x.device}")
    with torch.cuda.device(x.device):
        rc = _lib.load().pm_rows_quantize(ctypes.c_void_p(x.data_ptr()), n, int(dim), int(bitwidth),
                                          ctypes.c_void_p(out.data_ptr()), _stream(x))
    _lib.check(rc)
    return out


def dequantize_rows(q: torch.Tensor, dim: int, bitwidth: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """inverse of :func:`quantize_rows`: uint8 rows -> float32 ``[n_rows, dim]`` (``fma(code, scale, bias)``)"""
    _check_src(q, "dequantize_rows")
    if q.dtype != torch.uint8:
        raise TypeError(f"dequantize_rows: uint8 expected, got {q.dtype}")
    rb = row_bytes(dim, bitwidth)
    if q.numel() % rb != 0:
        raise ValueError(f"dequantize_rows: {q.numel()} bytes are not a whole number of {rb}-byte rows")
    n = q.numel() // rb
    if out is None:
        out = torch.empty((n, dim), dtype=torch.float32, device=q.device)
    elif out.dtype != torch.float32 or out.numel() != n * dim or not out.is_contiguous() or out.device != q.device:
        raise ValueError(f"dequantize_rows: out must be a contiguous float32 tensor of {n * dim} elements on {q.device}")
    with torch.cuda.device(q.device):
        rc = _lib.load().pm_rows_dequantize(ctypes.c_void_p(q.data_ptr()), n, int(dim), int(bitwidth),
                                            ctypes.c_void_p(out.data_ptr()), _stream(q))
    _lib.check(rc)
    return out
