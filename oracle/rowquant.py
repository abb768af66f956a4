"""oracle/rowquant.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the row-wise quantisation the quantised all-to-all applies to pooled fp32 embeddings
(``param_amd/csrc/rowquant.hip``; reference flags ``--bitwidth`` / ``--quant-a2a-embedding-dim``,
train/comms/pt/comms_utils.py:1788-1806; downcast / restore around the collective,
train/comms/pt/pytorch_dist_backend.py:48-76).

The reference's own all-to-all quantiser is not published (``all_to_allv_internal``, pytorch_dist_backend.py:29-31,273
imports it from ``fb.internals``).  What IS published is the row format those flags describe -- fbgemm's fused row-wise
quantisation, which torch ships on CPU as ``quantized::embedding_bag_{byte,4bit,2bit}_prepack`` / ``_unpack`` -- and the
16-bit case, which the reference states itself (``_downcast``: ``input.to(torch.float16)``, :48-54).  This module is
PINNED to those torch operators: ``tests/golden/gen_rowquant.py`` stores their outputs for seeded inputs in
``tests/golden/rowquant.npz`` and ``tests/test_rowquant.py`` checks the restatement against the fixture and, live,
against the operators of the installed torch (bit for bit).

Only tests/ and __graft_entry__.smoke() may import this module.
"""
from __future__ import annotations

import numpy as np


def row_bytes(dim: int, bits: int) -> int:
    return 2 * dim if bits == 16 else dim + 8 if bits == 8 else dim * bits // 8 + 4


def quantize_rows(x: np.ndarray, bits: int) -> np.ndarray:
    """fp32 ``[n, dim]`` -> uint8 ``[n, row_bytes(dim, bits)]``"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, dim = x.shape
    if bits == 16:
        with np.errstate(over="ignore"):                                   # beyond fp16 range: inf, as torch's cast
            return x.astype(np.float16).view(np.uint8).reshape(n, 2 * dim)
    mn, mx = x.min(axis=1), x.max(axis=1)
    if bits == 8:
        rng = mx - mn
        scale = rng / np.float32(255.0)
        inv = np.float32(255.0) / (rng + np.float32(1e-8))
        codes = np.rint((x - mn[:, None]) * inv[:, None]).astype(np.uint8)
        tail = np.stack([scale, mn], axis=1).astype(np.float32).view(np.uint8).reshape(n, 8)
        return np.concatenate([codes, tail], axis=1)
    levels = np.float32((1 << bits) - 1)
    bias_h = mn.astype(np.float16)
    bias = bias_h.astype(np.float32)
    rng = mx - bias
    with np.errstate(divide="ignore", over="ignore"):
        scale_h = np.where(rng == 0, np.float32(1), rng / levels).astype(np.float16)
        scale_h = np.where(scale_h == 0, np.float16(1), scale_h)
        inv = np.float32(1) / scale_h.astype(np.float32)
    bad = np.isinf(inv)
    scale_h = np.where(bad, np.float16(1), scale_h).astype(np.float16)
    inv = np.where(bad, np.float32(1), inv).astype(np.float32)
    q = np.clip(np.rint((x - bias[:, None]) * inv[:, None]), 0, levels).astype(np.uint8)
    per = 8 // bits
    packed = np.zeros((n, dim // per), np.uint8)
    for k in range(per):
        packed |= q[:, k::per] << np.uint8(k * bits)
    return np.concatenate([packed, scale_h.view(np.uint8).reshape(n, 2), bias_h.view(np.uint8).reshape(n, 2)], axis=1)


def dequantize_rows(q: np.ndarray, dim: int, bits: int) -> np.ndarray:
    """uint8 ``[n, row_bytes]`` -> fp32 ``[n, dim]``: ``fma(code, scale, bias)``"""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    n = q.shape[0]
    if bits == 16:
        return q.reshape(n, 2 * dim).view(np.float16).astype(np.float32)
    if bits == 8:
        codes = q[:, :dim].astype(np.float64)
        sb = q[:, dim:dim + 8].copy().view(np.float32).astype(np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            return (codes * sb[:, :1] + sb[:, 1:]).astype(np.float32)   # one rounding, as fmaf
    per = 8 // bits
    nb = dim // per
    codes = np.zeros((n, dim), np.float64)
    for k in range(per):
        codes[:, k::per] = (q[:, :nb] >> np.uint8(k * bits)) & np.uint8((1 << bits) - 1)
    scale = q[:, nb:nb + 2].copy().view(np.float16).astype(np.float64)
    bias = q[:, nb + 2:nb + 4].copy().view(np.float16).astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        return (codes * scale + bias).astype(np.float32)
