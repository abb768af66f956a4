"""oracle/embbag_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two CPU restatements of the reference's EmbeddingBag arithmetic:

* ``*_np``   -- numpy, a python loop over lookups (small cases only);
* ``COracle`` -- ctypes binding of ``embbag_oracle.c`` (``liboracle.so``), used
  for larger cases and as bench.py's ``cpu_baseline`` ("port", 1 core).

Reference call sites restated (paths relative to the reference root):
  forward   train/compute/pt/pytorch_emb.py:179,40,61 ; train/comms/pt/dlrm.py:380
  batched   train/compute/python/workloads/pytorch/split_table_batched_embeddings_ops.py:312,
            train/comms/pt/pytorch_dist_backend.py:221,845   (output [B, sum D])
  backward  split_table_batched_embeddings_ops.py:318-324, pytorch_dist_backend.py:854-857
The arithmetic is torch's (aten::_embedding_bag / _embedding_bag_dense_backward);
the oracle is pinned against torch outputs committed under tests/golden/
(generator: tests/golden/gen_golden.py), because the reference's own tests pin
nothing at this boundary.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

F32, BF16, F16 = 0, 1, 2


# --------------------------------------------------------------------------- #
# numpy restatement
# --------------------------------------------------------------------------- #
def bag_bounds(off: np.ndarray, n_bags: int, n_idx: int):
    """(start, end) per bag: last bag ends at len(indices); a trailing
    offsets[B]==N entry (TBE layout) is tolerated and never read."""
    off = np.asarray(off, dtype=np.int64)
    start = off[:n_bags]
    end = np.empty(n_bags, dtype=np.int64)
    end[:-1] = off[1:n_bags]
    end[-1] = n_idx
    return start, end


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    lsb = (u >> 16) & 1
    return ((u + 0x7FFF + lsb) >> 16).astype(np.uint16)


def embbag_fwd_np(W: np.ndarray, idx, off, psw=None) -> np.ndarray:
    """out[b,:] = sum_j W[idx[j],:], sequential fp32, W already fp32 [R,D]."""
    W = np.asarray(W, dtype=np.float32)
    idx = np.asarray(idx, dtype=np.int64)
    B = len(off)
    start, end = bag_bounds(off, B, len(idx))
    out = np.zeros((B, W.shape[1]), dtype=np.float32)
    for b in range(B):
        acc = np.zeros(W.shape[1], dtype=np.float32)
        for j in range(start[b], end[b]):
            r = idx[j]
            if r < 0 or r >= W.shape[0]:
                raise IndexError(f"index {r} out of range [0,{W.shape[0]})")
            row = W[r]
            if psw is not None:
                # fused multiply-add (one rounding): the f32*f32 product is exact in
                # f64, so add in f64 and round once
                acc = (acc.astype(np.float64) + np.float64(psw[j]) * row.astype(np.float64)).astype(np.float32)
            else:
                acc = (acc + row).astype(np.float32)
        out[b] = acc
    return out


def embbag_fwd_batched_np(tables, idx, off, B, psw=None) -> np.ndarray:
    """TBE layout: idx concatenated table-major, off [T*B(+1)], out [B, sum D]."""
    idx = np.asarray(idx, dtype=np.int64)
    T = len(tables)
    start, end = bag_bounds(off, T * B, len(idx))
    cols = [t.shape[1] for t in tables]
    out = np.zeros((B, sum(cols)), dtype=np.float32)
    c0 = 0
    for t, W in enumerate(tables):
        s, e = start[t * B], end[t * B + B - 1]
        local_off = start[t * B:(t + 1) * B] - s
        out[:, c0:c0 + cols[t]] = embbag_fwd_np(
            W, idx[s:e], local_off, None if psw is None else psw[s:e])
        c0 += cols[t]
    return out


def embbag_bwd_np(rows: int, idx, off, grad: np.ndarray, psw=None, alpha=1.0,
                  dst: np.ndarray | None = None) -> np.ndarray:
    """dst[idx[j],:] += alpha*psw[j]*grad[bag(j),:]  sequential fp32."""
    idx = np.asarray(idx, dtype=np.int64)
    grad = np.asarray(grad, dtype=np.float32)
    B, D = grad.shape
    start, end = bag_bounds(off, B, len(idx))
    if dst is None:
        dst = np.zeros((rows, D), dtype=np.float32)
    a = np.float32(alpha)
    for b in range(B):
        for j in range(start[b], end[b]):
            scale = a if psw is None else np.float32(a * np.float32(psw[j]))
            dst[idx[j]] = (dst[idx[j]] + (scale * grad[b]).astype(np.float32)).astype(np.float32)
    return dst


# --------------------------------------------------------------------------- #
# C restatement
# --------------------------------------------------------------------------- #
def build(force: bool = False) -> str:
    """Compile embbag_oracle.c -> liboracle.so (gcc). Building the checker is
    not using it; __graft_entry__.build() calls this."""
    src = os.path.join(_HERE, "embbag_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


def _ptr(a: np.ndarray | None, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctype))


class COracle:
    """ctypes binding of liboracle.so; numpy arrays in, numpy arrays out."""

    def __init__(self):
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, f32p = ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_embbag_fwd.restype = ctypes.c_int
        L.oracle_embbag_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int, i64, i32, i64p, i64, i64p, i64,
                                        f32p, f32p, i64]
        L.oracle_embbag_fwd_batched.restype = ctypes.c_int
        L.oracle_embbag_fwd_batched.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, i64p,
                                                ctypes.POINTER(i32), i32, i64p, i64, i64p, i64, f32p,
                                                f32p, i64p, i64]
        L.oracle_embbag_bwd_f32.restype = ctypes.c_int
        L.oracle_embbag_bwd_f32.argtypes = [f32p, i64, i32, i64p, i64, i64p, i64, f32p, f32p, i64,
                                            ctypes.c_float]
        L.oracle_embbag_bwd_bf16.restype = ctypes.c_int
        L.oracle_embbag_bwd_bf16.argtypes = [ctypes.POINTER(ctypes.c_uint16), f32p, i64, i32, i64p, i64,
                                             i64p, i64, f32p, f32p, i64, ctypes.c_float]
        L.oracle_embbag_bwd_rowwise_adagrad_wd_f32.restype = ctypes.c_int
        L.oracle_embbag_bwd_rowwise_adagrad_wd_f32.argtypes = [f32p, f32p, f32p, ctypes.POINTER(ctypes.c_uint8), i64, i32,
                                                               i64p, i64, i64p, i64, f32p, f32p, i64, ctypes.c_float,
                                                               ctypes.c_float, ctypes.c_float, i32]
        L.oracle_embbag_bwd_rowwise_adagrad_f32.restype = ctypes.c_int
        L.oracle_embbag_bwd_rowwise_adagrad_f32.argtypes = [f32p, f32p, f32p, ctypes.POINTER(ctypes.c_uint8), i64, i32,
                                                            i64p, i64, i64p, i64, f32p, f32p, i64, ctypes.c_float,
                                                            ctypes.c_float]
        self.L = L

    @staticmethod
    def _check(rc):
        if rc == -1:
            raise IndexError("oracle: index out of range")
        if rc == -2:
            raise ValueError("oracle: offsets not monotone / out of range")
        if rc != 0:
            raise RuntimeError(f"oracle: error {rc}")

    @staticmethod
    def _wdtype(W: np.ndarray, dtype):
        if dtype is not None:
            return dtype
        if W.dtype == np.float32:
            return F32
        if W.dtype == np.float16:
            return F16
        raise TypeError("pass dtype=BF16 with a uint16 array for bf16 tables")

    def fwd(self, W: np.ndarray, idx, off, psw=None, dtype=None) -> np.ndarray:
        W = np.ascontiguousarray(W)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        off = np.ascontiguousarray(off, dtype=np.int64)
        psw = None if psw is None else np.ascontiguousarray(psw, dtype=np.float32)
        B, D = len(off), W.shape[1]
        out = np.empty((B, D), dtype=np.float32)
        rc = self.L.oracle_embbag_fwd(W.ctypes.data, self._wdtype(W, dtype), W.shape[0], D,
                                      _ptr(idx, ctypes.c_int64), len(idx), _ptr(off, ctypes.c_int64), B,
                                      _ptr(psw, ctypes.c_float), _ptr(out, ctypes.c_float), D)
        self._check(rc)
        return out

    def fwd_batched(self, tables, idx, off, B, psw=None, dtype=None, layout="bd") -> np.ndarray:
        """layout "bd": out [B, sum D] (TBE); "tbd": out [T, B, D] (dlrm.py stack)."""
        tables = [np.ascontiguousarray(t) for t in tables]
        T = len(tables)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        off = np.ascontiguousarray(off, dtype=np.int64)
        psw = None if psw is None else np.ascontiguousarray(psw, dtype=np.float32)
        rows = np.array([t.shape[0] for t in tables], dtype=np.int64)
        dims = np.array([t.shape[1] for t in tables], dtype=np.int32)
        if layout == "bd":
            out_off = np.concatenate([[0], np.cumsum(dims[:-1])]).astype(np.int64)
            stride = int(dims.sum())
            out = np.empty((B, stride), dtype=np.float32)
        else:
            assert len(set(dims.tolist())) == 1
            D = int(dims[0])
            out_off = (np.arange(T, dtype=np.int64) * B * D)
            stride = D
            out = np.empty((T, B, D), dtype=np.float32)
        ptrs = (ctypes.c_void_p * T)(*[t.ctypes.data for t in tables])
        rc = self.L.oracle_embbag_fwd_batched(ptrs, self._wdtype(tables[0], dtype),
                                              _ptr(rows, ctypes.c_int64), _ptr(dims, ctypes.c_int32), T,
                                              _ptr(idx, ctypes.c_int64), len(idx),
                                              _ptr(off, ctypes.c_int64), B, _ptr(psw, ctypes.c_float),
                                              _ptr(out, ctypes.c_float), _ptr(out_off, ctypes.c_int64),
                                              stride)
        self._check(rc)
        return out

    def bwd_f32(self, dst: np.ndarray, idx, off, grad: np.ndarray, psw=None, alpha=1.0) -> np.ndarray:
        """in-place on dst (fp32 [R,D]); returns dst."""
        assert dst.dtype == np.float32 and dst.flags.c_contiguous
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        off = np.ascontiguousarray(off, dtype=np.int64)
        grad = np.ascontiguousarray(grad, dtype=np.float32)
        psw = None if psw is None else np.ascontiguousarray(psw, dtype=np.float32)
        rc = self.L.oracle_embbag_bwd_f32(_ptr(dst, ctypes.c_float), dst.shape[0], dst.shape[1],
                                          _ptr(idx, ctypes.c_int64), len(idx), _ptr(off, ctypes.c_int64),
                                          len(off), _ptr(psw, ctypes.c_float), _ptr(grad, ctypes.c_float),
                                          grad.shape[1], float(alpha))
        self._check(rc)
        return dst

    def bwd_rowwise_adagrad(self, W: np.ndarray, mom: np.ndarray, idx, off, grad: np.ndarray, psw=None, lr=0.01,
                            eps=1e-8, weight_decay=0.0, weight_decay_mode=0):
        """in place on W (fp32 [R,D]) and mom (fp32 [R]); PARITY UNPINNED (fbgemm absent), see the C header."""
        assert W.dtype == np.float32 and W.flags.c_contiguous and mom.dtype == np.float32 and mom.shape == (W.shape[0],)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        off = np.ascontiguousarray(off, dtype=np.int64)
        grad = np.ascontiguousarray(grad, dtype=np.float32)
        psw = None if psw is None else np.ascontiguousarray(psw, dtype=np.float32)
        scratch = np.empty_like(W)
        touched = np.empty(W.shape[0], dtype=np.uint8)
        rc = self.L.oracle_embbag_bwd_rowwise_adagrad_wd_f32(
            _ptr(W, ctypes.c_float), _ptr(mom, ctypes.c_float), _ptr(scratch, ctypes.c_float),
            _ptr(touched, ctypes.c_uint8), W.shape[0], W.shape[1], _ptr(idx, ctypes.c_int64), len(idx),
            _ptr(off, ctypes.c_int64), len(off), _ptr(psw, ctypes.c_float), _ptr(grad, ctypes.c_float), grad.shape[1],
            float(lr), float(eps), float(weight_decay), int(weight_decay_mode))
        self._check(rc)
        return W, mom

    def bwd_bf16(self, dst_bits: np.ndarray, idx, off, grad: np.ndarray, psw=None, alpha=1.0):
        """in-place on a uint16 (bf16 bit pattern) table."""
        assert dst_bits.dtype == np.uint16 and dst_bits.flags.c_contiguous
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        off = np.ascontiguousarray(off, dtype=np.int64)
        grad = np.ascontiguousarray(grad, dtype=np.float32)
        psw = None if psw is None else np.ascontiguousarray(psw, dtype=np.float32)
        scratch = np.empty(dst_bits.shape, dtype=np.float32)
        rc = self.L.oracle_embbag_bwd_bf16(_ptr(dst_bits, ctypes.c_uint16), _ptr(scratch, ctypes.c_float),
                                           dst_bits.shape[0], dst_bits.shape[1],
                                           _ptr(idx, ctypes.c_int64), len(idx),
                                           _ptr(off, ctypes.c_int64), len(off),
                                           _ptr(psw, ctypes.c_float), _ptr(grad, ctypes.c_float),
                                           grad.shape[1], float(alpha))
        self._check(rc)
        return dst_bits
