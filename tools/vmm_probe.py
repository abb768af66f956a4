#!/usr/bin/env python3
"""Experiment: does the VIRTUAL-address alignment of the table slab change the forward's bandwidth?
The forward on a > 40 GB random working set is translation-bound (profiles/r01_pmc_translation.md: UTCL2 94 % busy).
amdgpu can map physically contiguous VRAM blocks with translation fragments larger than 2 MiB when the virtual and
physical addresses share the alignment.  This probe allocates the same tables (a) through torch (hipMalloc) and
(b) through HIP's virtual-memory API (hipMemCreate + hipMemAddressReserve with a chosen alignment + hipMemMap) and
times the same forward request on both through the C ABI."""
import argparse, ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from param_amd import _lib
from param_amd.indices import tbe_request

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=16)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--align-gb", type=float, default=1.0)
ap.add_argument("--chunk-gb", type=float, default=0.0, help="physical handle size (0 = one handle for everything)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
L = _lib.load()
T, R, D, B, Lp = a.tables, a.rows, 128, 8192, 20
table_bytes = R * D * 4


class Loc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class AllocFlags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class Prop(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", Loc),
                ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", AllocFlags)]


class AccessDesc(ctypes.Structure):
    _fields_ = [("location", Loc), ("flags", ctypes.c_int)]


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: hipError {rc}")


def vmm_alloc(nbytes, align, chunk):
    prop = Prop(); prop.type = 1; prop.requestedHandleType = 0; prop.location = Loc(1, 0)
    gran = ctypes.c_size_t()
    chk(hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 1), "granularity")
    g = max(gran.value, 1 << 21)
    size = (nbytes + g - 1) // g * g
    ptr = ctypes.c_void_p()
    # the reserve call does not honour large alignments: over-reserve and align the mapping start by hand
    chk(hip.hipMemAddressReserve(ctypes.byref(ptr), ctypes.c_size_t(size + align), ctypes.c_size_t(0), None, ctypes.c_ulonglong(0)), "reserve")
    ptr = ctypes.c_void_p((ptr.value + align - 1) // align * align)
    chunk = size if chunk <= 0 else (int(chunk) + g - 1) // g * g
    off = 0
    while off < size:
        n = min(chunk, size - off)
        h = ctypes.c_void_p()
        chk(hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(n), ctypes.byref(prop), ctypes.c_ulonglong(0)), "create")
        chk(hip.hipMemMap(ctypes.c_void_p(ptr.value + off), ctypes.c_size_t(n), ctypes.c_size_t(0), h, ctypes.c_ulonglong(0)), "map")
        off += n
    desc = AccessDesc(Loc(1, 0), 3)
    chk(hip.hipMemSetAccess(ptr, ctypes.c_size_t(size), ctypes.byref(desc), ctypes.c_size_t(1)), "setaccess")
    return ptr.value, g, size


def run(base_ptr, tag, extra):
    ptrs = torch.tensor([base_ptr + t * table_bytes for t in range(T)], dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for t in range(T):
        _lib.check(L.pm_fill_random(base_ptr + t * table_bytes, R * D, _lib.PM_F32, 1, 0.0, 1.0, 1000 + t, stream))
    rows = torch.tensor([R] * T, dtype=torch.int64, device=dev)
    dims = torch.tensor([D] * T, dtype=torch.int32, device=dev)
    col0 = torch.arange(T, dtype=torch.int64, device=dev) * D
    out = torch.empty((B, T * D), device=dev)
    res = {}
    for alpha in (0.0, 1.05):
        idx, off = tbe_request([R] * T, B, Lp, alpha, device=dev, seed=3)
        op = _lib.pm_embbag_batch()
        op.num_tables, op.weight_dtype, op.index_dtype, op.max_dim = T, _lib.PM_F32, _lib.PM_I64, D
        op.batch, op.num_indices, op.bag_begin, op.bag_count = B, idx.numel(), 0, B
        op.tables, op.rows, op.dims, op.out_offsets = ptrs.data_ptr(), rows.data_ptr(), dims.data_ptr(), col0.data_ptr()
        op.out_stride, op.indices, op.offsets, op.per_sample_weights = T * D, idx.data_ptr(), off.data_ptr(), None
        fn = lambda: _lib.check(L.pm_embbag_fwd(ctypes.byref(op), out.data_ptr(), stream))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        s = e0.elapsed_time(e1) * 1e-3 / 20
        alg = T * B * Lp * (D * 4 + 8) + T * B * (D * 4 + 8)
        res[f"alpha{alpha}"] = {"ms": s * 1e3, "alg_GBps": alg / s / 1e9, "frac": alg / s / 8e12}
    print(json.dumps({"alloc": tag, **extra, "tables": T, "GB": T * table_bytes / 1e9, **res}), flush=True)


slab = torch.empty(T * table_bytes, dtype=torch.uint8, device=dev)
run(slab.data_ptr(), "torch/hipMalloc", {"va_mod_1GiB_MiB": (slab.data_ptr() % (1 << 30)) >> 20, "va_mod_2MiB": slab.data_ptr() % (1 << 21)})
del slab
torch.cuda.empty_cache()
class CAI:  # zero-copy torch view of foreign device memory
    def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

for align_gb, chunk_gb in ((a.align_gb, a.chunk_gb), (2.0 / 1024, 0.0), (a.align_gb, 4.0)):
    try:
        p, g, size = vmm_alloc(T * table_bytes, int(align_gb * (1 << 30)), chunk_gb * (1 << 30))
        run(p, "hip VMM", {"align_GiB": align_gb, "chunk_GiB": chunk_gb, "granularity": g, "va_mod_1GiB_MiB": (p % (1 << 30)) >> 20})
        try:
            v = torch.as_tensor(CAI(p, 1 << 20), device=dev)
            print(json.dumps({"torch_view": True, "same_ptr": v.data_ptr() == p, "sum_finite": bool(torch.isfinite(v.sum()))}), flush=True)
        except Exception as e:
            print(json.dumps({"torch_view": False, "error": str(e)[:200]}), flush=True)
    except Exception as e:
        print(json.dumps({"alloc": "hip VMM", "align_GiB": align_gb, "chunk_GiB": chunk_gb, "error": str(e)}), flush=True)
