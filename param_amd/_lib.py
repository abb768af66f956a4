"""ctypes binding of libparam_amd.so (C ABI: include/param_amd.h).

There is NO fallback: if the shared library is missing or a symbol is absent the
import of the binding raises, and every op raises on a non-ROCm tensor.  Build the
library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C param_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import threading

PM_F32, PM_BF16, PM_F16, PM_I32, PM_I64 = 0, 1, 2, 10, 11
PM_OK, PM_ERR_INVALID, PM_ERR_UNSUPPORTED, PM_ERR_HIP, PM_ERR_INDEX = 0, -1, -2, -3, -4
PM_ABI_VERSION = 7
PM_WD_NONE, PM_WD_L2, PM_WD_DECOUPLE = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PARAM_AMD_LIB") or os.path.join(_HERE, "libparam_amd.so")   # PARAM_AMD_LIB: kernel experiments only
# the ALTERNATES build (make -C param_amd/csrc alt, -DPM_ALTERNATES): the product library's sources plus the measured alternatives and
# cross-checks -- atomic backward, round 2's LSD sort, rocPRIM's radix sort.  tests/ and tools/ only (use_alternates() below).
ALT_LIB_PATH = os.path.join(_HERE, "libparam_amd_alt.so")

# every symbol include/param_amd.h declares (tests/test_capi_symbols.py parses the header
# and checks this list and the loaded library against it)
EXPORTED_SYMBOLS = (
    "pm_abi_version",
    "pm_build_info",
    "pm_last_error",
    "pm_embbag_fwd",
    "pm_embbag_fwd_split",
    "pm_embbag_bwd_sorted_workspace",
    "pm_embbag_sort_indices",
    "pm_embbag_sort_indices_ex",
    "pm_embbag_sort_plan",
    "pm_embbag_sorted_pairs",
    "pm_embbag_fwd_quantized",
    "pm_embbag_check_ex",
    "pm_rows_quantized_bytes",
    "pm_rows_quantize",
    "pm_rows_dequantize",
    "pm_embbag_bwd_sorted",
    "pm_embbag_bwd_sorted_adagrad",
    "pm_embbag_bwd_sorted_adagrad_ex",
    "pm_embbag_bwd_fused",
    "pm_embbag_bwd_fused_adagrad",
    "pm_dlrm_regroup",
    "pm_embbag_check",
    "pm_fill_random",
    "pm_set_tuning",
    "pm_set_forward_tuning",
    "pm_set_backward_tuning",
    "pm_set_sort_tuning",
    "pm_embbag_sort_status",
    "pm_set_hybrid_tuning",
    "pm_set_hybrid_rest",
    "pm_set_hybrid_min_tiles",
)


class pm_rowwise_adagrad(ctypes.Structure):
    """Mirror of ``struct pm_rowwise_adagrad`` (include/param_amd.h)."""

    _fields_ = [
        ("lr", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("weight_decay", ctypes.c_float),
        ("weight_decay_mode", ctypes.c_int32),
        ("stochastic_rounding", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
    ]


class pm_sort_status(ctypes.Structure):
    """Mirror of ``struct pm_sort_status`` (include/param_amd.h)."""

    _fields_ = [
        ("lookback_fallbacks", ctypes.c_uint32),
        ("pairs_sorted", ctypes.c_uint32),
        ("hybrid_tables", ctypes.c_uint32),
        ("hybrid_launched", ctypes.c_uint32),
        ("lds_pairs", ctypes.c_uint32),
        ("lds_tables", ctypes.c_uint32),
    ]


class pm_embbag_batch(ctypes.Structure):
    """Mirror of ``struct pm_embbag_batch`` (include/param_amd.h)."""

    _fields_ = [
        ("num_tables", ctypes.c_int32),
        ("weight_dtype", ctypes.c_int32),
        ("index_dtype", ctypes.c_int32),
        ("max_dim", ctypes.c_int32),
        ("batch", ctypes.c_int64),
        ("num_indices", ctypes.c_int64),
        ("bag_begin", ctypes.c_int64),
        ("bag_count", ctypes.c_int64),
        ("tables", ctypes.c_void_p),
        ("rows", ctypes.c_void_p),
        ("dims", ctypes.c_void_p),
        ("out_offsets", ctypes.c_void_p),
        ("out_stride", ctypes.c_int64),
        ("indices", ctypes.c_void_p),
        ("offsets", ctypes.c_void_p),
        ("per_sample_weights", ctypes.c_void_p),
        ("fixed_pooling", ctypes.c_int64),
        ("table_group", ctypes.c_int32),
        ("grad_block_shift", ctypes.c_int32),
        ("grad_block_extra", ctypes.c_int64),
        ("min_dim", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
    ]


class ParamAmdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libparam_amd error {code}: {msg}")
        self.code = code


# ... and what the header declares under #ifdef PM_ALTERNATES (libparam_amd_alt.so exports these as well)
ALTERNATE_SYMBOLS = (
    "pm_embbag_bwd",
    "pm_radix_sort_scratch_bytes",
    "pm_radix_sort_pairs",
)

_lock = threading.Lock()
_lib = None          # the library load() hands out: the product library, or the alternates build inside use_alternates()
_product = None
_alt = None


def _open(path: str, alternates: bool) -> ctypes.CDLL:
    """dlopen one build of the library, check its exports against the header's list and give every entry point its signature"""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is not built. There is no CPU/eager fallback for the MI355X embedding path: run "
            f"`make -C param_amd/csrc{' alt' if alternates else ''}` (hipcc --offload-arch=gfx950) first.")
    L = ctypes.CDLL(path)
    want = EXPORTED_SYMBOLS + (ALTERNATE_SYMBOLS if alternates else ())
    missing = [s for s in want if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{path} lacks symbols {missing}; rebuild it")
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.pm_abi_version.restype = ctypes.c_int
    L.pm_abi_version.argtypes = []
    L.pm_build_info.restype = ctypes.c_char_p
    L.pm_build_info.argtypes = []
    L.pm_last_error.restype = ctypes.c_char_p
    L.pm_last_error.argtypes = []
    L.pm_embbag_fwd.restype = ctypes.c_int
    L.pm_embbag_fwd.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp]
    L.pm_embbag_fwd_split.restype = ctypes.c_int
    L.pm_embbag_fwd_split.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp]
    L.pm_embbag_bwd_sorted_workspace.restype = ctypes.c_int64
    L.pm_embbag_bwd_sorted_workspace.argtypes = [ctypes.POINTER(pm_embbag_batch), i64]
    L.pm_embbag_sort_indices.restype = ctypes.c_int
    L.pm_embbag_sort_indices.argtypes = [ctypes.POINTER(pm_embbag_batch), i64, vp, i64, vp]
    L.pm_embbag_sort_indices_ex.restype = ctypes.c_int
    L.pm_embbag_sort_indices_ex.argtypes = [ctypes.POINTER(pm_embbag_batch), i64, i32, vp, i64, vp]
    L.pm_embbag_fwd_quantized.restype = ctypes.c_int
    L.pm_embbag_fwd_quantized.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, i32, vp]
    L.pm_embbag_check_ex.restype = ctypes.c_int
    L.pm_embbag_check_ex.argtypes = [ctypes.POINTER(pm_embbag_batch), i32, vp, vp]
    L.pm_rows_quantized_bytes.restype = i64
    L.pm_rows_quantized_bytes.argtypes = [i64, i32, i32]
    L.pm_rows_quantize.restype = ctypes.c_int
    L.pm_rows_quantize.argtypes = [vp, i64, i32, i32, vp, vp]
    L.pm_rows_dequantize.restype = ctypes.c_int
    L.pm_rows_dequantize.argtypes = [vp, i64, i32, i32, vp, vp]
    L.pm_embbag_sort_plan.restype = ctypes.c_int
    L.pm_embbag_sort_plan.argtypes = [ctypes.POINTER(pm_embbag_batch), i64, i32, ctypes.c_char_p, i32]
    L.pm_embbag_bwd_sorted.restype = ctypes.c_int
    L.pm_embbag_bwd_sorted.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp, i32, ctypes.c_float, i64, vp,
                                       i64, vp]
    L.pm_embbag_bwd_sorted_adagrad.restype = ctypes.c_int
    L.pm_embbag_bwd_sorted_adagrad.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp, i32, vp, ctypes.c_float,
                                               ctypes.c_float, i64, vp, i64, vp]
    L.pm_embbag_bwd_sorted_adagrad_ex.restype = ctypes.c_int
    L.pm_embbag_bwd_sorted_adagrad_ex.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp, i32, vp,
                                                  ctypes.POINTER(pm_rowwise_adagrad), i64, vp, i64, vp]
    L.pm_embbag_bwd_fused.restype = ctypes.c_int
    L.pm_embbag_bwd_fused.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp, i32, ctypes.c_float, i64, vp, i64, vp]
    L.pm_embbag_bwd_fused_adagrad.restype = ctypes.c_int
    L.pm_embbag_bwd_fused_adagrad.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp, i32, vp,
                                              ctypes.POINTER(pm_rowwise_adagrad), i64, vp, i64, vp]
    L.pm_dlrm_regroup.restype = ctypes.c_int
    L.pm_dlrm_regroup.argtypes = [vp, vp, i32, i32, i64, vp, vp, vp, vp]
    L.pm_embbag_check.restype = ctypes.c_int
    L.pm_embbag_check.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp]
    L.pm_fill_random.restype = ctypes.c_int
    L.pm_fill_random.argtypes = [vp, i64, i32, i32, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, vp]
    L.pm_set_tuning.restype = ctypes.c_int
    L.pm_set_tuning.argtypes = [i32, i32, i32, i32]
    L.pm_set_forward_tuning.restype = ctypes.c_int
    L.pm_set_forward_tuning.argtypes = [i32, i32, i32]
    L.pm_set_backward_tuning.restype = ctypes.c_int
    L.pm_set_backward_tuning.argtypes = [i32, i32, i32, i32]
    L.pm_embbag_sorted_pairs.restype = ctypes.c_int
    L.pm_embbag_sorted_pairs.argtypes = [ctypes.POINTER(pm_embbag_batch), i64, vp, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                         ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.pm_set_sort_tuning.restype = ctypes.c_int
    L.pm_set_sort_tuning.argtypes = [i32]
    L.pm_embbag_sort_status.restype = ctypes.c_int
    L.pm_embbag_sort_status.argtypes = [ctypes.POINTER(pm_embbag_batch), i64, vp, ctypes.POINTER(pm_sort_status), vp]
    L.pm_set_hybrid_tuning.restype = ctypes.c_int
    L.pm_set_hybrid_tuning.argtypes = [i32, i64]
    L.pm_set_hybrid_rest.restype = ctypes.c_int
    L.pm_set_hybrid_rest.argtypes = [i32]
    L.pm_set_hybrid_min_tiles.restype = ctypes.c_int
    L.pm_set_hybrid_min_tiles.argtypes = [i32]
    if alternates:
        L.pm_embbag_bwd.restype = ctypes.c_int
        L.pm_embbag_bwd.argtypes = [ctypes.POINTER(pm_embbag_batch), vp, vp, i32, ctypes.c_float, vp]
        L.pm_radix_sort_scratch_bytes.restype = ctypes.c_int64
        L.pm_radix_sort_scratch_bytes.argtypes = [i64]
        L.pm_radix_sort_pairs.restype = ctypes.c_int
        L.pm_radix_sort_pairs.argtypes = [vp, vp, vp, vp, i64, vp, i32, i32, i32, i64, vp, i64, ctypes.POINTER(ctypes.c_int32), vp]
    if L.pm_abi_version() != PM_ABI_VERSION:
        raise ImportError(f"{path}: ABI version {L.pm_abi_version()} != {PM_ABI_VERSION}")
    return L


def load() -> ctypes.CDLL:
    """The library the package calls: libparam_amd.so, loaded once; raises ImportError (loudly) if it is not built."""
    global _lib, _product
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if _product is None:
                _product = _open(LIB_PATH, alternates=False)
            _lib = _product
    return _lib


def load_alternates() -> ctypes.CDLL:
    """libparam_amd_alt.so (tests / tools): every product entry point plus ALTERNATE_SYMBOLS.  Its knobs and its records of sorted
    workspaces are its own: a sort issued through one library is applied through the same one."""
    global _alt
    with _lock:
        if _alt is None:
            _alt = _open(ALT_LIB_PATH, alternates=True)
    return _alt


class use_alternates:
    """``with _lib.use_alternates():`` -- every call of the package goes to the alternates build inside the block (tests that cross-check
    the product path against round 2's sort / rocPRIM / the atomic kernel; tools that time them).  Not re-entrant across threads."""

    def __enter__(self):
        global _lib
        load()
        self._prev = _lib
        _lib = load_alternates()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def check(rc: int) -> None:
    if rc != PM_OK:
        raise ParamAmdError(rc, load().pm_last_error().decode())


def set_tuning(unroll: int = 0, bags_per_block: int = 0, xcd_affine: int = -1, nt_loads: int = -1) -> None:
    check(load().pm_set_tuning(unroll, bags_per_block, xcd_affine, nt_loads))


def set_forward_tuning(stage_out: int = -1, flat_grid: int = -1, flat_target: int = -1) -> None:
    """``pm_set_forward_tuning``: stage_out 1 (default) = LDS-staged output burst per tile, 0 = one row store per finished bag;
    flat_grid (flat-walk forward: short-bag / mixed-dim requests) 1 (default) = workgroups walking the tile order, 0 = one workgroup
    per (table, smallest tile), N > 1 = exactly N workgroups; flat_target = lookups per flat-walk tile (default 256)"""
    check(load().pm_set_forward_tuning(stage_out, flat_grid, flat_target))


def set_backward_tuning(sort_impl: int = -1, order: int = -1, xcd_affine: int = -1, max_phases: int = -1) -> None:
    """sorted-backward knobs (``pm_set_backward_tuning``): sort_impl 0 segmented sort (segments / pooling established on the
    device) / 1 rocPRIM / 2 round 2's own LSD sort with host-side plans (1 and 2: inside ``use_alternates()`` only -- the product
    library refuses them), order 1 (table, row) / 0 (row, table) and max_phases 2 / 1 for sort_impl 1 and 2, xcd_affine 1 / 0;
    -1 = default.  Read when a request is sorted; its apply follows."""
    global _sort_impl
    check(load().pm_set_backward_tuning(sort_impl, order, xcd_affine, max_phases))
    _sort_impl = sort_impl


_sort_impl = -1


def needs_pooling_hint() -> bool:
    """True when the selected key sort is one of the alternatives that read ``pm_embbag_batch.fixed_pooling`` (sort_impl 1 /
    2, by knob or ``PARAM_AMD_SORT``); the default segmented sort establishes pooling on the device and the Python layer then
    neither computes nor caches a verdict about the offsets' contents"""
    if _alt is None or _lib is not _alt:
        return False                      # the product library has the segmented sort and nothing else
    if _sort_impl >= 0:
        return _sort_impl != 0
    return os.environ.get("PARAM_AMD_SORT", "") in ("rocprim", "legacy")


def set_sort_tuning(mode: int = -1) -> None:
    """``pm_set_sort_tuning``: segmented sort mode 0 LSD passes, one look-back kernel per pass / 1 low-digit partition +
    bucket-local LDS sort / 2 top-digit partition + local sort / 3 LSD passes of three kernels each; -1 = default"""
    check(load().pm_set_sort_tuning(mode))


def set_hybrid_tuning(enable: int = -1, lookback_spin_cap: int = 0) -> None:
    """``pm_set_hybrid_tuning``: the hybrid backward (rows looked up once are applied bag-major, only the repeats are sorted).
    enable 0 off / 1 on (default; tables classified on the device at every sort) / 2 every structurally eligible table (tests);
    lookback_spin_cap > 0 lowers the number of polls after which a look-back walk counts for its predecessor (tests)."""
    check(load().pm_set_hybrid_tuning(enable, lookback_spin_cap))


def set_hybrid_min_tiles(tiles: int = -1) -> None:
    """``pm_set_hybrid_min_tiles``: the hybrid backward is offered to requests of at least this many bag-major workgroups
    (``num_tables * ceil(bag_count / 128)``): -1 = default (1024: below, a handful of workgroups pool while the chip idles and the
    sorted path is up to 4 x faster), 0 = no lower bound (tests that drive the hybrid kernels with small requests)."""
    check(load().pm_set_hybrid_min_tiles(tiles))


def set_hybrid_rest(mode: int = -1) -> None:
    """``pm_set_hybrid_rest``: 1 (default) = what the bag-major kernel leaves of a hybrid table (the flagged lookups, at most 8192 of
    them) is sorted in LDS and applied by ONE launch; 0 = the lists go through the key sort and the sorted apply (round 5)."""
    check(load().pm_set_hybrid_rest(mode))
