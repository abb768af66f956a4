"""The C-ABI library loads on a CPU-only host and exports every symbol include/param_amd.h
declares (no compute calls here: those need a GPU)."""
import ctypes
import os
import re

import pytest

from param_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    """(product symbols, alternates-only symbols): what include/param_amd.h declares outside / inside `#ifdef PM_ALTERNATES`"""
    src = open(os.path.join(ROOT, "include", "param_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    alt_blocks = re.findall(r"#ifdef PM_ALTERNATES(.*?)#endif", src, flags=re.S)
    product = re.sub(r"#ifdef PM_ALTERNATES.*?#endif", "", src, flags=re.S)
    names = lambda text: sorted(set(re.findall(r"\b(pm_[a-z_0-9]+)\s*\(", text)))        # noqa: E731
    return names(product), names("\n".join(alt_blocks))


def test_header_symbols_match_binding_list():
    product, alternates = _declared()
    assert product == sorted(_lib.EXPORTED_SYMBOLS)
    assert alternates == sorted(_lib.ALTERNATE_SYMBOLS) and not set(product) & set(alternates)


def test_library_loads_and_exports_every_symbol():
    L = _lib.load()
    product, alternates = _declared()
    for name in product:
        assert hasattr(L, name), name
    assert L.pm_abi_version() == _lib.PM_ABI_VERSION
    assert b"gfx950" in L.pm_build_info()
    A = _lib.load_alternates()                              # tests / tools build: everything, plus the alternates
    for name in product + alternates:
        assert hasattr(A, name), name
    assert A.pm_abi_version() == _lib.PM_ABI_VERSION


def test_product_library_holds_the_winning_path_only():
    """Round 6: rocPRIM's radix sort, round 2's LSD sort and the atomic backward are measured alternatives and cross-checks -- they
    live in libparam_amd_alt.so (`make alt`, -DPM_ALTERNATES).  The product library exports none of their entry points, contains no
    rocPRIM symbol and none of their kernels, and refuses to be switched to them."""
    import subprocess

    L = _lib.load()
    for name in _lib.ALTERNATE_SYMBOLS:
        assert not hasattr(L, name), name
    dyn = subprocess.run(["nm", "-DC", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "rocprim" not in dyn.lower()
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"7rocprim", b"rs_scatter_kernel", b"17embbag_bwd_kernel", b"build_keys_kernel"):       # (mangled names: code, not messages)
        assert needle not in blob, needle
    assert needle in open(_lib.ALT_LIB_PATH, "rb").read()                   # (the probe does find them where they are)
    for impl in (1, 2):
        assert L.pm_set_backward_tuning(impl, -1, -1, -1) == _lib.PM_ERR_UNSUPPORTED and b"alternates build" in L.pm_last_error()
    assert L.pm_set_backward_tuning(-1, -1, -1, -1) == _lib.PM_OK


def test_struct_layout_matches_header():
    # 4 x int32, 4 x int64, 4 pointers, int64, 3 pointers, int64 = 16 + 32 + 32 + 8 + 24 + 8; ABI v6: + 2 x int32 + int64 (blocked layouts)
    # ABI v7: + 2 x int32 (min_dim, reserved)
    assert ctypes.sizeof(_lib.pm_embbag_batch) == 144
    assert _lib.pm_embbag_batch.min_dim.offset == 136
    assert _lib.pm_embbag_batch.table_group.offset == 120 and _lib.pm_embbag_batch.grad_block_shift.offset == 124
    assert _lib.pm_embbag_batch.grad_block_extra.offset == 128
    assert _lib.pm_embbag_batch.fixed_pooling.offset == 112
    assert _lib.pm_embbag_batch.tables.offset == 48
    assert _lib.pm_embbag_batch.out_stride.offset == 80
    assert _lib.pm_embbag_batch.per_sample_weights.offset == 104


def test_argument_validation_without_gpu():
    """Host-side validation paths return error codes before any HIP call."""
    L = _lib.load()
    op = _lib.pm_embbag_batch()
    assert L.pm_embbag_fwd(ctypes.byref(op), None, None) == _lib.PM_ERR_INVALID
    assert b"num_tables" in L.pm_last_error()
    op.num_tables, op.weight_dtype, op.index_dtype, op.max_dim = 1, _lib.PM_F32, _lib.PM_I64, 6
    op.tables = op.rows = op.dims = op.out_offsets = 8  # non-null dummies, never dereferenced on the host
    assert L.pm_embbag_fwd(ctypes.byref(op), None, None) == _lib.PM_ERR_UNSUPPORTED
    assert b"multiple of 4" in L.pm_last_error()
    op.max_dim, op.index_dtype = 8, 99
    assert L.pm_embbag_fwd(ctypes.byref(op), None, None) == _lib.PM_ERR_INVALID
    op.index_dtype, op.batch, op.bag_begin, op.bag_count = _lib.PM_I64, 4, 2, 3
    assert L.pm_embbag_fwd(ctypes.byref(op), None, None) == _lib.PM_ERR_INVALID
    assert b"bag_begin" in L.pm_last_error()
    assert L.pm_set_tuning(5, 0, -1, -1) == _lib.PM_ERR_INVALID
    assert L.pm_set_tuning(0, 0, -1, -1) == _lib.PM_OK
    assert L.pm_set_backward_tuning(3, 0, 0, -1) == _lib.PM_ERR_INVALID and L.pm_set_sort_tuning(4) == _lib.PM_ERR_INVALID and L.pm_set_sort_tuning(-1) == _lib.PM_OK and L.pm_set_backward_tuning(-1, -1, -1, -1) == _lib.PM_OK
    in_b = ctypes.c_int32(-1)
    A = _lib.load_alternates()                                    # round 2's sort: the alternates build
    assert A.pm_radix_sort_pairs(None, None, None, None, 0, None, 4, 0, 24, 0, None, 0, ctypes.byref(in_b), None) == _lib.PM_OK
    assert in_b.value == 1                                        # 3 passes: the result would be in the b buffers
    assert A.pm_radix_sort_pairs(None, None, None, None, 10, None, 3, 0, 24, 0, None, 0, ctypes.byref(in_b), None) == _lib.PM_ERR_INVALID
    assert A.pm_radix_sort_pairs(None, None, None, None, 8192, None, 4, 0, 24, 1000, None, 0, ctypes.byref(in_b), None) == _lib.PM_ERR_INVALID
    assert b"segment_len" in A.pm_last_error()
    assert A.pm_radix_sort_scratch_bytes(1 << 20) > 0 and A.pm_radix_sort_scratch_bytes(-1) == _lib.PM_ERR_INVALID
    assert L.pm_fill_random(None, -1, _lib.PM_F32, 0, 0.0, 1.0, 0, None) == _lib.PM_ERR_INVALID
    # empty request: nothing to launch, succeeds without a device
    op.batch = op.bag_begin = op.bag_count = 0
    assert L.pm_embbag_fwd(ctypes.byref(op), None, None) == _lib.PM_OK
    assert L.pm_embbag_fwd_split(ctypes.byref(op), None, None) == _lib.PM_OK
    # the sorted backward carries positions and bags in 32 bits: 2^32 lookups (or a 2^31-row table) are refused, not wrapped
    op.batch, op.bag_count, op.num_indices, op.indices, op.offsets = 4, 4, 1 << 32, 8, 8
    assert L.pm_embbag_bwd_sorted_workspace(ctypes.byref(op), 1000) == _lib.PM_ERR_UNSUPPORTED
    assert b"2^32" in L.pm_last_error()
    op.num_indices = 100
    assert L.pm_embbag_bwd_sorted_workspace(ctypes.byref(op), (1 << 31) + 1) == _lib.PM_ERR_INVALID
    opt = _lib.pm_rowwise_adagrad(0.01, 1e-8, 0.0, 7, 0, 0, 0)                      # unknown weight-decay mode
    assert L.pm_embbag_bwd_sorted_adagrad_ex(ctypes.byref(op), 8, 8, _lib.PM_F32, 8, ctypes.byref(opt), 1000, 8, 1 << 40,
                                             None) == _lib.PM_ERR_INVALID
    assert b"weight_decay_mode" in L.pm_last_error()
    assert L.pm_dlrm_regroup(8, 8, 65, 64, 4, 8, 8, 8, None) == _lib.PM_ERR_UNSUPPORTED   # world * tables > 4096
    # ABI v6: the fused backward validates like its two halves (nothing is launched before the arguments are accepted) ...
    assert L.pm_embbag_bwd_fused(None, 8, 8, _lib.PM_F32, 1.0, 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    assert L.pm_embbag_bwd_fused(ctypes.byref(op), 8, 8, _lib.PM_F32, 1.0, 1000, None, 0, None) == _lib.PM_ERR_INVALID
    assert b"workspace" in L.pm_last_error()
    assert L.pm_embbag_bwd_fused_adagrad(ctypes.byref(op), 8, 8, _lib.PM_F32, 8, ctypes.byref(opt), 1000, None, 0, None) == _lib.PM_ERR_INVALID
    #  ... including what only the APPLY half reads: a NULL gradient, a bad destination dtype, missing optimizer state --
    # refused before the sort half could launch anything (a valid workspace pointer is never dereferenced on the host)
    assert L.pm_embbag_bwd_fused(ctypes.byref(op), None, 8, _lib.PM_F32, 1.0, 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    assert b"grad" in L.pm_last_error()
    assert L.pm_embbag_bwd_fused(ctypes.byref(op), 8, 8, 7, 1.0, 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    assert b"dtype" in L.pm_last_error()
    good = _lib.pm_rowwise_adagrad(0.01, 1e-8, 0.0, _lib.PM_WD_NONE, 0, 0, 0)
    assert L.pm_embbag_bwd_fused_adagrad(ctypes.byref(op), 8, 8, _lib.PM_F32, None, ctypes.byref(good), 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    assert b"momentum" in L.pm_last_error()
    assert L.pm_embbag_bwd_fused_adagrad(ctypes.byref(op), 8, 8, _lib.PM_F32, 8, None, 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    #  a blocked gradient needs a block of at least two bags that tiles the batch
    op.grad_block_extra = 64
    assert L.pm_embbag_bwd_fused(ctypes.byref(op), 8, 8, _lib.PM_F32, 1.0, 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    assert b"grad_block_shift >= 1" in L.pm_last_error()
    op.grad_block_shift = 3                                                         # blocks of 8 bags, batch 4
    assert L.pm_embbag_bwd_fused(ctypes.byref(op), 8, 8, _lib.PM_F32, 1.0, 1000, 8, 1 << 40, None) == _lib.PM_ERR_INVALID
    assert b"multiple of 2^grad_block_shift" in L.pm_last_error()
    op.grad_block_shift, op.grad_block_extra = 0, 0
    op.min_dim = 6                                                                   # ABI v7: min_dim is a width like max_dim
    assert L.pm_embbag_fwd(ctypes.byref(op), 8, None) == _lib.PM_ERR_INVALID and b"min_dim" in L.pm_last_error()
    op.min_dim = 0
    op.num_indices = 0                                                               # ... and an empty request succeeds without a device
    assert L.pm_embbag_bwd_fused(ctypes.byref(op), None, None, _lib.PM_F32, 1.0, 1000, None, 0, None) == _lib.PM_OK
    op.num_indices = 100


def test_graft_entry_build_runs():
    """the driver's build check: compiles (a no-op when up to date), loads the library, checks the ABI version"""
    import __graft_entry__ as g

    g.build()


def _build_c_demo(tmp_path):
    """examples/c_abi/embbag_demo.c: plain C11 + gcc against include/param_amd.h and libparam_amd.so (no C++, no torch)"""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc / ROCm headers not available")
    exe = str(tmp_path / "embbag_demo")
    libdir = os.path.join(root, "param_amd")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi", "embbag_demo.c"), "-o", exe,
           "-L" + libdir, "-lparam_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_abi_demo_compiles_as_plain_c(tmp_path):
    """the header is C-clean and every entry point the demo uses links (no GPU needed to build)"""
    assert os.path.exists(_build_c_demo(tmp_path))


@pytest.mark.gpu
def test_c_abi_demo_runs_on_gpu(tmp_path):
    """the same program on the device: forward, sorted (sort-aside) and fused (ABI v6) backward bit-exact against its own sequential host loops"""
    import subprocess

    r = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "forward bit-exact, sorted and fused backward bit-exact" in r.stdout


def _plan(L, T, B, Lp, max_rows, phases=1, fixed=True, slice_=None, weighted=False):
    op = _lib.pm_embbag_batch()
    op.num_tables, op.weight_dtype, op.index_dtype, op.max_dim = T, _lib.PM_F32, _lib.PM_I64, 128
    op.batch, op.num_indices = B, T * B * Lp
    op.bag_begin, op.bag_count = (0, B) if slice_ is None else slice_
    op.tables = op.rows = op.dims = op.out_offsets = op.indices = op.offsets = 8     # never dereferenced on the host
    op.per_sample_weights = 8 if weighted else None
    op.out_stride = T * 128
    op.fixed_pooling = Lp if fixed else 0
    buf = ctypes.create_string_buffer(512)
    assert L.pm_embbag_sort_plan(ctypes.byref(op), max_rows, phases, buf, 512) == _lib.PM_OK, L.pm_last_error()
    return dict(kv.split("=") for kv in buf.value.decode().split())


def test_sort_plan_decisions_on_the_host():
    """which layout the sorted backward picks for a request is a host-side decision (make_plan): pinned here without a GPU"""
    L = _lib.load()
    assert L.pm_set_backward_tuning(-1, -1, -1, -1) == _lib.PM_OK and L.pm_set_sort_tuning(-1) == _lib.PM_OK
    # default (round 3): ONE plan for every request -- segments, per-table pooling and the pair count are device-side
    for kw in ({}, {"fixed": False}, {"slice_": (100, 50)}, {"weighted": True}):
        seg = _plan(L, 48, 8192, 20, 10_000_000, **kw)
        assert seg["sort"] == "seg" and seg["segmented"] == "1" and seg["segments"] == "device" and seg["pooling"] == "device"
        assert seg["xcd"] == "1" and seg["key_bytes"] == "4" and seg["rbits"] == "24" and seg["kbits"] == "30" and seg["sort_bits"] == "24"
        assert seg["fused_keys"] == ("0" if kw.get("weighted") else "1")
    assert _plan(L, 26, 8192, 8, 40_000_000)["rbits"] == "26"
    for mode, passes, in_b in ((0, "3", "1"), (1, "1", "1"), (2, "1", "1"), (3, "3", "1")):
        assert L.pm_set_sort_tuning(mode) == _lib.PM_OK
        seg = _plan(L, 48, 8192, 20, 10_000_000)
        assert seg["mode"] == str(mode) and seg["passes"] == passes and seg["result_in_b"] == in_b
        assert seg["local"] == ("1" if mode in (1, 2) else "0") and seg["lookback"] == ("1" if mode == 0 else "0")
    assert L.pm_set_sort_tuning(0) == _lib.PM_OK
    criteo = _plan(L, 26, 8192, 8, 40_000_000)                    # 26 row bits: 9-bit digits save a pass (3 instead of 4)
    assert criteo["passes"] == "3" and criteo["radix_bits"] == "9" and criteo["result_in_b"] == "1"
    assert _plan(L, 26, 8192, 8, 10_000_000)["radix_bits"] == "8" and _plan(L, 4, 512, 8, 200_000)["passes"] == "2"
    assert _plan(L, 1024, 64, 64, 1 << 30)["key_bytes"] == "8"
    assert L.pm_set_sort_tuning(-1) == _lib.PM_OK
    # round 2's host-side plans (sort_impl 2), kept as the measured alternative -- in the alternates build
    L = _lib.load_alternates()
    assert L.pm_set_backward_tuning(2, -1, -1, -1) == _lib.PM_OK
    bench = _plan(L, 48, 8192, 20, 10_000_000)                       # the benchmark step
    assert bench["sort"] == "own" and bench["key_bytes"] == "4" and bench["rbits"] == "24" and bench["kbits"] == "30"
    assert bench["segmented"] == "1" and bench["seg_len"] == str(8192 * 20) and bench["passes"] == "3" and bench["sort_bits"] == "24"
    assert bench["apply_seg_tiles"] == "160" and bench["xcd"] == "1" and bench["phases"] == "1" and bench["result_in_b"] == "1"
    assert bench["fused_keys"] == "1"                                 # no key-building kernel: the first radix pass reads the indices
    assert _plan(L, 48, 8192, 20, 10_000_000, weighted=True)["fused_keys"] == "0"      # weights need the position as the value
    ragged = _plan(L, 48, 8192, 20, 10_000_000, fixed=False)          # no fixed-pooling claim: all key bits, global, linear tiles
    assert ragged["segmented"] == "0" and ragged["sort_bits"] == "30" and ragged["passes"] == "4" and ragged["xcd"] == "0"
    assert ragged["apply_seg_tiles"] == "0" and ragged["result_in_b"] == "0" and ragged["fused_keys"] == "0"
    sliced = _plan(L, 48, 8192, 20, 10_000_000, slice_=(100, 50))     # batch slice: padding keys must sort last
    assert sliced["sliced"] == "1" and sliced["sort_bits"] == "31" and sliced["segmented"] == "0" and sliced["xcd"] == "0"
    odd = _plan(L, 5, 100, 7, 1000)                                   # fixed pooling, nothing tile-aligned
    assert odd["segmented"] == "0" and odd["apply_seg_tiles"] == "0" and odd["sort_bits"] == str(10 + 3)
    one = _plan(L, 1, 8192, 20, 10_000_000)                           # a single table: segments yes, XCD mapping pointless
    assert one["segmented"] == "1" and one["xcd"] == "0" and one["apply_seg_tiles"] == "0"
    wide = _plan(L, 1024, 64, 64, 1 << 30)                            # 30 + 10 key bits: 8-byte keys
    assert wide["key_bytes"] == "8" and wide["kbits"] == "40"
    # two bag phases only on request AND with the knob: default max_phases = 1
    assert _plan(L, 48, 8192, 20, 10_000_000, phases=2)["phases"] == "1"
    assert L.pm_set_backward_tuning(2, -1, -1, 2) == _lib.PM_OK
    two = _plan(L, 48, 8192, 20, 10_000_000, phases=2)
    assert two["phases"] == "2" and two["hbits"] == "1" and two["kbits"] == "31" and two["seg_len"] == str(4096 * 20)
    assert two["apply_seg_tiles"] == "80" and two["key_bytes"] == "4" and two["fused_keys"] == "0"
    assert _plan(L, 48, 8192, 20, 10_000_000, phases=1)["phases"] == "1"
    # row order: only the row bits, no segments, no XCD mapping; rocPRIM: no segments either
    assert L.pm_set_backward_tuning(2, 0, -1, -1) == _lib.PM_OK
    row = _plan(L, 48, 8192, 20, 10_000_000)
    assert row["sort_bits"] == "24" and row["segmented"] == "0" and row["xcd"] == "0"
    assert L.pm_set_backward_tuning(1, -1, -1, -1) == _lib.PM_OK
    rp = _plan(L, 48, 8192, 20, 10_000_000)
    assert rp["sort"] == "rocprim" and rp["segmented"] == "0" and rp["sort_bits"] == "30" and rp["xcd"] == "1" and rp["result_in_b"] == "1"
    assert L.pm_set_backward_tuning(-1, -1, -1, -1) == _lib.PM_OK
