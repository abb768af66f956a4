"""Round 2's radix sort (param_amd/csrc/radix_sort.hip, C ABI pm_radix_sort_pairs) against numpy's stable argsort,
and the sorted backward under every sort / order / XCD-mapping setting against the CPU oracle.

The whole module runs on the ALTERNATES build (libparam_amd_alt.so, `make -C param_amd/csrc alt`): round 2's sort, rocPRIM's
radix sort and the `sort_impl` knob left the product library in round 6; here they are cross-checks of the product path (sort_impl 0
rows of the matrix below are the product's own sort, compiled from the same sources)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    import param_amd

    from param_amd import _lib

    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    param_amd.load_library()
    with _lib.use_alternates():
        yield
        param_amd.set_backward_tuning()


def _sort(keys: np.ndarray, begin: int, end: int, count=None, seg_len: int = 0):
    """runs pm_radix_sort_pairs with values = original positions; returns (keys_out, vals_out) of the first `count`"""
    from param_amd import _lib

    L = _lib.load()
    n = keys.size
    kb = keys.dtype.itemsize
    ka = torch.from_numpy(keys.view(np.int32 if kb == 4 else np.int64).copy()).to(DEV)
    kbuf = torch.full_like(ka, -1)
    va = torch.arange(n, dtype=torch.int32, device=DEV)
    vb = torch.full_like(va, -1)
    need = L.pm_radix_sort_scratch_bytes(n)
    assert need > 0
    scratch = torch.empty(need, dtype=torch.uint8, device=DEV)
    dcount = None if count is None else torch.tensor([count], dtype=torch.int32, device=DEV)
    in_b = ctypes.c_int32(-1)
    _lib.check(L.pm_radix_sort_pairs(ka.data_ptr(), kbuf.data_ptr(), va.data_ptr(), vb.data_ptr(), n,
                                     None if dcount is None else dcount.data_ptr(), kb, begin, end, seg_len, scratch.data_ptr(), need,
                                     ctypes.byref(in_b), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert in_b.value == (((end - begin) + 7) // 8) % 2
    ko, vo = (kbuf, vb) if in_b.value else (ka, va)
    m = n if count is None else min(count, n)
    return ko.cpu().numpy().view(keys.dtype)[:m], vo.cpu().numpy().view(np.uint32)[:m]


def _expect(keys, begin, end, count=None):
    m = keys.size if count is None else min(count, keys.size)
    k = keys[:m]
    if end == begin:
        return k, np.arange(m, dtype=np.uint32)
    digit = (k.astype(np.uint64) >> np.uint64(begin)) & np.uint64((1 << (end - begin)) - 1)
    perm = np.argsort(digit, kind="stable")
    return k[perm], perm.astype(np.uint32)


@pytest.mark.parametrize("n", [1, 63, 64, 4095, 4096, 4097, 100_003, 3_000_017])
def test_radix_sort_matches_stable_argsort_u32(n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    for begin, end in [(0, 24), (0, 30), (3, 17), (0, 1), (0, 0), (0, 32)]:
        ko, vo = _sort(keys, begin, end)
        ek, ev = _expect(keys, begin, end)
        assert np.array_equal(vo, ev), (n, begin, end)           # the permutation itself: stability included
        assert np.array_equal(ko, ek), (n, begin, end)


def test_radix_sort_skewed_equal_and_u64_keys():
    rng = np.random.default_rng(5)
    n = 1_200_001
    # Zipf-like: most keys in a few values (whole waves / tiles of one digit), a long tail
    z = np.minimum(rng.zipf(1.2, n), 1 << 23).astype(np.uint32)
    tables = np.repeat(np.arange(8, dtype=np.uint32), (n + 7) // 8)[:n]           # table-major, like the backward's keys
    keys = (tables << np.uint32(24)) | z
    for begin, end in [(0, 24), (0, 27)]:
        ko, vo = _sort(keys, begin, end)
        ek, ev = _expect(keys, begin, end)
        assert np.array_equal(vo, ev) and np.array_equal(ko, ek), (begin, end)
    same = np.full(70_000, 0x00ABCDEF, dtype=np.uint32)
    ko, vo = _sort(same, 0, 24)
    assert np.array_equal(vo, np.arange(same.size, dtype=np.uint32))              # all equal: nothing moves
    k64 = rng.integers(0, 1 << 62, 300_007, dtype=np.uint64)
    for begin, end in [(0, 41), (20, 64), (0, 64)]:
        ko, vo = _sort(k64, begin, end)
        ek, ev = _expect(k64, begin, end)
        assert np.array_equal(vo, ev) and np.array_equal(ko, ek), (begin, end)


def test_radix_sort_device_side_count():
    """the element count read from device memory: only the first `count` pairs are sorted, whatever n_max is"""
    rng = np.random.default_rng(9)
    keys = rng.integers(0, 1 << 24, 100_003, dtype=np.uint32)
    for count in (0, 1, 4096, 70_001, 100_003, 1 << 30):
        ko, vo = _sort(keys, 0, 24, count=count)
        ek, ev = _expect(keys, 0, 24, count=count)
        assert np.array_equal(vo, ev) and np.array_equal(ko, ek), count


def test_radix_sort_segmented():
    """segments of 8192 / 4096 pairs sorted independently: the order inside every segment is numpy's stable order, and no
    pair leaves its segment"""
    rng = np.random.default_rng(12)
    for seg, nseg in [(8192, 37), (4096, 5), (163840, 3)]:
        n = seg * nseg
        keys = rng.integers(0, 1 << 24, n, dtype=np.uint32) | (np.repeat(np.arange(nseg, dtype=np.uint32), seg) << np.uint32(24))
        keys[:seg // 2] = keys[0]                                # a run of equal keys inside segment 0
        ko, vo = _sort(keys, 0, 24, seg_len=seg)
        for s_ in range(nseg):
            sl = slice(s_ * seg, (s_ + 1) * seg)
            ek, ev = _expect(keys[sl], 0, 24)
            assert np.array_equal(ko[sl], ek), (seg, s_)
            assert np.array_equal(vo[sl], ev + np.uint32(s_ * seg)), (seg, s_)


@pytest.mark.parametrize("sort_impl,order,xcd,phases", [(0, 1, 0, 1), (0, 1, 1, 1),                                       # segmented sort (default)
                                                        (2, 0, 0, 1), (2, 1, 0, 1), (2, 1, 1, 1), (2, 1, 1, 2), (2, 1, 0, 2),   # round 2's own sort
                                                        (1, 0, 0, 1), (1, 1, 1, 1), (1, 1, 1, 2)])                               # rocPRIM
def test_sorted_backward_under_every_tuning(coracle, sort_impl, order, xcd, phases):
    """8 tables x 1024 bags x 16 lookups (per-table lookups = 16 tiles of 1024: the XCD-affine mapping engages), Zipf
    duplicates incl. rows past the exact-run limit: every (own | rocPRIM) x (row | table order) x XCD-mapping setting
    gives the oracle's bits on rows looked up <= 256 times and 1e-5 of an fp64 sum on the hot rows."""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.indices import zipf_indices

    param_amd.set_backward_tuning(sort_impl, order, xcd, phases)
    try:
        T, R, D, B, L = 8, 30000, 128, 1024, 16
        m = BatchedEmbeddingBagMI355([R] * T, D, device=DEV, init="normal", seed=3, fused_update=False)
        g = torch.Generator().manual_seed(11)
        idx = torch.cat([zipf_indices(1.15, R, B * L, 1, dedupe=False, generator=g) for _ in range(T)])
        idx[:600] = 5                                    # > 256 lookups of one row in table 0
        off = torch.arange(T * B + 1) * L
        grad = torch.randn(B, T * D, generator=g)
        W0 = [m.table(t).cpu().numpy().copy() for t in range(T)]
        m.scatter_add_(grad.to(DEV), idx.to(DEV), off.to(DEV), alpha=-0.1)
        ih, gh = idx.numpy(), grad.numpy()
        for t in range(T):
            s, e = t * B * L, (t + 1) * B * L
            gt = np.ascontiguousarray(gh[:, t * D:(t + 1) * D])
            exp = coracle.bwd_f32(W0[t].copy(), ih[s:e], np.arange(B, dtype=np.int64) * L, gt, None, alpha=-0.1)
            got = m.table(t).cpu().numpy()
            cnt = np.bincount(ih[s:e], minlength=R)
            cold = cnt <= 256
            assert np.array_equal(got[cold], exp[cold]), (t, sort_impl, order, xcd)
            truth = W0[t].astype(np.float64)
            np.add.at(truth, ih[s:e], -0.1 * gt.astype(np.float64)[np.repeat(np.arange(B), L)])
            mag = np.abs(W0[t]).astype(np.float64)
            np.add.at(mag, ih[s:e], 0.1 * np.abs(gt).astype(np.float64)[np.repeat(np.arange(B), L)])
            assert (np.abs(got - truth) <= 1e-5 * mag + 1e-30).all(), (t, sort_impl, order, xcd)
            if t == 0:
                assert (~cold).sum() >= 1
    finally:
        param_amd.set_backward_tuning()


def test_two_phase_apply_engages_and_adagrad_refuses_it(coracle):
    """fixed pooling + aligned sizes: the scatter-add sorts for two bag phases (pm_embbag_sort_indices_ex(phases=2)); the
    same request declared ragged (pooling=0) takes the general one-launch path -- both give the oracle's bits; the fused
    Adagrad refuses a two-phase sort; pm_embbag_check verifies a fixed-pooling claim."""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355, _lib
    from param_amd.embedding_bag import _sort_indices

    T, R, D, B, L = 4, 5000, 64, 512, 8          # (B / 2) * L = 2048: two apply tiles per (table, phase) segment
    param_amd.set_backward_tuning(sort_impl=2, max_phases=2)  # round 2's sort; the two-phase layout is opt-in there
    g = torch.Generator().manual_seed(4)
    idx = torch.randint(0, R, (T * B * L,), generator=g)
    idx[: 3 * L] = 7                             # one row looked up in bags 0..2 (lower half) ...
    idx[(B - 2) * L:(B - 2) * L + L] = 7         # ... and in bag B-2 (upper half): updated by both phases, in order
    off = torch.arange(T * B + 1) * L
    grad = torch.randn(B, T * D, generator=g)
    results = []
    for pooling in (None, 0):
        m = BatchedEmbeddingBagMI355([R] * T, D, device=DEV, init="normal", seed=9, fused_update=False)
        W0 = [m.table(t).cpu().numpy().copy() for t in range(T)]
        m.scatter_add_(grad.to(DEV), idx.to(DEV), off.to(DEV), alpha=0.5, pooling=pooling)
        for t in range(T):
            s, e = t * B * L, (t + 1) * B * L
            gt = np.ascontiguousarray(grad.numpy()[:, t * D:(t + 1) * D])
            exp = coracle.bwd_f32(W0[t].copy(), idx.numpy()[s:e], np.arange(B, dtype=np.int64) * L, gt, None, alpha=0.5)
            assert np.array_equal(m.table(t).cpu().numpy(), exp), (pooling, t)
        results.append(m)
    m = results[0]
    ts = m._tables()
    assert ts.fixed_pooling(idx.to(DEV), off.to(DEV), B) == L and ts.fixed_pooling(idx.to(DEV), off.to(DEV), B, claim=3) == 0
    di, do = idx.to(DEV), off.to(DEV)
    _sort_indices(ts, di, do, B, phases=2)
    m.optimizer = "rowwise_adagrad"
    with pytest.raises(param_amd.ParamAmdError, match="two-phase"):
        m.adagrad_step_(grad.to(DEV), di, do, presorted=True)
    m.adagrad_step_(grad.to(DEV), di, do)                        # sorts for itself (one phase): fine
    # a wrong fixed-pooling claim is caught by the check entry point
    ragged = off.clone()
    ragged[5] += 1
    op = ts.request(di, ragged.to(DEV), B, None, 0, None)
    op.fixed_pooling = L
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(_lib.load().pm_embbag_check(ctypes.byref(op), err.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert int(err.item()) >= 1
    op.fixed_pooling = 0
    param_amd.set_backward_tuning()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARAM_AMD_FUZZ_SEEDS", "24"))))
def test_random_fixed_pooling_requests_under_random_tuning(coracle, seed):
    """fixed-pooling requests whose sizes do / do not line up with the sort tile (4096) and the apply tile (1024), tables of
    3 ... 70000 rows (runs far beyond the exact-run limit included), with and without per-sample weights, under a random
    backward tuning each: per-table sort segments, XCD-affine tiles, two bag phases, the general path -- all against the
    sequential oracle (bit-exact up to 256 lookups per row, 1e-5 of an fp64 sum beyond)."""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(7000 + seed)
    T = int(rng.integers(1, 21))
    B = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
    L = int(rng.choice([1, 2, 4, 8, 16, 32]))
    D = int(rng.choice([16, 64, 128]))
    rows = [int(rng.choice([3, 100, 5000, 70000])) for _ in range(T)]
    weighted = rng.random() < 0.3
    knobs = (int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(1, 3)))
    mode = int(rng.integers(0, 3))
    param_amd.set_backward_tuning(*knobs)
    param_amd.set_sort_tuning(mode)
    try:
        m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=seed, fused_update=False)
        idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
        off = (np.arange(T * B + 1) * L).astype(np.int64)
        psw = rng.uniform(0.5, 1.5, idx.size).astype(np.float32) if weighted else None
        grad = rng.standard_normal((B, T * D)).astype(np.float32)
        W0 = [m.table(t).cpu().numpy().copy() for t in range(T)]
        t_ = lambda a: torch.from_numpy(a).to(DEV)   # noqa: E731
        m.scatter_add_(t_(grad), t_(idx), t_(off), alpha=0.25, per_sample_weights=None if psw is None else t_(psw))
        for t in range(T):
            s, e = t * B * L, (t + 1) * B * L
            gt = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
            pw = None if psw is None else psw[s:e]
            exp = coracle.bwd_f32(W0[t].copy(), idx[s:e], np.arange(B, dtype=np.int64) * L, gt, pw, alpha=0.25)
            got = m.table(t).cpu().numpy()
            cnt = np.bincount(idx[s:e], minlength=rows[t])
            cold = cnt <= 256
            assert np.array_equal(got[cold], exp[cold]), (seed, t, knobs, mode, T, B, L, D, rows[t])
            contrib = 0.25 * gt.astype(np.float64)[np.repeat(np.arange(B), L)] * (1.0 if pw is None else pw.astype(np.float64)[:, None])
            truth = W0[t].astype(np.float64)
            mag = np.abs(W0[t]).astype(np.float64)
            np.add.at(truth, idx[s:e], contrib)
            np.add.at(mag, idx[s:e], np.abs(contrib))
            tol = np.maximum(1e-5, (256 + cnt[:, None] / 32) * 2.0 ** -24) * mag + 1e-30
            assert (np.abs(got - truth) <= tol).all(), (seed, t, knobs, mode, T, B, L, D, rows[t])
    finally:
        param_amd.set_backward_tuning()
        param_amd.set_sort_tuning()


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_first_pass_reading_the_indices_equals_the_key_building_kernel(idx_dtype, monkeypatch):
    """per-table segments, one phase, no weights: the first radix pass forms (key, bag) from the index array itself
    (no build_keys launch).  Same sorted pairs, hence bit-identical tables, as with PARAM_AMD_SORT_FUSED_KEYS=0 -- for
    both index types, table counts that are / are not multiples of 8, fp32 and bf16 tables, SGD and Adagrad."""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355, _lib
    from param_amd.indices import tbe_request

    param_amd.set_backward_tuning(sort_impl=2)       # round 2's sort (the default sort decides this per table on the device)
    for T, B, L, dtype, opt in ((5, 2048, 8, torch.float32, "sgd"), (8, 4096, 20, torch.bfloat16, "sgd"),
                                (3, 1024, 4, torch.float32, "rowwise_adagrad")):
        rows = [50_000 + 7 * t for t in range(T)]
        idx, off = tbe_request(rows, B, L, alpha=1.05, device=DEV, seed=T, index_dtype=idx_dtype)
        grad = torch.randn(B, T * 64, device=DEV)
        results = []
        for fused in ("1", "0"):
            monkeypatch.setenv("PARAM_AMD_SORT_FUSED_KEYS", fused)
            m = BatchedEmbeddingBagMI355(rows, 64, dtype=dtype, device=DEV, init="normal", seed=9, learning_rate=0.05, optimizer=opt)
            plan = ctypes.create_string_buffer(512)
            op = m._tables().request(idx, off, B, None, 0, None)
            op.fixed_pooling = L
            assert _lib.load().pm_embbag_sort_plan(ctypes.byref(op), max(rows), 1, plan, 512) == _lib.PM_OK
            assert f"fused_keys={fused}".encode() in plan.value and b"segmented=1" in plan.value
            if opt == "sgd":
                m.scatter_add_(grad, idx, off, alpha=-0.05, batch=B)
                results.append([m.table(t).clone() for t in range(T)])
            else:
                m.adagrad_step_(grad, idx, off, batch=B)
                results.append([m.table(t).clone() for t in range(T)] + [m.momentum_table(t).clone() for t in range(T)])
        assert all(torch.equal(a, b) for a, b in zip(*results)), (T, dtype, opt)
    param_amd.set_backward_tuning()
