"""The segmented key sort of the sorted backward (param_amd/csrc/seg_sort.hip; ``pytest -m gpu``).

Every request is sorted through the C ABI (``pm_embbag_sort_indices_ex``) and the sorted pairs are read back from the
workspace (``pm_embbag_sorted_pairs``) and compared, element for element, with numpy's stable order of the same keys:
  mode 0 / 2   (table, row, position)
  mode 1       (table, row & 255, row >> 8, position)
values = the lookup's bag inside its table (its position in the request for weighted requests).  Covered: per-table
segments from the offsets (fixed pooling, per-table pooling, ragged incl. empty bags and empty tables), batch slices (compact
output, device-side count), weights, int32 indices, 8-byte keys, tables with fewer than 8 key bits, and buckets of every
size class of the bucket-local sort (registers / one workgroup per CU / external single-workgroup sort).
Then the whole backward on the same kinds of request against the C oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    import param_amd

    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    param_amd.load_library()
    # these tests read the sorted pairs of a BARE sort; while the hybrid backward's kernels are launched the sort finishes inside
    # the apply call (include/param_amd.h, pm_set_hybrid_tuning), so they run with it off (tests/test_gpu_hybrid.py covers it)
    param_amd.set_hybrid_tuning(0)
    yield
    param_amd.set_hybrid_tuning()
    param_amd.set_backward_tuning()
    param_amd.set_sort_tuning()


def _request(rng, rows, B, lens_of_table, hot=None, idx_dtype=np.int64):
    """indices / offsets of a TBE request; hot: per table either None or (fraction, candidate rows)"""
    lens = np.concatenate([np.asarray(lens_of_table(t), dtype=np.int64) for t in range(len(rows))])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    parts = []
    for t, r in enumerate(rows):
        n = int(lens[t * B:(t + 1) * B].sum())
        idx = rng.integers(0, r, n)
        if hot is not None and hot[t] is not None:
            frac, cand = hot[t]
            pick = rng.random(n) < frac
            idx[pick] = rng.choice(cand, int(pick.sum()))
        parts.append(idx)
    return np.concatenate(parts).astype(idx_dtype), off.astype(idx_dtype)


def _expected(idx, off, T, B, mode, tshift, b0=0, nb=None, weighted=False):
    nb = B if nb is None else nb
    keys, vals = [], []
    for t in range(T):
        s, e = int(off[t * B + b0]), int(off[t * B + b0 + nb])
        row = idx[s:e].astype(np.int64)
        pos = np.arange(s, e)
        bag = np.searchsorted(off[t * B:(t + 1) * B + 1], pos, side="right") - 1
        order = np.lexsort((pos, row >> 8, row & 255)) if mode == 1 else np.lexsort((pos, row))
        keys.append((np.int64(t) << np.int64(tshift)) | row[order])
        vals.append((pos if weighted else bag)[order])
    return np.concatenate(keys), np.concatenate(vals)


def _sorted(m, idx, off, B, psw=None, b0=0, nb=None):
    from param_amd.embedding_bag import _sort_indices, sorted_pairs

    ts = m._tables()
    _sort_indices(ts, idx, off, B, psw, b0, nb)
    k, v, tshift = sorted_pairs(ts, idx, off, B, psw, b0, nb)
    k = k.cpu().numpy()
    k = (k.astype(np.int64) & 0xffffffff) if k.dtype == np.int32 else k.astype(np.int64)
    return k, v.cpu().numpy().astype(np.int64) & 0xffffffff, tshift


CASES = {
    # rows, B, lens(t), hot
    "fixed": dict(rows=[70000, 300, 3, 1_000_000, 5000], B=2048, lens=lambda t: np.full(2048, 8)),
    "per_table_pooling": dict(rows=[70000, 300, 3, 1_000_000, 5000], B=1024, lens=lambda t: np.full(1024, [3, 1, 20, 7, 2][t])),
    "ragged": dict(rows=[70000, 300, 3, 1_000_000, 5000], B=512, lens=None),
    "empty_table": dict(rows=[1000, 50, 99999], B=256, lens=lambda t: np.full(256, [5, 0, 9][t])),
    # one (table, digit) bucket of ~20 K pairs: the external single-workgroup sort; others of 2-10 K: one workgroup per CU
    "skew_low_digit": dict(rows=[100_000, 100_000], B=4096, lens=lambda t: np.full(4096, 16),
                           hot=[(0.3, 7 + 256 * np.arange(200)), (0.06, 9 + 256 * np.arange(300))]),
    "skew_top_digit": dict(rows=[100_000, 1_000_000], B=4096, lens=lambda t: np.full(4096, 16),
                           hot=[(0.4, np.arange(256)), (0.08, np.arange(3000))]),
    "hot_rows": dict(rows=[1 << 20, 4000], B=4096, lens=lambda t: np.full(4096, 12), hot=[(0.5, np.array([5, 77, 1 << 19])), None]),
}


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_sorted_pairs_equal_numpys_stable_order(case, mode, idt):
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    if idt == torch.int32 and case not in ("fixed", "ragged", "skew_low_digit"):
        pytest.skip("int32 indices: three representative cases")
    c = CASES[case]
    rng = np.random.default_rng(sum(map(ord, case)) + mode)
    rows, B = c["rows"], c["B"]
    T = len(rows)
    if c["lens"] is None:
        def lens(t):
            ln = rng.integers(0, 13, B)
            ln[0] = 0
            ln[-1] = 0
            ln[B // 2] = 700
            return ln
    else:
        lens = c["lens"]
    idx, off = _request(rng, rows, B, lens, c.get("hot"))
    param_amd.set_sort_tuning(mode)
    m = BatchedEmbeddingBagMI355(rows, 8, device=DEV, init=None, fused_update=False)
    it, ot = torch.from_numpy(idx).to(DEV).to(idt), torch.from_numpy(off).to(DEV).to(idt)
    k, v, tshift = _sorted(m, it, ot, B)
    ek, ev = _expected(idx, off, T, B, mode, tshift)
    assert k.shape == ek.shape
    assert np.array_equal(k, ek), (case, mode, int(np.argmax(k != ek)))
    assert np.array_equal(v, ev), (case, mode, int(np.argmax(v != ev)))
    # a batch slice sorts only its own lookups (compact, device-side count)
    b0, nb = B // 4, B // 2 + 3
    k, v, tshift = _sorted(m, it, ot, B, None, b0, nb)
    ek, ev = _expected(idx, off, T, B, mode, tshift, b0, nb)
    assert k.shape == ek.shape and np.array_equal(k, ek) and np.array_equal(v, ev), (case, mode, "slice")
    # weighted: values are request positions
    psw = torch.rand(idx.size, device=DEV)
    k, v, tshift = _sorted(m, it, ot, B, psw)
    ek, ev = _expected(idx, off, T, B, mode, tshift, weighted=True)
    assert np.array_equal(k, ek) and np.array_equal(v, ev), (case, mode, "weighted")
    param_amd.set_sort_tuning()

@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("big", [300, 512, 513, 200_000, 1 << 18, 40_000_000, 100_000_000, (1 << 27) + 1])
def test_nine_bit_digits_where_they_save_a_pass(big, mode):
    """mode 0 sorts 9 bits per pass when ceil(bits / 9) < ceil(bits / 8) for the request's widest table (9, 17, 18, 25 .. 27,
    33 .. 36 row bits): the plan says so, smaller tables ride along (fewer digits, copy passes), and the order is numpy's.
    (1 << 27) + 1 rows = 28 bits and 513 rows = 10 bits stay on 8-bit digits.)"""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.embedding_bag import sort_plan

    rng = np.random.default_rng(big % 1000003)
    bits = max(1, int(big - 1).bit_length())
    others = [3, 200, 100, 60] if big < 1000 else [3, 70000, 600, 1 << 24 if big > (1 << 24) else 5000]
    rows, B = [big] + others, 1024
    pools = [9, 2, 5, 1, 12]
    if big == 100_000_000:           # 27 row bits + 6 table bits: 8-byte keys on 9-bit digits
        rows, pools = rows + [50] * 36, pools + [1] * 36
    dim = 4
    free, _ = torch.cuda.mem_get_info()
    if free < sum(rows) * dim * 4 + (8 << 30):
        pytest.skip("not enough free HBM for the table")
    hot = [(0.2, np.array([big - 1, 0, big // 2, min(big - 1, 511), min(big - 1, 512)]))] + [None] * (len(rows) - 1)
    idx, off = _request(rng, rows, B, lambda t: np.full(B, pools[t]), hot=hot)
    param_amd.set_sort_tuning(mode)
    m = BatchedEmbeddingBagMI355(rows, dim, device=DEV, init=None, fused_update=False)
    it, ot = torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV)
    plan = dict(kv.split("=") for kv in sort_plan(m._tables(), it, ot, B).split())
    assert plan["lookback"] == ("1" if mode == 0 else "0"), plan
    nine = -(-bits // 9) < -(-bits // 8) and bits > 8
    assert plan["rbits"] == str(bits) and plan["radix_bits"] == ("9" if nine else "8"), plan
    assert plan["key_bytes"] == ("8" if len(rows) > 32 else "4"), plan
    assert plan["passes"] == str(-(-bits // (9 if nine else 8))), plan
    k, v, tshift = _sorted(m, it, ot, B)
    ek, ev = _expected(idx, off, len(rows), B, 0, tshift)
    assert np.array_equal(k, ek), (big, int(np.argmax(k != ek)))
    assert np.array_equal(v, ev), (big, int(np.argmax(v != ev)))
    b0, nb = 100, 517
    k, v, tshift = _sorted(m, it, ot, B, None, b0, nb)
    ek, ev = _expected(idx, off, len(rows), B, 0, tshift, b0, nb)
    assert np.array_equal(k, ek) and np.array_equal(v, ev), (big, "slice")
    param_amd.set_sort_tuning()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_eight_byte_keys_and_many_tables(mode):
    """2^30-row table next to small ones (33 key bits -> 8-byte keys; 30 row bits: 4 global passes / 3 local rounds) and a
    request of 300 tiny tables"""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    free, _ = torch.cuda.mem_get_info()
    if free < (40 << 30):
        pytest.skip("needs ~20 GiB")
    param_amd.set_sort_tuning(mode)
    rng = np.random.default_rng(5 + mode)
    rows, B = [1 << 30, 1000, 5000, 77, 300], 512
    idx, off = _request(rng, rows, B, lambda t: np.full(B, 6), hot=[(0.3, np.array([(1 << 30) - 1, 12345, 1 << 29])), None, None, None, None])
    m = BatchedEmbeddingBagMI355(rows, 4, device=DEV, init=None, fused_update=False)
    k, v, tshift = _sorted(m, torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV), B)
    ek, ev = _expected(idx, off, len(rows), B, mode, tshift)
    assert tshift == 30 and np.array_equal(k, ek) and np.array_equal(v, ev)
    del m
    rows = [int(r) for r in rng.integers(1, 400, 300)]
    B = 64
    idx, off = _request(rng, rows, B, lambda t: rng.integers(0, 5, B))
    m = BatchedEmbeddingBagMI355(rows, 4, device=DEV, init=None, fused_update=False)
    k, v, tshift = _sorted(m, torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV), B)
    ek, ev = _expected(idx, off, len(rows), B, mode, tshift)
    assert np.array_equal(k, ek) and np.array_equal(v, ev)
    param_amd.set_sort_tuning()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["per_table_pooling", "ragged", "skew_low_digit", "skew_top_digit"])
def test_backward_through_the_segmented_sort_vs_oracle(coracle, case, mode):
    """scatter-add of whole requests (and of a batch slice) against the sequential C oracle: bit-exact on rows looked up
    <= 256 times, 1e-5 of an fp64 sum beyond; sort plan says segmented / XCD for every kind of request"""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.embedding_bag import sort_plan

    c = CASES[case]
    rng = np.random.default_rng(77 + mode)
    rows, B, D = c["rows"], c["B"], 32
    T = len(rows)
    if c["lens"] is None:
        lens = lambda t: rng.integers(0, 13, B)   # noqa: E731
    else:
        lens = c["lens"]
    idx, off = _request(rng, rows, B, lens, c.get("hot"))
    param_amd.set_sort_tuning(mode)
    m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=3, fused_update=False)
    it, ot = torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV)
    plan = sort_plan(m._tables(), it, ot, B)
    assert "sort=seg" in plan and "segmented=1" in plan and "xcd=1" in plan and f"mode={mode}" in plan, plan
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    for b0, nb in ((0, B), (B // 3, B // 2)):
        W0 = [m.table(t).cpu().numpy().copy() for t in range(T)]
        m.scatter_add_(torch.from_numpy(grad).to(DEV), it, ot, alpha=0.25, batch=B, bag_begin=b0, bag_count=nb)
        for t in range(T):
            s, e = int(off[t * B + b0]), int(off[t * B + b0 + nb])
            loc = off[t * B + b0:t * B + b0 + nb] - s
            gt = np.ascontiguousarray(grad[b0:b0 + nb, t * D:(t + 1) * D])
            exp = coracle.bwd_f32(W0[t].copy(), idx[s:e], loc, gt, None, alpha=0.25)
            got = m.table(t).cpu().numpy()
            cnt = np.bincount(idx[s:e], minlength=rows[t])
            cold = cnt <= 256
            assert np.array_equal(got[cold], exp[cold]), (case, mode, t, b0)
            lens_t = np.diff(np.concatenate([loc, [e - s]]))
            contrib = 0.25 * gt.astype(np.float64)[np.repeat(np.arange(nb), lens_t)]
            truth, mag = W0[t].astype(np.float64), np.abs(W0[t]).astype(np.float64)
            np.add.at(truth, idx[s:e], contrib)
            np.add.at(mag, idx[s:e], np.abs(contrib))
            tol = np.maximum(1e-5, (256 + cnt[:, None] / 32) * 2.0 ** -24) * mag + 1e-30
            assert (np.abs(got - truth) <= tol).all(), (case, mode, t, b0)
    param_amd.set_sort_tuning()


def test_offsets_refilled_in_place_are_looked_at_again(coracle):
    """the stale-verdict hazard of round 2: a fixed-pooling request, then ``offsets.copy_(ragged, same total)`` into the SAME
    tensor.  The default sort reads the offsets on the device (no verdict to go stale); the round-2 sort (sort_impl 2) takes
    the verdict from the Python-side cache, whose key now carries the tensor's version counter.  Both must give the oracle's
    gradient for the ragged request."""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(3)
    rows, B, L, D = [5000, 300, 70000], 256, 8, 16
    T = len(rows)
    n = T * B * L
    idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
    fixed_off = (np.arange(T * B + 1) * L).astype(np.int64)
    # ragged with the same total AND the same per-table totals (so even a per-table divisibility test cannot tell)
    lens = np.full(T * B, L)
    for t in range(T):
        a = rng.permutation(B)[: B // 2 * 2].reshape(-1, 2) + t * B
        d = rng.integers(1, L, a.shape[0])
        lens[a[:, 0]] += d
        lens[a[:, 1]] -= d
    ragged_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    assert ragged_off[-1] == n and (lens >= 0).all() and not np.array_equal(ragged_off, fixed_off)
    grad = rng.standard_normal((B, T * D)).astype(np.float32)
    import contextlib

    from param_amd import _lib

    for impl in (0, 2):
        # (round 2's sort lives in the alternates build: libparam_amd_alt.so)
        with (_lib.use_alternates() if impl else contextlib.nullcontext()):
            if impl:
                param_amd.set_backward_tuning(sort_impl=impl)
            m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init=None, fused_update=False)
            it, g = torch.from_numpy(idx).to(DEV), torch.from_numpy(grad).to(DEV)
            off_buf = torch.from_numpy(fixed_off).to(DEV)              # the persistent buffer of a training loop
            for step, off_np in enumerate((fixed_off, ragged_off, fixed_off, ragged_off)):
                off_buf.copy_(torch.from_numpy(off_np))
                dws = m.dense_grad(g, it, off_buf, batch=B)
                for t in range(T):
                    s, e = off_np[t * B], off_np[(t + 1) * B]
                    exp = coracle.bwd_f32(np.zeros((rows[t], D), np.float32), idx[s:e], off_np[t * B:(t + 1) * B] - s,
                                          np.ascontiguousarray(grad[:, t * D:(t + 1) * D]))
                    assert np.array_equal(dws[t].cpu().numpy(), exp), (impl, step, t)
            if impl:
                param_amd.set_backward_tuning()


@pytest.mark.parametrize("hot_frac", [0.0, 0.6])
def test_one_table_a_thousand_tiles_deep_lookback(hot_frac):
    """ONE segment of ~1000 radix tiles (4 M lookups of one table): every tile's walk over its predecessors' status rows is as
    deep as the look-back form gets (all tiles start together); with `hot_frac` most pairs share a handful of rows -- one digit
    bucket holds most of every tile.  Sorted pairs element for element against numpy's stable order, modes 0 and 3."""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(41 + int(hot_frac * 10))
    rows, B, L = [3_000_000, 77], 1 << 17, 31           # table 0: 4 063 232 lookups = 992 tiles; table 1: the same count, 7 row bits
    hot = [(hot_frac, np.array([0, 1, 255, 256, 65535, 65536, 2_999_999])), None] if hot_frac > 0 else None
    idx, off = _request(rng, rows, B, lambda t: np.full(B, L), hot=hot)
    m = BatchedEmbeddingBagMI355(rows, 4, device=DEV, init=None, fused_update=False)
    it, ot = torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV)
    for mode in (0, 3):
        param_amd.set_sort_tuning(mode)
        k, v, tshift = _sorted(m, it, ot, B)
        ek, ev = _expected(idx, off, len(rows), B, mode, tshift)
        assert np.array_equal(k, ek), (mode, int(np.argmax(k != ek)))
        assert np.array_equal(v, ev), (mode, int(np.argmax(v != ev)))
    param_amd.set_sort_tuning()


def test_one_sort_applied_three_times_with_runs_to_mend(coracle):
    """the apply's work list of runs for the fix-up kernel is emptied on the device between launches (two counters in turn, no
    host state): ONE sort, THREE applies of a request full of long runs (rows looked up thousands of times, runs across every
    tile border) must equal three sequential applies of the oracle on rows with <= 256 lookups, 1e-5 of an fp64 sum beyond"""
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(8)
    rows, B, L, D = [3, 40, 20000, 1_000_000], 2048, 24, 32
    T = len(rows)
    hot = [None, None, (0.5, np.array([7, 8, 19999])), (0.3, np.array([0, 999_999]))]
    idx, off = _request(rng, rows, B, lambda t: np.full(B, L), hot=hot)
    m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=5, fused_update=False)
    it, ot = torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV)
    grads = [rng.standard_normal((B, T * D)).astype(np.float32) for _ in range(3)]
    W0 = [m.table(t).cpu().numpy().copy() for t in range(T)]
    m.sort_indices(it, ot, batch=B)
    for g in grads:
        m.scatter_add_(torch.from_numpy(g).to(DEV), it, ot, alpha=-0.125, batch=B, presorted=True)
    for t in range(T):
        s, e = int(off[t * B]), int(off[(t + 1) * B])
        loc = off[t * B:(t + 1) * B] - s
        exp = W0[t].copy()
        truth, mag = W0[t].astype(np.float64), np.abs(W0[t]).astype(np.float64)
        for g in grads:
            gt = np.ascontiguousarray(g[:, t * D:(t + 1) * D])
            exp = coracle.bwd_f32(exp, idx[s:e], loc, gt, None, alpha=-0.125)
            contrib = -0.125 * gt.astype(np.float64)[np.repeat(np.arange(B), L)]
            np.add.at(truth, idx[s:e], contrib)
            np.add.at(mag, idx[s:e], np.abs(contrib))
        got = m.table(t).cpu().numpy()
        cnt = np.bincount(idx[s:e], minlength=rows[t])
        cold = cnt <= 256
        assert np.array_equal(got[cold], exp[cold]), t
        tol = np.maximum(1e-5, 3 * (256 + cnt[:, None] / 32) * 2.0 ** -24) * mag + 1e-30
        assert (np.abs(got - truth) <= tol).all(), t
