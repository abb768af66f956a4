"""GPU parity tests (run on a real MI355X: ``pytest -m gpu``).

Every test goes through the C ABI (param_amd._lib -> libparam_amd.so); the CPU oracle
(oracle/) and the committed torch goldens are the checkers.  Bars (BASELINE.json north_star):
  * forward: bit-exact (the kernel adds in index order, as the oracle and torch's CPU kernel do);
  * backward (float atomics, order not fixed): |err| <= 1e-5 * sum_j |contribution_j| per element.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
EXACT_RUN = 256  # kExactRun in param_amd/csrc/embbag_bwd_sorted_kernels.inc


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    import param_amd

    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    param_amd.load_library()  # raises loudly if libparam_amd.so is missing: no fallback
    yield
    param_amd.set_tuning()


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _single(meta):
    return [n for n, m in meta.items() if "tables" not in m]


def _psw(data, name):
    return data[f"{name}.psw"] if f"{name}.psw" in data.files else None


def _module(W, dtype=torch.float32):
    from param_amd import EmbeddingBagMI355

    return EmbeddingBagMI355(W.shape[0], W.shape[1], _weight=_t(W, dtype))


# ----------------------------------------------------------------------------- forward
@pytest.mark.parametrize("unroll,xcd,nt", [(0, -1, -1), (2, 0, 0), (4, 1, 1), (8, 1, 0)])
def test_forward_goldens_bit_exact(cases, unroll, xcd, nt):
    import param_amd

    param_amd.set_tuning(unroll=unroll, xcd_affine=xcd, nt_loads=nt)
    data, meta = cases
    for name in _single(meta):
        m = _module(data[f"{name}.W"])
        psw = _psw(data, name)
        with torch.no_grad():
            out = m(_t(data[f"{name}.idx"]), _t(data[f"{name}.off"]), None if psw is None else _t(psw))
        assert np.array_equal(out.cpu().numpy(), data[f"{name}.out"]), (name, unroll, xcd, nt)


def test_forward_int32_indices_and_16bit_tables(cases):
    data, _ = cases
    with torch.no_grad():
        out = _module(data["u_d32.W"])(_t(data["u_d32.idx_i32"]), _t(data["u_d32.off_i32"]))
        assert np.array_equal(out.cpu().numpy(), data["u_d32.out"])
        # W is bf16-representable, so the fp32 -> bf16 cast of the fixture is exact
        mb = _module(data["bf16_d128.W"], torch.bfloat16)
        assert np.array_equal(mb.weight.data.view(torch.int16).cpu().numpy().view(np.uint16), data["bf16_d128.W_bits"])
        out = mb(_t(data["bf16_d128.idx"]), _t(data["bf16_d128.off"]))
        assert np.array_equal(out.cpu().numpy(), data["bf16_d128.out"])
        out = _module(data["f16_d64.W_f16"].astype(np.float32), torch.float16)(
            _t(data["f16_d64.idx"]), _t(data["f16_d64.off"]))
        assert np.array_equal(out.cpu().numpy(), data["f16_d64.out"])


def _batched_from_numpy(tabs, layout="bd", dtype=torch.float32):
    from param_amd import BatchedEmbeddingBagMI355

    m = BatchedEmbeddingBagMI355([t.shape[0] for t in tabs], [t.shape[1] for t in tabs], dtype=dtype,
                                 device=DEV, layout=layout, init=None, fused_update=False)
    for t, W in enumerate(tabs):
        m.table(t).copy_(_t(W, dtype))
    return m


def test_batched_goldens_bit_exact_both_layouts(cases):
    data, meta = cases
    for name, mm in meta.items():
        if "tables" not in mm:
            continue
        tabs = [data[f"{name}.W{t}"] for t in range(mm["tables"])]
        idx, off = _t(data[f"{name}.idx"]), _t(data[f"{name}.off"])
        out = _batched_from_numpy(tabs).lookup(idx, off)
        assert np.array_equal(out.cpu().numpy(), data[f"{name}.out"]), name
        # offsets without the trailing entry is the same request
        out2 = _batched_from_numpy(tabs).lookup(idx, off[:-1].contiguous(), batch=mm["bags"])
        assert torch.equal(out, out2)
        if len({t.shape[1] for t in tabs}) == 1:
            D = tabs[0].shape[1]
            tbd = _batched_from_numpy(tabs, "tbd").lookup(idx, off).cpu().numpy()
            for t in range(len(tabs)):
                assert np.array_equal(tbd[t], data[f"{name}.out"][:, t * D:(t + 1) * D])


def test_batch_slices_compose(cases):
    """bag_begin/bag_count chunks (the a2a pipelining unit) write exactly their rows."""
    data, meta = cases
    name = "tbe_same"
    tabs = [data[f"{name}.W{t}"] for t in range(meta[name]["tables"])]
    m = _batched_from_numpy(tabs)
    idx, off = _t(data[f"{name}.idx"]), _t(data[f"{name}.off"])
    B = meta[name]["bags"]
    out = torch.full((B, sum(t.shape[1] for t in tabs)), float("nan"), device=DEV)
    for b0, n in [(0, 5), (5, 1), (6, 0), (6, 10)]:
        m.lookup(idx, off, out=out, bag_begin=b0, bag_count=n)
    assert np.array_equal(out.cpu().numpy(), data[f"{name}.out"])


def test_seeded_mid_size_vs_c_oracle(coracle):
    """Sizes the C oracle finishes in seconds; ragged lengths, a bag longer than the LDS tile
    (fallback path), empty bags, every supported dim class, both index types."""
    rng = np.random.default_rng(11)
    for D, dtype in [(128, torch.float32), (64, torch.float32), (56, torch.float32), (256, torch.float32),
                     (512, torch.float32), (128, torch.bfloat16), (64, torch.float16), (8, torch.float32)]:
        R, B = 50000, 1500
        lens = rng.integers(0, 41, B)
        lens[7] = 9000          # > idx_cap: exercises the non-staged path for that tile
        lens[100:110] = 0
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        idx = rng.integers(0, R, int(lens.sum())).astype(np.int64)
        Wt = torch.randn(R, D, generator=torch.Generator().manual_seed(D)).to(dtype)
        Wf = Wt.float().numpy()
        exp = coracle.fwd(Wf, idx, off)
        for it in (torch.int64, torch.int32):
            with torch.no_grad():
                got = _module(Wf, dtype)(_t(idx, it), _t(off, it))
            assert np.array_equal(got.cpu().numpy(), exp), (D, dtype, it)


@pytest.mark.parametrize("wdt,D", [(torch.float32, 128), (torch.float32, 16), (torch.float32, 512), (torch.float32, 1024),
                                   (torch.bfloat16, 128), (torch.float16, 256)])
def test_staged_output_forward_bit_identical(coracle, wdt, D):
    """fixed-pooling requests take the LDS-staged output burst (pm_set_forward_tuning, default on): same bits as the
    row-by-row stores and as the oracle, both layouts, with and without per-sample weights, tile tails included
    (batch not a multiple of the tile), rows too wide for a 16 KB staging buffer (D = 1024: not staged)"""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(D)
    T, R, B, L = 5, 3000, 531, 7
    for layout in ("bd", "tbd"):
        m = BatchedEmbeddingBagMI355([R] * T, D, dtype=wdt, device=DEV, init="normal", seed=2, layout=layout, fused_update=False)
        tabs = [m.table(t).float().cpu().numpy() for t in range(T)]
        idx = rng.integers(0, R, T * B * L).astype(np.int64)
        off = (np.arange(T * B + 1) * L).astype(np.int64)
        for psw in (None, rng.standard_normal(idx.size).astype(np.float32)):
            exp = coracle.fwd_batched(tabs, idx, off, B, psw=psw, layout=layout)
            outs = []
            for stage in (1, 0):
                param_amd.set_forward_tuning(stage)
                outs.append(m.lookup(_t(idx), _t(off), None if psw is None else _t(psw)).cpu().numpy())
            param_amd.set_forward_tuning()
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], exp), (layout, psw is not None)
            # a batch slice through the staged kernel writes exactly its rows
            o2 = torch.full(exp.shape, float("nan"), device=DEV)
            m.lookup(_t(idx), _t(off), None if psw is None else _t(psw), out=o2, bag_begin=100, bag_count=333)
            o2 = o2.cpu().numpy()
            sel = (slice(100, 433),) if layout == "bd" else (slice(None), slice(100, 433))
            assert np.array_equal(o2[sel], exp[sel])
            assert np.isnan(np.delete(o2, np.arange(100, 433), axis=0 if layout == "bd" else 1)).all()


def test_staged_forward_when_lookups_divide_evenly_but_bags_are_ragged(coracle):
    """the output burst is chosen from a host-visible fact (lookups divide evenly over the bags) and sizes the index tile
    for the AVERAGE bag: a ragged request that happens to divide evenly -- a quarter of the bags 80 lookups long, the rest
    empty -- overflows that tile and must take the direct-index path, with the same bits"""
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(77)
    T, R, D, B = 3, 4000, 128, 256
    lens = np.zeros(T * B, dtype=np.int64)
    lens[::4] = 80                                   # average 20; tiles of 32 bags hold 640 lookups on average ...
    lens[:64] = 0
    lens[64:96] = 160                                # ... but this one 5120
    lens[96:128] = 0
    # make the total divide evenly: pad the last bag
    short = (-int(lens.sum())) % (T * B)
    lens[-1] += short
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([rng.integers(0, R, int(lens[t * B:(t + 1) * B].sum())) for t in range(T)]).astype(np.int64)
    assert idx.size % (T * B) == 0
    for layout in ("bd", "tbd"):
        m = BatchedEmbeddingBagMI355([R] * T, D, device=DEV, init="normal", seed=5, layout=layout, fused_update=False)
        tabs = [m.table(t).cpu().numpy() for t in range(T)]
        out = m.lookup(_t(idx), _t(off), batch=B)
        exp = coracle.fwd_batched(tabs, idx, off, B, layout=layout)
        assert np.array_equal(out.cpu().numpy(), exp), layout


def test_weighted_forward_vs_oracle(coracle):
    rng = np.random.default_rng(5)
    R, D, B, L = 3000, 128, 300, 20
    W = rng.standard_normal((R, D)).astype(np.float32)
    idx = rng.integers(0, R, B * L)
    off = np.arange(B) * L
    psw = rng.standard_normal(B * L).astype(np.float32)
    with torch.no_grad():
        got = _module(W)(_t(idx), _t(off), _t(psw))
    assert np.array_equal(got.cpu().numpy(), coracle.fwd(W, idx, off, psw))  # one FMA per step on both sides


# ----------------------------------------------------------------------------- backward
def _mag(W_shape, idx, off, grad, psw, alpha=1.0):
    from oracle.embbag_oracle import bag_bounds

    mag = np.zeros(W_shape, dtype=np.float64)
    start, end = bag_bounds(off, len(off), len(idx))
    bag_of = np.repeat(np.arange(len(off)), end - start)
    contrib = np.abs(grad.astype(np.float64))[bag_of] * abs(alpha)
    if psw is not None:
        contrib = contrib * np.abs(psw.astype(np.float64))[:, None]
    np.add.at(mag, idx, contrib)
    return mag


def test_backward_dense_grad_goldens(cases, coracle):
    data, meta = cases
    for name in _single(meta):
        W, idx, off, grad = data[f"{name}.W"], data[f"{name}.idx"], data[f"{name}.off"], data[f"{name}.grad"]
        psw = _psw(data, name)
        m = _module(W)
        out = m(_t(idx), _t(off), None if psw is None else _t(psw))
        out.backward(_t(grad))
        dW = m.weight.grad.cpu().numpy().astype(np.float64)
        tol = 1e-5 * _mag(W.shape, idx, off, grad, psw) + 1e-30
        assert (np.abs(dW - data[f"{name}.dW"]) <= tol).all(), name          # vs torch dense grad
        ref = coracle.bwd_f32(np.zeros_like(W), idx, off, grad, psw)
        # default (sorted, no atomics) path: same order of adds as the sequential oracle -> bit-exact
        # for every row looked up at most EXACT_RUN times; hotter rows: ordered chunk partials, 1e-5
        got = m.weight.grad.cpu().numpy()
        cold = np.bincount(idx, minlength=W.shape[0]) <= EXACT_RUN
        assert np.array_equal(got[cold], ref[cold]), name
        assert (np.abs(got.astype(np.float64) - ref) <= tol).all(), name
        # atomic path (order not fixed): 1e-5 relative
        from param_amd.embedding_bag import _bwd
        ts = m._tables()
        dA = torch.zeros(W.shape, dtype=torch.float32, device=DEV)
        ptr = torch.tensor([dA.data_ptr()], dtype=torch.int64, device=DEV)
        _bwd(ts, _t(grad), _t(idx), _t(off), len(off), ptr, torch.float32, 1.0, None if psw is None else _t(psw),
             method="atomic")
        assert (np.abs(dA.cpu().numpy().astype(np.float64) - ref) <= tol).all(), name
    # pure scatter (distinct rows, one contribution each) is exact
    W = data["gather_l1.W"]
    idx = np.arange(16, dtype=np.int64) * 3
    off = np.arange(16, dtype=np.int64)
    grad = data["gather_l1.grad"]
    m = _module(W)
    m(_t(idx), _t(off)).backward(_t(grad))
    exp = np.zeros_like(W)
    exp[idx] = grad
    assert np.array_equal(m.weight.grad.cpu().numpy(), exp)


def test_fused_inplace_update_fp32_and_bf16(cases, coracle):
    from oracle import embbag_oracle as O

    data, meta = cases
    name = "tbe_same"
    tabs = [data[f"{name}.W{t}"] for t in range(meta[name]["tables"])]
    idx, off, B = data[f"{name}.idx"], data[f"{name}.off"], meta[name]["bags"]
    D = tabs[0].shape[1]
    rng = np.random.default_rng(2)
    grad = rng.standard_normal((B, D * len(tabs))).astype(np.float32)
    lr = 0.05
    # fp32 tables: W += -lr * grad, in place, all tables in one launch
    m = _batched_from_numpy(tabs)
    m.scatter_add_(_t(grad), _t(idx), _t(off), alpha=-lr)                      # sorted: bit-exact
    ma = _batched_from_numpy(tabs)
    ma.scatter_add_(_t(grad), _t(idx), _t(off), alpha=-lr, method="atomic")   # atomics: 1e-5
    for t, W in enumerate(tabs):
        s, e = off[t * B], off[(t + 1) * B]
        loc = off[t * B:(t + 1) * B] - s
        g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
        exp = coracle.bwd_f32(W.copy(), idx[s:e], loc, g, alpha=-lr)
        assert np.array_equal(m.table(t).cpu().numpy(), exp), t
        tol = 1e-5 * (_mag(W.shape, idx[s:e], loc, g, None, lr) + np.abs(W)) + 1e-30
        assert (np.abs(ma.table(t).cpu().numpy().astype(np.float64) - exp) <= tol).all(), t
    # bf16 tables, sorted path: widen, accumulate the row's whole update in fp32, round once
    # == oracle_embbag_bwd_bf16, bit for bit
    ms = _batched_from_numpy(tabs, dtype=torch.bfloat16)
    ms.scatter_add_(_t(grad), _t(idx), _t(off), alpha=-lr)
    for t, W in enumerate(tabs):
        s, e = off[t * B], off[(t + 1) * B]
        loc = off[t * B:(t + 1) * B] - s
        g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
        exp_bits = coracle.bwd_bf16(O.f32_to_bf16_bits(W), idx[s:e], loc, g, alpha=-lr)
        got_bits = ms.table(t).view(torch.int16).cpu().numpy().view(np.uint16)
        assert np.array_equal(got_bits, exp_bits), t
    # bf16 tables, atomic path: packed bf16 atomics round once per add -> <= 1 bf16 ulp of the
    # running value per add
    mb = _batched_from_numpy(tabs, dtype=torch.bfloat16)
    mb.scatter_add_(_t(grad), _t(idx), _t(off), alpha=-lr, method="atomic")
    for t, W in enumerate(tabs):
        s, e = off[t * B], off[(t + 1) * B]
        loc = off[t * B:(t + 1) * B] - s
        g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
        Wb = O.bf16_bits_to_f32(O.f32_to_bf16_bits(W))
        exp = coracle.bwd_f32(Wb.copy(), idx[s:e], loc, g, alpha=-lr)
        cnt = np.zeros(W.shape[0])
        np.add.at(cnt, idx[s:e], 1)
        bound = (cnt[:, None] + 1) * 2.0 ** -8 * (np.abs(exp) + _mag(W.shape, idx[s:e], loc, g, None, lr)) + 1e-30
        got = mb.table(t).float().cpu().numpy()
        assert (np.abs(got - exp) <= bound).all(), t


def test_autograd_fused_update_path(cases):
    """BatchedEmbeddingBagMI355.forward + .backward == TBE-style fused SGD step."""
    data, meta = cases
    name = "tbe_mixed"
    tabs = [data[f"{name}.W{t}"] for t in range(meta[name]["tables"])]
    from param_amd import BatchedEmbeddingBagMI355

    m = BatchedEmbeddingBagMI355([t.shape[0] for t in tabs], [t.shape[1] for t in tabs], device=DEV, init=None,
                                 learning_rate=0.1, fused_update=True)
    for t, W in enumerate(tabs):
        m.table(t).copy_(_t(W))
    out = m(_t(data[f"{name}.idx"]), _t(data[f"{name}.off"]))
    assert np.array_equal(out.detach().cpu().numpy(), data[f"{name}.out"])
    out.backward(torch.ones_like(out))          # reference create_grad(): ones_like(fwd_out)
    B = meta[name]["bags"]
    idx, off = data[f"{name}.idx"], data[f"{name}.off"]
    for t, W in enumerate(tabs):
        exp = W.astype(np.float64).copy()
        np.add.at(exp, idx[off[t * B]:off[(t + 1) * B]], -0.1)
        assert np.allclose(m.table(t).cpu().numpy(), exp, rtol=1e-5, atol=1e-6), t


def test_sorted_backward_mid_size_bit_exact_and_deterministic(coracle):
    """Heavy duplicates (Zipf head + one row hit by every lookup of a bag range), ragged bags,
    per-sample weights, several dims, int32/int64 indices: sorted path == sequential oracle."""
    from oracle.embbag_oracle import bag_bounds
    from param_amd.embedding_bag import _bwd, _sort_indices
    from param_amd.indices import zipf_indices

    rng = np.random.default_rng(21)
    for D, weighted, it in [(128, False, torch.int64), (64, True, torch.int64), (56, False, torch.int32),
                            (256, False, torch.int64), (512, True, torch.int32), (8, False, torch.int64)]:
        R, B = 20000, 700
        lens = rng.integers(0, 60, B)
        lens[3] = 5000
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        n = int(lens.sum())
        idx = zipf_indices(1.2, R, n, 1, dedupe=False, generator=torch.Generator().manual_seed(D)).numpy()
        idx[off[3]:off[3] + 5000] = 17                     # one very long run
        grad = rng.standard_normal((B, D)).astype(np.float32)
        psw = rng.standard_normal(n).astype(np.float32) if weighted else None
        W = rng.standard_normal((R, D)).astype(np.float32)
        exp = coracle.bwd_f32(W.copy(), idx, off, grad, psw, alpha=-0.03)
        m = _module(W)
        ts = m._tables()
        tabs_ptr = ts.d_ptrs
        args = (_t(grad), _t(idx, it), _t(off, it), B, tabs_ptr, torch.float32, -0.03, None if psw is None else _t(psw))
        _bwd(ts, *args)
        got = m.weight.data.cpu().numpy()
        cold = np.bincount(idx, minlength=R) <= EXACT_RUN
        assert cold.sum() > 0.9 * R and (~cold).sum() >= 3          # both regimes are exercised
        assert np.array_equal(got[cold], exp[cold]), (D, weighted)   # bit-exact: sequential order
        # hot rows (ordered chunk partials): 1e-5 relative to an fp64 accumulation -- NOT to the
        # fp32 sequential oracle, whose own error on the 5000-fold identical add is ~2e-4 relative
        # (measured: |gpu - fp64| 5e-4 vs |oracle - fp64| 2.7e-2 on that row)
        start, end = bag_bounds(off, B, n)
        bag_of = np.repeat(np.arange(B), end - start)
        contrib = -0.03 * grad.astype(np.float64)[bag_of]
        if psw is not None:
            contrib = contrib * psw.astype(np.float64)[:, None]
        truth = W.astype(np.float64).copy()
        np.add.at(truth, idx, contrib)
        tol = 1e-5 * (_mag(W.shape, idx, off, grad, psw, 0.03) + np.abs(W)) + 1e-30
        assert (np.abs(got.astype(np.float64) - truth) <= tol).all(), (D, weighted)
        # pre-sorted on the request alone, then applied: SAME BITS as the one-call form (deterministic)
        m2 = _module(W)
        ts2 = m2._tables()
        _sort_indices(ts2, args[1], args[2], B, args[7])
        _bwd(ts2, args[0], args[1], args[2], B, ts2.d_ptrs, torch.float32, -0.03, args[7], presorted=True)
        assert np.array_equal(m2.weight.data.cpu().numpy(), got), (D, weighted)


def test_sorted_backward_batch_slice_and_multi_table(cases, coracle):
    data, meta = cases
    name = "tbe_same"
    tabs = [data[f"{name}.W{t}"] for t in range(meta[name]["tables"])]
    idx, off, B = data[f"{name}.idx"], data[f"{name}.off"], meta[name]["bags"]
    D = tabs[0].shape[1]
    grad = np.random.default_rng(4).standard_normal((B, D * len(tabs))).astype(np.float32)
    m = _batched_from_numpy(tabs)
    # two disjoint batch slices compose to the full update
    m.scatter_add_(_t(grad), _t(idx), _t(off), alpha=0.5, bag_begin=0, bag_count=6)
    m.scatter_add_(_t(grad), _t(idx), _t(off), alpha=0.5, bag_begin=6, bag_count=B - 6)
    for t, W in enumerate(tabs):
        s = off[t * B]
        loc = off[t * B:(t + 1) * B + 1] - s
        g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
        exp = coracle.bwd_f32(W.copy(), idx[s:s + loc[6]], loc[:6], g[:6], alpha=0.5)
        exp = coracle.bwd_f32(exp, idx[s + loc[6]:s + loc[B]], loc[6:B] - loc[6], g[6:], alpha=0.5)
        assert np.array_equal(m.table(t).cpu().numpy(), exp), t


def test_sorted_backward_wide_keys():
    """rows close to 2^30 with several tables -> 64-bit sort keys.  Checked against a SEQUENTIAL fp32 restatement on the touched
    rows only (the tables do not fit a dense oracle: the lookups of a table are compacted to slots, numpy's unbuffered add.at
    applies them one after the other in lookup order, exactly the oracle's loop -- oracle/embbag_oracle.c: oracle_embbag_bwd_f32),
    bit for bit for rows looked up at most 256 times, and for exact run-to-run determinism."""
    from param_amd import BatchedEmbeddingBagMI355

    free, _ = torch.cuda.mem_get_info()
    if free < (60 << 30):
        pytest.skip("needs ~40 GiB")
    rows = [1 << 30, 1000, 5000, 77, 300]          # bits(2^30)=30 + bits(5)=3 -> 33-bit keys
    m = BatchedEmbeddingBagMI355(rows, 4, device=DEV, init=None, fused_update=False)
    m.weights.data.zero_()
    B, L = 64, 6
    gen = torch.Generator(device=DEV).manual_seed(5)
    idx = torch.cat([torch.randint(0, r, (B * L,), device=DEV, generator=gen) for r in rows])
    idx[:40] = (1 << 30) - 1                       # last row of the big table, many duplicates
    off = torch.arange(len(rows) * B + 1, device=DEV) * L
    grad = torch.randn(B, 4 * len(rows), device=DEV, generator=gen)
    m.scatter_add_(grad, idx, off, alpha=0.75)
    touched = [m.table(t)[idx[t * B * L:(t + 1) * B * L]].clone() for t in range(len(rows))]
    m.weights.data.zero_()
    m.scatter_add_(grad, idx, off, alpha=0.75)
    idx_h, g_h = idx.cpu().numpy(), grad.cpu().numpy()
    for t in range(len(rows)):
        assert torch.equal(m.table(t)[idx[t * B * L:(t + 1) * B * L]], touched[t])
        it = idx_h[t * B * L:(t + 1) * B * L]
        uniq, slot = np.unique(it, return_inverse=True)
        acc = np.zeros((uniq.size, 4), dtype=np.float32)
        contrib = (np.float32(0.75) * g_h[:, 4 * t:4 * t + 4])[np.repeat(np.arange(B), L)]      # alpha * g, one product per lookup
        np.add.at(acc, slot, contrib)                                                          # sequential, lookup order, fp32
        got = touched[t].cpu().numpy()
        cnt = np.bincount(slot)
        cold = cnt[slot] <= EXACT_RUN
        assert cold.all() or t in (0, 3)
        assert np.array_equal(got[cold], acc[slot][cold]), t
        np.testing.assert_allclose(got[~cold], acc[slot][~cold], rtol=1e-5, atol=1e-6)
    assert float(m.table(0)[(1 << 30) - 1].abs().sum()) > 0


def test_criteo_shaped_tables_multi_hot(coracle):
    """BASELINE configs[4] geometry scaled down: 26 tables of very different sizes (3 rows .. 50 K), per-table
    multi-hot pooling 1 .. 100 in one batched request: forward bit-exact, in-place update exact on cold rows."""
    from oracle.embbag_oracle import bag_bounds
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.compute.pt import dataset as ds
    from param_amd.indices import tbe_request

    rows = [min(r, 50000) for r in ds.criteo_v2_rows]
    pools, D, B = ds.criteo_v2_multi_hot, 128, 256
    m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=5, fused_update=False)
    idx, off = tbe_request(rows, B, pools, alpha=1.05, device=DEV, seed=9)
    assert off.numel() == 26 * B + 1 and int(off[-1]) == B * sum(pools) == idx.numel()
    m.check(idx, off)
    tabs = [m.table(t).cpu().numpy().copy() for t in range(26)]
    out = m.lookup(idx, off)
    ih, oh = idx.cpu().numpy(), off.cpu().numpy()
    assert np.array_equal(out.cpu().numpy(), coracle.fwd_batched(tabs, ih, oh, B))
    grad = torch.randn(B, 26 * D, device=DEV)
    m.scatter_add_(grad, idx, off, alpha=-0.01)
    gh = grad.cpu().numpy()
    for t in range(26):
        s, e = oh[t * B], oh[(t + 1) * B]
        loc = oh[t * B:(t + 1) * B] - s
        g = np.ascontiguousarray(gh[:, t * D:(t + 1) * D])
        exp = coracle.bwd_f32(tabs[t].copy(), ih[s:e], loc, g, alpha=-0.01)
        got = m.table(t).cpu().numpy()
        cold = np.bincount(ih[s:e], minlength=rows[t]) <= EXACT_RUN
        assert np.array_equal(got[cold], exp[cold]), t
        start, end = bag_bounds(loc, B, e - s)
        truth = tabs[t].astype(np.float64)
        np.add.at(truth, ih[s:e], -0.01 * g.astype(np.float64)[np.repeat(np.arange(B), end - start)])
        tol = 1e-5 * (_mag(tabs[t].shape, ih[s:e], loc, g, None, 0.01) + np.abs(tabs[t])) + 1e-30
        assert (np.abs(got - truth) <= tol).all(), t


# ----------------------------------------------------------------------------- utilities
def test_check_request_raises_like_torch(cases):
    from param_amd import check_request

    data, _ = cases
    m = _module(data["u_d32.W"])
    ts = m._tables()
    idx, off = _t(data["u_d32.idx"]), _t(data["u_d32.off"])
    check_request(ts, idx, off, off.numel())
    bad = idx.clone()
    bad[5] = data["u_d32.W"].shape[0]
    with pytest.raises(IndexError):
        check_request(ts, bad, off, off.numel())
    bad[5] = -1
    with pytest.raises(IndexError):
        check_request(ts, bad, off, off.numel())
    boff = off.clone()
    boff[3] = boff[4] + 1
    with pytest.raises(IndexError):
        check_request(ts, idx, boff, off.numel())


def test_fill_random_statistics_and_determinism():
    from param_amd import fill_random_

    a = fill_random_(torch.empty(1 << 22, device=DEV), "normal", 0.0, 1.0, seed=7)
    b = fill_random_(torch.empty(1 << 22, device=DEV), "normal", 0.0, 1.0, seed=7)
    c = fill_random_(torch.empty(1 << 22, device=DEV), "normal", 0.0, 1.0, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1.0) < 5e-3
    assert abs((a.abs() < 1.0).float().mean().item() - 0.6827) < 5e-3
    u = fill_random_(torch.empty(1000003, device=DEV, dtype=torch.bfloat16), "uniform", -0.5, 0.5, seed=1)
    assert u.min().item() >= -0.5 and u.max().item() <= 0.5 and abs(u.float().mean().item()) < 5e-3
    # prefix property: element i depends only on (seed, i)
    s = fill_random_(torch.empty(1000, device=DEV), "normal", 0.0, 1.0, seed=7)
    assert torch.equal(s, a[:1000])


# ----------------------------------------------------------------------------- full size
_FULL = {}


def _full_size_model():
    """the benchmark's tables (10 M x 128 fp32, as many as fit: 48 on a 288 GB MI355X), built once per test process"""
    from param_amd import BatchedEmbeddingBagMI355

    if "m" not in _FULL:
        free, _total = torch.cuda.mem_get_info()
        R, D = 10_000_000, 128
        T = int(min(48, (free - (16 << 30)) // (R * D * 4)) // 8 * 8)
        assert T >= 8, f"only {free / 2**30:.0f} GiB free"
        _FULL["m"] = BatchedEmbeddingBagMI355([R] * T, D, device=DEV, init="normal", seed=1, fused_update=False)
    return _FULL["m"]


def test_full_size_properties_and_live_torch_oracle():
    """BASELINE.json configs[1] geometry at the largest table count that fits (fp32), checked
    through size-independent properties and against torch-ROCm's own embedding_bag as a live
    second oracle on a subset of tables."""
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.indices import tbe_request

    R, D, B, L = 10_000_000, 128, 8192, 20
    m = _full_size_model()
    T = len(m.rows)
    idx, off = tbe_request([R] * T, B, L, alpha=0.0, device=DEV, seed=3)
    m.check(idx, off)
    out = m.lookup(idx, off)
    # (1) live oracle: torch's own GPU kernel on 3 tables (same inputs), 1e-5 relative
    for t in (0, T // 2, T - 1):
        sl = slice(t * B * L, (t + 1) * B * L)
        ref = torch.nn.functional.embedding_bag(idx[sl], m.table(t), off[t * B:(t + 1) * B] - t * B * L, mode="sum")
        mag = torch.nn.functional.embedding_bag(idx[sl], m.table(t).abs(), off[t * B:(t + 1) * B] - t * B * L, mode="sum")
        assert ((out[:, t * D:(t + 1) * D] - ref).abs() <= 1e-5 * mag + 1e-30).all(), t
    # (2) gather is bit-exact: L=1 lookups return the table rows themselves
    g_idx = torch.randint(0, R, (T * B,), device=DEV)
    g_off = torch.arange(T * B + 1, device=DEV)
    g_out = m.lookup(g_idx, g_off)
    for t in (0, T - 1):
        assert torch.equal(g_out[:, t * D:(t + 1) * D], m.table(t)[g_idx[t * B:(t + 1) * B]])
    # (3) linearity / checksum of checksums: sum of all pooled outputs == sum over lookups of row sums (fp64)
    t = 1
    rows = m.table(t)[idx[t * B * L:(t + 1) * B * L]].double().sum()
    assert abs(out[:, t * D:(t + 1) * D].double().sum().item() - rows.item()) <= 1e-6 * B * L * D
    # (4) determinism + tuning variants agree bit-for-bit at full size
    import param_amd

    for unroll, xcd, nt in [(4, 0, 0), (8, 1, 1), (2, 1, 0)]:
        param_amd.set_tuning(unroll=unroll, xcd_affine=xcd, nt_loads=nt)
        assert torch.equal(m.lookup(idx, off), out)
    param_amd.set_tuning()
    param_amd.set_forward_tuning(0)                 # row-by-row output stores instead of the staged burst
    assert torch.equal(m.lookup(idx, off), out)
    param_amd.set_forward_tuning()
    # (5) fused update round trip on the big slab: +g then -g restores touched rows to ~1e-6, untouched exactly
    t0 = m.table(0)
    probe = t0[:1000].clone()
    grad = torch.randn(B, T * D, device=DEV)
    m.scatter_add_(grad, idx, off, alpha=0.5)
    m.scatter_add_(grad, idx, off, alpha=-0.5)
    assert (t0[:1000] - probe).abs().max().item() < 1e-4
    hit = torch.zeros(R, dtype=torch.bool, device=DEV)
    hit[idx[:B * L]] = True
    untouched = (~hit[:1000]).nonzero().squeeze(1)
    assert torch.equal(t0[untouched], probe[untouched])


def test_full_size_zipf_headline_workload():
    """The HEADLINE workload of bench.py at full size -- 48 x 10 M x 128 fp32, B = 8192, L = 20, Zipf(1.05) indices (hot rows =
    low ids, per-bag dedupe) -- which the alpha = 0 test above does not exercise: the forward against torch-ROCm's own
    embedding_bag on 3 tables including table 0 (whose head rows are hit thousands of times per step, from L2), and the
    sorted backward of the WHOLE request against an fp64 accumulation on a slice of table 0 that holds the hottest rows
    (chunk-partial path), mid-frequency rows (exact re-walk path) and rows looked up once."""
    from param_amd.indices import tbe_request

    R, D, B, L = 10_000_000, 128, 8192, 20
    m = _full_size_model()
    T = len(m.rows)
    idx, off = tbe_request([R] * T, B, L, alpha=1.05, device=DEV, seed=1)
    m.check(idx, off)
    counts0 = torch.bincount(idx[:B * L], minlength=R)
    assert int(counts0.max()) > 2000 and int((counts0 == 1).sum()) > 20000      # skewed: a hot head and a long tail
    out = m.lookup(idx, off)
    for t in (0, T // 2, T - 1):
        sl = slice(t * B * L, (t + 1) * B * L)
        ref = torch.nn.functional.embedding_bag(idx[sl], m.table(t), off[t * B:(t + 1) * B] - t * B * L, mode="sum")
        mag = torch.nn.functional.embedding_bag(idx[sl], m.table(t).abs(), off[t * B:(t + 1) * B] - t * B * L, mode="sum")
        assert ((out[:, t * D:(t + 1) * D] - ref).abs() <= 1e-5 * mag + 1e-30).all(), t
    # checksum of checksums on table 0: sum of all pooled outputs == sum over lookups of row sums (fp64)
    rows_sum = m.table(0)[idx[:B * L]].double().sum()
    assert abs(out[:, :D].double().sum().item() - rows_sum.item()) <= 1e-6 * B * L * D
    # tuning variants agree bit for bit on the skewed request too (XCD-affine mapping on / off)
    import param_amd

    for unroll, xcd in [(4, 0), (2, 1)]:
        param_amd.set_tuning(unroll=unroll, xcd_affine=xcd)
        assert torch.equal(m.lookup(idx, off), out)
    param_amd.set_tuning()

    # sorted backward of the whole 7.86 M-lookup request; checked on table 0's rows in a slice
    i0 = idx[:B * L]
    hot = torch.argsort(counts0, descending=True)[:48]                             # > 256 lookups each: chunk partials
    mid = (((counts0 >= 2) & (counts0 <= 256)).nonzero().squeeze(1))[:2000]         # exact in-tile / re-walk paths
    once = ((counts0 == 1).nonzero().squeeze(1))[:2000]
    never = ((counts0 == 0).nonzero().squeeze(1))[:2000]
    sel = torch.unique(torch.cat([hot, mid, once, never]))                         # sorted
    before = m.table(0)[sel].clone()
    grad = torch.randn(B, T * D, device=DEV)
    alpha = -0.25
    m.scatter_add_(grad, idx, off, alpha=alpha)
    after = m.table(0)[sel]
    pos = torch.isin(i0, sel).nonzero().squeeze(1)                                  # lookups of table 0 that hit a selected row
    slot = torch.searchsorted(sel, i0[pos])
    bag = pos // L
    g0 = grad[:, :D].double()
    G = torch.zeros(sel.numel(), D, dtype=torch.float64, device=DEV).index_add_(0, slot, g0[bag])
    Gabs = torch.zeros(sel.numel(), D, dtype=torch.float64, device=DEV).index_add_(0, slot, g0[bag].abs())
    exp = before.double() + alpha * G
    err = (after.double() - exp).abs()
    assert (err <= 1e-5 * (abs(alpha) * Gabs + before.double().abs()) + 1e-30).all(), float(err.max())
    sel_counts = counts0[sel]
    untouched = sel_counts == 0
    assert torch.equal(after[untouched], before[untouched])                         # rows never looked up keep their bits
    assert int((sel_counts > 256).sum()) >= 40 and int((sel_counts == 1).sum()) >= 1000
    # a second application with the opposite sign brings every selected row back (to rounding)
    m.scatter_add_(grad, idx, off, alpha=-alpha)
    back = m.table(0)[sel]
    assert ((back - before).abs().double() <= 4e-5 * (abs(alpha) * Gabs + before.double().abs()) + 1e-30).all()
    # last user of the 245 GB slab in this process: hand the memory back (later tests allocate tens of GB)
    del m, back, after, before
    _FULL.clear()
    torch.cuda.empty_cache()


def test_full_size_bf16_tables_forward_and_inplace_update():
    """BASELINE configs[2]'s other dtype at full size: all 64 x 10 M x 128 bf16 tables (163.8 GB) resident, B = 8192, L = 20,
    Zipf(1.05).  Forward against torch-ROCm's fp32 embedding_bag on the widened copies of 2 tables (rows are widened exactly,
    accumulation is fp32 on both sides: 1e-5 of sum |row|); L = 1 gather bit-exact; the sorted in-place update (fp32
    accumulate, ONE rounding per touched row) on a slice of table 0 against that definition computed in fp64 -> bf16; rows
    never looked up keep their bits."""
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.indices import tbe_request

    free, _total = torch.cuda.mem_get_info()
    R, D, B, L = 10_000_000, 128, 8192, 20
    T = int(min(64, (free - (24 << 30)) // (R * D * 2)) // 8 * 8)
    assert T >= 8, f"only {free / 2**30:.0f} GiB free"
    m = BatchedEmbeddingBagMI355([R] * T, D, dtype=torch.bfloat16, device=DEV, init="normal", seed=4, fused_update=False)
    idx, off = tbe_request([R] * T, B, L, alpha=1.05, device=DEV, seed=6)
    out = m.lookup(idx, off)
    for t in (0, T - 1):
        Wf = m.table(t).float()
        sl = slice(t * B * L, (t + 1) * B * L)
        ref = torch.nn.functional.embedding_bag(idx[sl], Wf, off[t * B:(t + 1) * B] - t * B * L, mode="sum")
        mag = torch.nn.functional.embedding_bag(idx[sl], Wf.abs(), off[t * B:(t + 1) * B] - t * B * L, mode="sum")
        assert ((out[:, t * D:(t + 1) * D] - ref).abs() <= 1e-5 * mag + 1e-30).all(), t
        del Wf, ref, mag
    g_idx = torch.randint(0, R, (T * B,), device=DEV)
    g_out = m.lookup(g_idx, torch.arange(T * B + 1, device=DEV))
    assert torch.equal(g_out[:, :D], m.table(0)[g_idx[:B]].float())
    # in-place update of the whole request; checked on table 0: hottest rows, mid-frequency rows, singles, untouched rows
    i0 = idx[:B * L]
    counts0 = torch.bincount(i0, minlength=R)
    sel = torch.unique(torch.cat([torch.argsort(counts0, descending=True)[:32],
                                  ((counts0 >= 2) & (counts0 <= 256)).nonzero().squeeze(1)[:1500],
                                  (counts0 == 1).nonzero().squeeze(1)[:1500], (counts0 == 0).nonzero().squeeze(1)[:1500]]))
    before = m.table(0)[sel].clone()
    grad = torch.randn(B, T * D, device=DEV)
    alpha = -0.5
    m.scatter_add_(grad, idx, off, alpha=alpha)
    after = m.table(0)[sel]
    pos = torch.isin(i0, sel).nonzero().squeeze(1)
    slot = torch.searchsorted(sel, i0[pos])
    g0 = grad[:, :D].double()
    G = torch.zeros(sel.numel(), D, dtype=torch.float64, device=DEV).index_add_(0, slot, g0[pos // L])
    Gabs = torch.zeros(sel.numel(), D, dtype=torch.float64, device=DEV).index_add_(0, slot, g0[pos // L].abs())
    exact = before.double() + alpha * G
    # one bf16 rounding of the fp32-accumulated row: within half a bf16 ulp of the exact value, plus the fp32 accumulation error
    tol = exact.abs() * 2.0 ** -8 + 1e-5 * (abs(alpha) * Gabs + before.double().abs()) + 1e-30
    assert ((after.double() - exact).abs() <= tol).all()
    untouched = counts0[sel] == 0
    assert torch.equal(after[untouched], before[untouched]) and int((counts0[sel] > 256).sum()) >= 16
    del m
    torch.cuda.empty_cache()


def test_fused_rowwise_adagrad_vs_oracle(coracle):
    """pm_embbag_bwd_sorted_adagrad vs the CPU restatement of fbgemm's exact row-wise Adagrad (parity UNPINNED:
    fbgemm is absent; the oracle itself is checked against an fp64 numpy form here).  Two steps, Zipf duplicates
    incl. a hot row (chunk-partial path), per-sample weights, dims whose lane group has idle lanes (D=56)."""
    from param_amd import BatchedEmbeddingBagMI355
    from param_amd.indices import zipf_indices

    rng = np.random.default_rng(8)
    for D, weighted in [(128, False), (56, True), (64, False), (256, True)]:
        rows, B, L = [4000, 700], 300, 10
        m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=D, learning_rate=0.05,
                                     optimizer="rowwise_adagrad", eps=1e-6)
        W = [m.table(t).cpu().numpy().copy() for t in range(2)]
        mom = [np.zeros(r, np.float32) for r in rows]
        for step in range(2):
            idx = torch.cat([zipf_indices(1.3, r, B * L, 1, dedupe=False, generator=torch.Generator().manual_seed(10 * step + t))
                             for t, r in enumerate(rows)])
            idx[:450] = 3                                        # hot row: > EXACT_RUN lookups in table 0
            off = torch.arange(2 * B + 1) * L
            grad = torch.from_numpy(rng.standard_normal((B, 2 * D)).astype(np.float32))
            psw = torch.from_numpy(rng.uniform(0.5, 1.5, idx.numel()).astype(np.float32)) if weighted else None
            prev = [(m.table(t).cpu().numpy().copy(), m.momentum_table(t).cpu().numpy().copy()) for t in range(2)]
            m.adagrad_step_(grad.to(DEV), idx.to(DEV), off.to(DEV), None if psw is None else psw.to(DEV))
            for t in range(2):
                s, e = t * B * L, (t + 1) * B * L
                g = np.ascontiguousarray(grad.numpy()[:, t * D:(t + 1) * D])
                W0, m0 = W[t].copy(), mom[t].copy()
                coracle.bwd_rowwise_adagrad(W[t], mom[t], idx.numpy()[s:e], np.arange(B) * L, g,
                                            None if psw is None else psw.numpy()[s:e], lr=0.05, eps=1e-6)
                # the oracle against an fp64 form of the published algorithm
                G = np.zeros(W0.shape)
                w8 = np.ones(B * L) if psw is None else psw.numpy()[s:e].astype(np.float64)
                np.add.at(G, idx.numpy()[s:e], g.astype(np.float64)[np.repeat(np.arange(B), L)] * w8[:, None])
                touched = np.bincount(idx.numpy()[s:e], minlength=rows[t]) > 0
                m64 = m0 + (G ** 2).mean(1)
                W64 = W0 - 0.05 / (np.sqrt(m64)[:, None] + 1e-6) * G
                assert np.allclose(mom[t][touched], m64[touched], rtol=2e-5, atol=1e-12)
                assert np.allclose(W[t][touched], W64[touched], rtol=1e-4, atol=2e-5)
                # the GPU against the oracle
                gm = m.momentum_table(t).cpu().numpy()
                gw = m.table(t).cpu().numpy()
                # rows not looked up in THIS step keep their bits (weights and state)
                assert np.array_equal(gw[~touched], prev[t][0][~touched]) and np.array_equal(gm[~touched], prev[t][1][~touched])
                assert np.allclose(gm, mom[t], rtol=2e-5, atol=1e-12), (D, step, t)
                assert np.allclose(gw, W[t], rtol=2e-5, atol=2e-6), (D, step, t)
        # deterministic: a second module fed the same two steps ends bit-identical
    ma = BatchedEmbeddingBagMI355([500], 128, device=DEV, init="normal", seed=1, optimizer="rowwise_adagrad")
    mb = BatchedEmbeddingBagMI355([500], 128, device=DEV, init="normal", seed=1, optimizer="rowwise_adagrad")
    idx = torch.randint(0, 500, (64 * 20,), device=DEV)
    off = torch.arange(65, device=DEV) * 20
    g = torch.randn(64, 128, device=DEV)
    for mod in (ma, mb):
        out = mod(idx, off)                 # autograd path: backward == fused Adagrad step
        out.backward(g)
    assert torch.equal(ma.table(0), mb.table(0)) and torch.equal(ma.momentum, mb.momentum) and float(ma.momentum.sum()) > 0


@pytest.mark.parametrize("mode,code", [("l2", 1), ("decouple", 2)])
def test_rowwise_adagrad_weight_decay_vs_oracle(coracle, mode, code):
    """weight decay modes of the fused row-wise Adagrad (the options the reference's TBE operator forwards,
    split_table_batched_embeddings_ops.py:258-300) vs the oracle's restatement of fbgemm's rule -- parity UNPINNED
    (fbgemm absent); the oracle is itself checked against an fp64 form.  Two steps, duplicates, a hot row."""
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(31)
    rows, D, B, L, lr, eps, wd = [3000, 900], 64, 200, 8, 0.05, 1e-6, 0.02
    m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=5, learning_rate=lr, optimizer="rowwise_adagrad",
                                 eps=eps, weight_decay=wd, weight_decay_mode=mode)
    W = [m.table(t).cpu().numpy().copy() for t in range(2)]
    mom = [np.zeros(r, np.float32) for r in rows]
    for step in range(2):
        idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
        idx[:300] = 7                                         # > 256 lookups of one row: chunk-partial path
        off = np.arange(2 * B + 1, dtype=np.int64) * L
        grad = rng.standard_normal((B, 2 * D)).astype(np.float32)
        m.adagrad_step_(torch.from_numpy(grad).to(DEV), torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV))
        for t in range(2):
            s, e = t * B * L, (t + 1) * B * L
            g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
            W0, m0 = W[t].astype(np.float64), mom[t].astype(np.float64)
            coracle.bwd_rowwise_adagrad(W[t], mom[t], idx[s:e], np.arange(B) * L, g, None, lr=lr, eps=eps,
                                        weight_decay=wd, weight_decay_mode=code)
            G = np.zeros(W0.shape)
            np.add.at(G, idx[s:e], g.astype(np.float64)[np.repeat(np.arange(B), L)])
            touched = np.bincount(idx[s:e], minlength=rows[t]) > 0
            gx = G + wd * W0 if code == 1 else G
            m64 = m0 + (gx ** 2).mean(1)
            mult = lr / (np.sqrt(m64) + eps)
            corr = np.ones(rows[t]) - (mult * wd if code == 1 else lr * wd)
            W64 = corr[:, None] * W0 - mult[:, None] * G
            assert np.allclose(mom[t][touched], m64[touched], rtol=2e-5) and np.allclose(W[t][touched], W64[touched], rtol=1e-4, atol=2e-5)
            assert np.array_equal(W[t][~touched], W0[~touched].astype(np.float32))          # untouched rows do not decay
            assert np.allclose(m.momentum_table(t).cpu().numpy(), mom[t], rtol=2e-5, atol=1e-12), (step, t)
            assert np.allclose(m.table(t).cpu().numpy(), W[t], rtol=2e-5, atol=2e-6), (step, t)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rowwise_adagrad_stochastic_rounding(dtype):
    """16-bit tables, stochastic_rounding=True (the reference operator's fixed choice, :291): every stored element is
    one of the two table-type neighbours of the exact fp32 update, the MEAN over many rows that receive the same update
    equals it (round-to-nearest would be off by a fixed bias), and the run is reproducible for a given step."""
    from param_amd import BatchedEmbeddingBagMI355

    R, D, lr = 8192, 64, 0.01

    def run(sr):
        m = BatchedEmbeddingBagMI355([R], D, dtype=dtype, device=DEV, init=None, learning_rate=lr, optimizer="rowwise_adagrad",
                                     eps=1e-8, stochastic_rounding=sr)
        m.table(0).fill_(1.0)
        idx = torch.arange(R, device=DEV)                      # every row looked up once, same gradient
        off = torch.arange(R + 1, device=DEV)
        g = torch.full((R, D), 0.37, device=DEV)
        m.adagrad_step_(g, idx, off)
        return m.table(0).clone()

    # exact fp32 update of every element: m = 0.37^2, mult = lr / 0.37 -> w = 1 - lr (up to fp32 rounding)
    g32 = np.float32(0.37)
    mom = np.float32(g32 * g32 * D) / np.float32(D)
    exact = float(np.float32(1.0) - np.float32(np.float32(lr) / (np.sqrt(mom, dtype=np.float32) + np.float32(1e-8))) * g32)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11     # spacing just below 1.0
    lo_f = np.floor(exact / ulp) * ulp
    hi_f = lo_f + ulp
    w = run(True).float().cpu().numpy()
    assert set(np.unique(w)) <= {np.float32(lo_f), np.float32(hi_f)} and len(np.unique(w)) == 2
    p_up = (exact - lo_f) / ulp
    n = w.size
    assert abs(w.mean() - exact) < 5 * ulp * np.sqrt(p_up * (1 - p_up) / n) + 1e-7
    assert abs((w == np.float32(hi_f)).mean() - p_up) < 0.01
    rne = run(False).float().cpu().numpy()
    assert len(np.unique(rne)) == 1 and abs(rne.mean() - exact) > 0.2 * min(p_up, 1 - p_up) * ulp   # nearest: a fixed bias
    assert np.array_equal(run(True).float().cpu().numpy(), w)                                      # reproducible


@pytest.mark.parametrize("D", [16, 32, 64, 128, 256, 512])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_sorted_backward_runs_around_chunk_and_tile_boundaries(coracle, D, idt):
    """Runs of 100..300 lookups per row (every lane-group width: D = 16 .. 512): almost every run crosses a chunk
    boundary, many cross a 1024-position tile boundary, some exceed the 256-lookup exact limit.  The in-tile
    owner-applies-once path, the staged fix-up walk and the chunk-partial path must each hit exactly their rows: rows with
    <= 256 lookups bit-identical to the sequential oracle, the others within 1e-5 of an fp64 sum; weighted and unweighted,
    in-place SGD and dense gradient."""
    from oracle import embbag_oracle as O
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(D + (1 if idt == torch.int32 else 0))
    rows, B, L = [70, 37], 700, 25                       # 17500 lookups per table -> ~250 and ~473 per row
    m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=D, fused_update=False)
    W = [m.table(t).cpu().numpy().copy() for t in range(2)]
    idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
    idx[:40] = 5                                          # a short run at the very start of the data
    off = np.arange(2 * B + 1, dtype=np.int64) * L
    grad = rng.standard_normal((B, 2 * D)).astype(np.float32)
    for weighted in (False, True):
        psw = rng.uniform(0.5, 1.5, idx.size).astype(np.float32) if weighted else None
        it, ot = torch.from_numpy(idx).to(DEV).to(idt), torch.from_numpy(off).to(DEV).to(idt)
        pt = None if psw is None else torch.from_numpy(psw).to(DEV)
        dense = m.dense_grad(torch.from_numpy(grad).to(DEV), it, ot, pt, batch=B)
        for t in range(2):
            s, e = t * B * L, (t + 1) * B * L
            g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
            pw = None if psw is None else psw[s:e]
            ref = coracle.bwd_f32(np.zeros((rows[t], D), np.float32), idx[s:e], np.arange(B) * L, g, pw)
            cnt = np.bincount(idx[s:e], minlength=rows[t])
            cold = cnt <= 256
            assert t == 1 or (cold.any() and (~cold).any())      # table 0 straddles the exact-run limit
            got = dense[t].cpu().numpy()
            assert np.array_equal(got[cold], ref[cold]), (D, weighted, t)
            contrib = g.astype(np.float64)[np.repeat(np.arange(B), L)] * (1.0 if pw is None else pw.astype(np.float64)[:, None])
            truth, mag = np.zeros((rows[t], D)), np.zeros((rows[t], D))
            np.add.at(truth, idx[s:e], contrib)
            np.add.at(mag, idx[s:e], np.abs(contrib))
            assert (np.abs(got - truth) <= 1e-5 * mag + 1e-30).all(), (D, weighted, t)
    # in place on the tables (SGD form), unweighted
    m.scatter_add_(torch.from_numpy(grad).to(DEV), it, ot, alpha=-0.25, batch=B, per_sample_weights=pt)
    for t in range(2):
        s, e = t * B * L, (t + 1) * B * L
        g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
        ref = coracle.bwd_f32(W[t].copy(), idx[s:e], np.arange(B) * L, g, psw[s:e], alpha=-0.25)
        cold = np.bincount(idx[s:e], minlength=rows[t]) <= 256
        assert np.array_equal(m.table(t).cpu().numpy()[cold], ref[cold]), (D, t)


@pytest.mark.parametrize("wdt,D", [(torch.float32, 128), (torch.float32, 16), (torch.float32, 512), (torch.bfloat16, 128),
                                   (torch.float16, 64)])
def test_split_bag_forward_long_bags(coracle, wdt, D):
    """pm_embbag_fwd_split (one workgroup per bag, wave-shuffle + LDS partial reductions) on few, long, ragged bags incl.
    empty ones and bags longer than the LDS index chunk: within 1e-5 of sum|row| of the oracle's sequential sum (the order
    differs by design), deterministic, weighted and unweighted, both index types; and faster than the sequential-per-bag
    kernel on this shape."""
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(D)
    rows, B = [50000, 777], 6
    lens = np.array([5000, 0, 2049, 1, 300, 9000, 64, 2048, 0, 4097, 33, 7], np.int64)      # 2 tables x 6 bags
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([rng.integers(0, rows[t], int(lens[t * B:(t + 1) * B].sum())) for t in range(2)]).astype(np.int64)
    m = BatchedEmbeddingBagMI355(rows, D, dtype=wdt, device=DEV, init="normal", seed=3, fused_update=False)
    tabs = [m.table(t).float().cpu().numpy() for t in range(2)]
    for weighted in (False, True):
        psw = rng.standard_normal(idx.size).astype(np.float32) if weighted else None
        for it in (torch.int64, torch.int32):
            i_t, o_t = torch.from_numpy(idx).to(DEV).to(it), torch.from_numpy(off).to(DEV).to(it)
            p_t = None if psw is None else torch.from_numpy(psw).to(DEV)
            got = m.lookup(i_t, o_t, p_t, batch=B, split_bags=True)
            exp = coracle.fwd_batched(tabs, idx, off, B, psw=psw, layout="bd")
            mag = coracle.fwd_batched([np.abs(x) for x in tabs], idx, off, B, psw=None if psw is None else np.abs(psw), layout="bd")
            assert (np.abs(got.cpu().numpy() - exp) <= 1e-5 * mag + 1e-30).all(), (weighted, it)
            assert torch.equal(got, m.lookup(i_t, o_t, p_t, batch=B, split_bags=True))              # deterministic
            seq = m.lookup(i_t, o_t, p_t, batch=B)
            assert np.array_equal(seq.cpu().numpy(), exp)                                           # the default stays bit-exact
    # a batch slice writes exactly its rows
    o2 = torch.full_like(got, float("nan"))
    m.lookup(i_t, o_t, p_t, out=o2, batch=B, bag_begin=2, bag_count=3, split_bags=True)
    assert torch.equal(o2[2:5], got[2:5]) and torch.isnan(o2[:2]).all() and torch.isnan(o2[5:]).all()
    if D == 128 and wdt == torch.float32:
        big = BatchedEmbeddingBagMI355([2_000_000] * 4, 128, device=DEV, init="normal", seed=5, fused_update=False)
        bi = torch.randint(0, 2_000_000, (4 * 8 * 20000,), device=DEV)
        bo = torch.arange(4 * 8 + 1, device=DEV) * 20000                         # 32 bags of 20 000 lookups

        def ms(split):
            for _ in range(2):
                big.lookup(bi, bo, batch=8, split_bags=split)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                big.lookup(bi, bo, batch=8, split_bags=split)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 5
        t_seq, t_split = ms(False), ms(True)
        print(f"32 bags x 20000 lookups: sequential-per-bag {t_seq:.3f} ms, split {t_split:.3f} ms")
        assert t_split < t_seq


def test_forward_more_than_2_31_lookups():
    """64-bit positions end to end: 2.2e9 lookups in one request (17.6 GB of int64 indices), bag starts beyond 2^31.
    Table W[r, :] = r and indices j % 1024 make every pooled value an exact small integer with a closed form."""
    from param_amd import BatchedEmbeddingBagMI355

    R, D, B, L = 1024, 4, 1 << 20, 2100
    n = B * L
    assert n > 2 ** 31
    m = BatchedEmbeddingBagMI355([R], D, device=DEV, init=None, fused_update=False)
    m.table(0).copy_(torch.arange(R, device=DEV, dtype=torch.float32).unsqueeze(1).expand(R, D))
    idx = torch.arange(n, device=DEV, dtype=torch.int64).remainder_(R)
    off = torch.arange(B + 1, device=DEV, dtype=torch.int64) * L
    out = m.lookup(idx, off, batch=B)
    # sum over j in [b*L, (b+1)*L) of (j mod R): prefix sums of the periodic sequence
    def prefix(x):                                   # sum_{j < x} (j mod R), int64 tensor
        q, r = x // R, x % R
        return q * (R * (R - 1) // 2) + r * (r - 1) // 2
    exp = (prefix(off[1:]) - prefix(off[:-1])).to(torch.float32)
    assert float(exp.max()) < 2 ** 24                 # exact in fp32 whatever the order
    assert torch.equal(out, exp.unsqueeze(1).expand(B, D))
    del idx, out
    torch.cuda.empty_cache()


def test_sorted_backward_more_than_2_31_lookups():
    """the deterministic backward on the same 2.2e9-lookup request: 64-bit positions through key building, the radix sort
    (4-byte keys / values: positions and bags stay below 2^32), 69 M chunks, runs of 2.1 M lookups (chunk partials summed
    by the fix-up).  Gradient 1 everywhere makes every row's update an exact integer: its lookup count."""
    from param_amd import BatchedEmbeddingBagMI355

    R, D, B, L = 1024, 4, 1 << 20, 2100
    n = B * L
    m = BatchedEmbeddingBagMI355([R], D, device=DEV, init=None, fused_update=False)
    m.table(0).zero_()
    idx = torch.arange(n, device=DEV, dtype=torch.int64).remainder_(R)
    off = torch.arange(B + 1, device=DEV, dtype=torch.int64) * L
    grad = torch.ones((B, D), device=DEV)
    m.scatter_add_(grad, idx, off, alpha=1.0, batch=B)
    counts = torch.bincount(idx, minlength=R).to(torch.float32)
    assert float(counts.max()) < 2 ** 24
    assert torch.equal(m.table(0), counts.unsqueeze(1).expand(R, D))
    del idx, grad
    m._ts = None
    torch.cuda.empty_cache()


def test_fused_rowwise_adagrad_equals_torch_adagrad_where_rowwise_is_elementwise():
    """f2, the part that CAN be pinned to a real implementation without fbgemm: with all columns of a row equal, exact row-wise
    Adagrad is torch.optim.Adagrad per row (tests/test_oracle.py has the same check for the oracle).  Three fused steps on the
    GPU, two tables, rows hit 0 .. many times per step, Zipf-like duplicates."""
    from param_amd import BatchedEmbeddingBagMI355
    from tests.test_oracle import _torch_adagrad_steps

    rng = np.random.default_rng(21)
    rows, D, B, L, lr, eps = [300, 50], 16, 64, 5, 0.05, 1e-6
    m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=2, learning_rate=lr, optimizer="rowwise_adagrad", eps=eps)
    cols = [rng.standard_normal(r).astype(np.float32) for r in rows]
    for t, c in enumerate(cols):
        m.table(t).copy_(torch.from_numpy(np.repeat(c[:, None], D, axis=1)))
    sums = [[], []]
    for step in range(3):
        idx = torch.cat([torch.from_numpy(np.minimum(rng.zipf(1.4, B * L) - 1, r - 6).astype(np.int64)) for r in rows])   # last 5 rows unhit
        off = torch.arange(2 * B + 1) * L
        g_bag = (rng.integers(-8, 9, (B, 2)) / 4.0).astype(np.float32)          # dyadic: sums exact in any order
        grad = torch.from_numpy(np.repeat(g_bag, D, axis=1))                     # [B, 2*D]: table t's columns all g_bag[:, t]
        m.adagrad_step_(grad.to(DEV), idx.to(DEV), off.to(DEV))
        for t, r in enumerate(rows):
            G = np.zeros(r, np.float32)
            np.add.at(G, idx.numpy()[t * B * L:(t + 1) * B * L], g_bag[np.repeat(np.arange(B), L), t])
            sums[t].append(G)
    for t, r in enumerate(rows):
        w_t, s_t = _torch_adagrad_steps(cols[t], sums[t], lr, eps)
        gw, gm = m.table(t).cpu().numpy(), m.momentum_table(t).cpu().numpy()
        assert np.allclose(gm, s_t, rtol=1e-6, atol=0), t
        assert np.allclose(gw, w_t[:, None], rtol=2e-6, atol=1e-7), t
        assert np.array_equal(gw[-5:], np.repeat(cols[t][-5:, None], D, axis=1))
