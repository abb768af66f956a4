"""Runs that fill whole tiles of the sorted order (round 5: tile-level partial sums, embbag_bwd_sorted_kernels.inc ``tile_joined``)
-- GPU parity (``pytest -m gpu``).

A row looked up tens of thousands of times in one step -- a 3-row Criteo table, the head of a Zipf distribution at a rank's
shape -- spans dozens of the sorted apply's tiles.  The main kernel hands the fix-up kernel ONE partial sum for every tile that
lies wholly inside such a run (its chunks' sums added in chunk order) and the fix-up walks those tiles G at a time.  Bars: rows
looked up at most EXACT_RUN = 256 times keep the sequential oracle's bits; the hot rows are within 1e-5 of an fp64 sum (sum of
|contribution|), identical from launch to launch, identical between the fused call and sort-aside + presorted apply, and
identical between the gradient layouts; row-wise Adagrad within the tolerance of tests/test_gpu_parity.py's oracle test.
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
    # (round 6: the library offers the hybrid path to requests of >= 1024 bag-major workgroups only; these tests drive its kernels with
    # small requests, so the bound is lifted here -- test_small_requests_are_not_offered_the_hybrid_path pins the product rule)
    param_amd.set_hybrid_min_tiles(0)
    yield
    param_amd.set_hybrid_min_tiles()
    param_amd.set_hybrid_tuning()


def _request(rows, B, L, seed, idt=torch.int64):
    rng = np.random.default_rng(seed)
    idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
    off = np.arange(len(rows) * B + 1, dtype=np.int64) * L
    return idx, off, torch.from_numpy(idx).to(DEV).to(idt), torch.from_numpy(off).to(DEV).to(idt)


def _truth(rows_t, D, idx_t, B, L, g, pw=None):
    contrib = g.astype(np.float64)[np.repeat(np.arange(B), L)] * (1.0 if pw is None else pw.astype(np.float64)[:, None])
    truth, mag = np.zeros((rows_t, D)), np.zeros((rows_t, D))
    np.add.at(truth, idx_t, contrib)
    np.add.at(mag, idx_t, np.abs(contrib))
    return truth, mag


@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
@pytest.mark.parametrize("dims,B,L", [([128, 128, 128], 4096, 16), ([128, 64, 32], 2048, 24), ([256, 256, 256], 1024, 40), ([16, 16, 16], 8192, 6)])
def test_runs_over_many_tiles_dense_gradient(coracle, dims, B, L, idt):
    """3-, 11- and 30000-row tables in one request: ~20 K, ~6 K and ~2 lookups per row; every lane-group width; mixed widths (the
    narrow tables leave lanes of the wide instance idle); weighted and unweighted"""
    from param_amd import BatchedEmbeddingBagMI355

    rows = [3, 11, 30000]
    T = len(rows)
    m = BatchedEmbeddingBagMI355(rows, dims, device=DEV, init="normal", seed=B, fused_update=False)
    idx, off, it, ot = _request(rows, B, L, seed=L, idt=idt)
    rng = np.random.default_rng(B + L)
    grad = rng.standard_normal((B, sum(dims))).astype(np.float32)
    col = np.concatenate([[0], np.cumsum(dims)])
    for weighted in (False, True):
        psw = rng.uniform(0.5, 1.5, idx.size).astype(np.float32) if weighted else None
        pt = None if psw is None else torch.from_numpy(psw).to(DEV)
        dense = m.dense_grad(torch.from_numpy(grad).to(DEV), it, ot, pt, batch=B)
        again = m.dense_grad(torch.from_numpy(grad).to(DEV), it, ot, pt, batch=B)
        for t in range(T):
            s, e = t * B * L, (t + 1) * B * L
            g = np.ascontiguousarray(grad[:, col[t]:col[t + 1]])
            pw = None if psw is None else psw[s:e]
            got = dense[t].cpu().numpy()
            assert np.array_equal(got, again[t].cpu().numpy()), (t, weighted)                 # launch to launch
            truth, mag = _truth(rows[t], dims[t], idx[s:e], B, L, g, pw)
            assert (np.abs(got - truth) <= 1e-5 * mag + 1e-30).all(), (dims, weighted, t)
            cold = np.bincount(idx[s:e], minlength=rows[t]) <= 256
            if cold.any():
                ref = coracle.bwd_f32(np.zeros((rows[t], dims[t]), np.float32), idx[s:e], np.arange(B) * L, g, pw)
                assert np.array_equal(got[cold], ref[cold]), (dims, weighted, t)
    assert int(np.bincount(idx[:B * L], minlength=3).min()) > 8 * 1024                         # table 0: every row spans > 8 tiles


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16, torch.float16])
def test_runs_over_many_tiles_in_place_all_entry_points_agree(wdt):
    """in-place SGD on fp32 / 16-bit tables: fused call == sort aside + presorted apply == the [T, B, D] gradient layout == the
    blocked layout, bit for bit; against fp64 within 1e-5 (+ one rounding of the table's type)"""
    from param_amd import BatchedEmbeddingBagMI355

    rows, D, Bl, W, L = [3, 5000, 7], 128, 1024, 4, 12
    B = Bl * W
    T = len(rows)
    idx, off, it, ot = _request(rows, B, L, seed=5)
    g_tbd = torch.randn(T, B, D, device=DEV)
    g_bd = g_tbd.permute(1, 0, 2).reshape(B, T * D).contiguous()
    g_blk = g_tbd.view(T, W, Bl, D).permute(1, 0, 2, 3).contiguous()
    mk = lambda lay, **k: BatchedEmbeddingBagMI355(rows, D, dtype=wdt, device=DEV, init="normal", layout=lay, seed=3, fused_update=False, **k)  # noqa: E731
    m_bd, m_pre, m_tbd, m_blk = mk("bd"), mk("bd"), mk("tbd"), mk("blocked", block_bags=Bl)
    before = [m_bd.table(t).double().cpu().numpy().copy() for t in range(T)]
    m_bd.scatter_add_(g_bd, it, ot, alpha=-0.125, batch=B)
    m_pre.sort_indices(it, ot, batch=B)
    m_pre.scatter_add_(g_bd, it, ot, alpha=-0.125, batch=B, presorted=True)
    m_tbd.scatter_add_(g_tbd, it, ot, alpha=-0.125, batch=B)
    m_blk.scatter_add_(g_blk, it, ot, alpha=-0.125, batch=B)
    for other in (m_pre, m_tbd, m_blk):
        for t in range(T):
            assert torch.equal(m_bd.table(t), other.table(t)), (wdt, t)
    eps = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[wdt]
    for t in range(T):
        s, e = t * B * L, (t + 1) * B * L
        truth, mag = _truth(rows[t], D, idx[s:e], B, L, g_tbd[t].cpu().numpy())
        exact = before[t] - 0.125 * truth
        got = m_bd.table(t).double().cpu().numpy()
        assert (np.abs(got - exact) <= np.abs(exact) * eps + 1e-5 * (0.125 * mag + np.abs(before[t])) + 1e-30).all(), (wdt, t)


@pytest.mark.parametrize("weighted", [False, True])
def test_runs_over_many_tiles_rowwise_adagrad(coracle, weighted):
    """fused row-wise Adagrad (one column pass, all lanes): hot rows of a 3-row table against the CPU restatement, tolerance of
    test_fused_rowwise_adagrad_vs_oracle; weighted requests take the per-chunk path (no join in that instance): same bars"""
    from param_amd import BatchedEmbeddingBagMI355

    rows, B, L = [3, 900], 4096, 8
    for D in (128, 56):
        m = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=D, learning_rate=0.05, optimizer="rowwise_adagrad", eps=1e-6)
        m2 = BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=D, learning_rate=0.05, optimizer="rowwise_adagrad", eps=1e-6)
        W = [m.table(t).cpu().numpy().copy() for t in range(2)]
        mom = [np.zeros(r, np.float32) for r in rows]
        rng = np.random.default_rng(D)
        for step in range(2):
            idx, off, it, ot = _request(rows, B, L, seed=10 * step + D)
            grad = (0.01 * rng.standard_normal((B, 2 * D))).astype(np.float32)
            psw = rng.uniform(0.5, 1.5, idx.size).astype(np.float32) if weighted else None
            pt = None if psw is None else torch.from_numpy(psw).to(DEV)
            for mod in (m, m2):
                mod.adagrad_step_(torch.from_numpy(grad).to(DEV), it, ot, pt)
            for t in range(2):
                s, e = t * B * L, (t + 1) * B * L
                g = np.ascontiguousarray(grad[:, t * D:(t + 1) * D])
                coracle.bwd_rowwise_adagrad(W[t], mom[t], idx[s:e], np.arange(B) * L, g, None if psw is None else psw[s:e], lr=0.05, eps=1e-6)
                assert np.allclose(m.momentum_table(t).cpu().numpy(), mom[t], rtol=2e-5, atol=1e-12), (D, step, t)
                assert np.allclose(m.table(t).cpu().numpy(), W[t], rtol=2e-5, atol=2e-6), (D, step, t)
        for t in range(2):          # (the tables, not the raw buffer: its alignment padding is never written)
            assert torch.equal(m.table(t), m2.table(t)) and torch.equal(m.momentum_table(t), m2.momentum_table(t)), (D, t)
