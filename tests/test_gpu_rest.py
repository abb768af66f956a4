"""The hybrid backward's left-overs finished in LDS (round 6, ``hyb_rest_kernel``) -- GPU parity tests (``pytest -m gpu``), C ABI.

What the bag-major kernel leaves of a hybrid table -- the lookups its dup map flags -- is sorted inside LDS and applied run by run by
ONE launch when a table has at most 8192 of them; longer lists are compacted for the key sort as before.  The bar is the sorted
backward's: bit-exact against the sequential oracle (here for ANY run length: the LDS walk is sequential), bit-identical to the
round-5 route (``pm_set_hybrid_rest(0)``) for rows looked up at most 256 times, and to the hybrid-off route likewise.
Reference semantics: fbgemm TBE backward / aten::_embedding_bag_dense_backward at
train/comms/pt/pytorch_dist_backend.py:854-857, split_table_batched_embeddings_ops.py:318-324.
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
    param_amd.set_hybrid_rest()


def _model(rows, D, dtype=torch.float32, layout="bd", seed=0):
    import param_amd

    return param_amd.BatchedEmbeddingBagMI355(rows, D, dtype=dtype, device=DEV, init="normal", layout=layout, seed=seed, fused_update=False)


def _request(rows, B, pools, alpha, seed, index_dtype=torch.int64):
    from param_amd.indices import tbe_request

    return tbe_request(rows, B, pools, alpha=alpha, device=DEV, seed=seed, index_dtype=index_dtype)


def _oracle_table(coracle, W, idx_h, off_h, t, B, g_t, alpha):
    s = off_h[t * B]
    e = off_h[(t + 1) * B] if (t + 1) * B < len(off_h) else len(idx_h)
    return coracle.bwd_f32(W.copy(), idx_h[s:e], off_h[t * B:(t + 1) * B] - s, np.ascontiguousarray(g_t), alpha=alpha)


@pytest.mark.parametrize("D,layout,idt", [(128, "bd", torch.int64), (64, "tbd", torch.int32), (256, "bd", torch.int64), (32, "bd", torch.int32)])
def test_leftovers_finished_in_lds_equal_oracle_and_both_other_routes(coracle, D, layout, idt):
    """Four uniform tables of one request: two whose flagged lookups fit the LDS sort (81 920 lookups into 5 M rows: ~2 K flagged)
    and two that overflow it (the same lookups into 330 K rows: 22 % true repeats = ~18 K).  Every row equals the sequential oracle
    bit for bit; the same request with the LDS kernel off (round 5's route) and with the hybrid path off gives the same bits; the
    status words account for every lookup."""
    import param_amd

    rows, B, L = [5_000_000, 330_000, 5_000_000, 330_000], 4096, 20
    idx, off = _request(rows, B, L, 0.0, 31, idt)
    gshape = (B, len(rows) * D) if layout == "bd" else (len(rows), B, D)
    grad = torch.randn(gshape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    got, status = {}, {}
    for tag, hyb, rest in (("lds", 1, 1), ("r5", 1, 0), ("sorted", 0, 1)):
        param_amd.set_hybrid_tuning(hyb)
        param_amd.set_hybrid_rest(rest)
        m = _model(rows, D, layout=layout, seed=3)
        if tag == "lds":
            tabs = [m.table(t).cpu().numpy() for t in range(len(rows))]
        m.scatter_add_(grad, idx, off, alpha=-0.25, batch=B)
        status[tag] = m.sort_status(idx, off, batch=B)
        got[tag] = [m.table(t).cpu().numpy() for t in range(len(rows))]
        del m
    st = status["lds"]
    assert st["hybrid_tables"] == 4 and st["lds_tables"] == 2, st
    assert 1000 < st["lds_pairs"] <= 2 * 8192 and 2 * 8192 < st["pairs_sorted"] < 0.6 * 2 * B * L, st
    r5 = status["r5"]
    assert r5["hybrid_tables"] == 4 and r5["lds_tables"] == 0 and r5["lds_pairs"] == 0, r5
    assert r5["pairs_sorted"] == st["pairs_sorted"] + st["lds_pairs"], (r5, st)        # the same lookups were flagged either way
    assert status["sorted"]["pairs_sorted"] == idx.numel() and status["sorted"]["lds_tables"] == 0
    idx_h, off_h, g_h = idx.cpu().numpy().astype(np.int64), off.cpu().numpy().astype(np.int64), grad.cpu().numpy()
    for t in range(len(rows)):
        g_t = g_h[:, t * D:(t + 1) * D] if layout == "bd" else g_h[t]
        exp = _oracle_table(coracle, tabs[t], idx_h, off_h, t, B, g_t, -0.25)
        assert np.array_equal(got["lds"][t], exp), t
        assert np.array_equal(got["r5"][t], exp) and np.array_equal(got["sorted"][t], exp), t



@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_leftovers_in_lds_16_bit_tables_and_batch_slices(coracle, dtype):
    """16-bit tables (widened, fp32 accumulation, one rounding per row) in two batch slices: LDS kernel on == off == hybrid off,
    bit for bit; bf16 also against the oracle's bf16 routine for table 0 in one whole-batch call."""
    import param_amd

    rows, D, B, L = [1_500_000] * 5, 128, 2048, 16
    idx, off = _request(rows, B, L, 0.0, 17)
    grad = torch.randn((B, len(rows) * D), device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    res = {}
    for tag, hyb, rest in (("lds", 1, 1), ("r5", 1, 0), ("sorted", 0, 1)):
        param_amd.set_hybrid_tuning(hyb)
        param_amd.set_hybrid_rest(rest)
        m = _model(rows, D, dtype=dtype, seed=2)
        m.scatter_add_(grad, idx, off, alpha=-0.5, batch=B, bag_begin=0, bag_count=1200)
        st = m.sort_status(idx, off, batch=B, bag_begin=0, bag_count=1200)
        if tag == "lds":
            assert st["hybrid_tables"] == 5 and st["lds_tables"] == 5 and st["pairs_sorted"] == 0 and st["lds_pairs"] > 0, st
        m.scatter_add_(grad, idx, off, alpha=-0.5, batch=B, bag_begin=1200, bag_count=B - 1200)
        res[tag] = m.weights.data.view(torch.int16).cpu().numpy().copy()
        del m
    assert np.array_equal(res["lds"], res["r5"]) and np.array_equal(res["lds"], res["sorted"])
    if dtype == torch.bfloat16:
        param_amd.set_hybrid_tuning(1)
        param_amd.set_hybrid_rest(1)
        m = _model(rows, D, dtype=dtype, seed=2)
        W0 = m.table(0).view(torch.int16).cpu().numpy().view(np.uint16).copy()
        m.scatter_add_(grad, idx, off, alpha=-0.5, batch=B)
        idx_h, off_h = idx.cpu().numpy(), off.cpu().numpy()
        exp = coracle.bwd_bf16(W0, idx_h[:off_h[B]], off_h[:B], np.ascontiguousarray(grad.cpu().numpy()[:, :D]), alpha=-0.5)
        assert np.array_equal(m.table(0).view(torch.int16).cpu().numpy().view(np.uint16), exp)


@pytest.mark.parametrize("dtype,wd_mode", [(torch.float32, "l2"), (torch.bfloat16, "decouple"), (torch.float32, "none")])
def test_leftovers_in_lds_rowwise_adagrad_same_bits_as_the_sorted_path(coracle, dtype, wd_mode):
    """Fused row-wise Adagrad: tables AND optimizer state after two steps, LDS kernel on == off == hybrid off, bit for bit (the LDS
    walk hands adagrad_finish the same gradient sum, row and state as the sorted apply's in-chunk run); fp32 / L2 also vs the oracle."""
    import param_amd

    rows, D, B, L = [900_000] * 3, 128, 1024, 12
    idx, off = _request(rows, B, L, 0.0, 8)
    grad = torch.randn((B, len(rows) * D), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    res = {}
    for tag, hyb, rest in (("lds", 1, 1), ("r5", 1, 0), ("sorted", 0, 1)):
        param_amd.set_hybrid_tuning(hyb)
        param_amd.set_hybrid_rest(rest)
        m = _model(rows, D, dtype=dtype, seed=6)
        m.optimizer, m.learning_rate, m.eps, m.weight_decay, m.weight_decay_mode = "rowwise_adagrad", 0.05, 1e-8, 0.01, wd_mode
        W0 = m.table(0).float().cpu().numpy().copy()
        for _ in range(2):
            m.adagrad_step_(grad, idx, off, batch=B)
        if tag == "lds":
            st = m.sort_status(idx, off, batch=B)
            assert st["hybrid_tables"] == 3 and st["lds_tables"] == 3 and st["pairs_sorted"] == 0, st
        res[tag] = (m.weights.data.float().cpu().numpy().copy(), m.momentum.cpu().numpy().copy())
        del m
    for other in ("r5", "sorted"):
        assert np.array_equal(res["lds"][0], res[other][0]) and np.array_equal(res["lds"][1], res[other][1]), other
    if dtype == torch.float32 and wd_mode == "l2":
        idx_h, off_h = idx.cpu().numpy(), off.cpu().numpy()
        g0 = np.ascontiguousarray(grad.cpu().numpy()[:, :D])
        W, mom = W0.copy(), np.zeros(rows[0], dtype=np.float32)
        for _ in range(2):
            W, mom = coracle.bwd_rowwise_adagrad(W, mom, idx_h[:off_h[B]], off_h[:B], g0, lr=0.05, eps=1e-8, weight_decay=0.01,
                                                 weight_decay_mode=1)
        got = res["lds"][0].reshape(-1)[:rows[0] * D].reshape(rows[0], D)
        np.testing.assert_allclose(got, W, rtol=2e-5, atol=1e-6)


def test_long_runs_walked_in_lds_equal_the_sequential_oracle(coracle):
    """Forced eligibility (enable = 2) on Zipf tables of 8704 lookups: nearly every lookup is flagged (the list still fits the LDS
    sort) and the head rows are runs of hundreds to thousands of lookups, walked sequentially by one lane group -- the sequential
    oracle's bits for EVERY row, whatever its run length (the sorted path sums runs over 256 lookups as chunk partials: those rows
    agree with it to fp32 association only)."""
    import param_amd

    rows, D, B, L = [300_000, 300_000, 300_000], 64, 512, 17
    idx, off = _request(rows, B, L, 1.05, 41)
    grad = torch.randn((B, len(rows) * D), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    param_amd.set_hybrid_tuning(2)
    param_amd.set_hybrid_rest(1)
    m = _model(rows, D, seed=5)
    tabs = [m.table(t).cpu().numpy() for t in range(len(rows))]
    m.scatter_add_(grad, idx, off, alpha=0.5, batch=B)
    st = m.sort_status(idx, off, batch=B)
    assert st["hybrid_tables"] == 3 and st["lds_tables"] == 3 and st["pairs_sorted"] == 0 and st["lds_pairs"] > 3 * 3000, st
    idx_h, off_h, g_h = idx.cpu().numpy().astype(np.int64), off.cpu().numpy().astype(np.int64), grad.cpu().numpy()
    longest = 0
    for t in range(len(rows)):
        exp = _oracle_table(coracle, tabs[t], idx_h, off_h, t, B, g_h[:, t * D:(t + 1) * D], 0.5)
        assert np.array_equal(m.table(t).cpu().numpy(), exp), t
        longest = max(longest, int(np.bincount(idx_h[off_h[t * B]:off_h[(t + 1) * B] if t + 1 < len(rows) else len(idx_h)]).max()))
    assert longest > 256, longest                       # the case is what it says: runs beyond the sorted path's exact length
    # the sorted path agrees on every row looked up at most 256 times, and to fp32 association on the others
    param_amd.set_hybrid_tuning(0)
    ref = _model(rows, D, seed=5)
    ref.scatter_add_(grad, idx, off, alpha=0.5, batch=B)
    for t in range(len(rows)):
        e = off_h[(t + 1) * B] if t + 1 < len(rows) else len(idx_h)
        cold = np.bincount(idx_h[off_h[t * B]:e], minlength=rows[t]) <= 256
        a, b = m.table(t).cpu().numpy(), ref.table(t).cpu().numpy()
        assert np.array_equal(a[cold], b[cold]), t
        np.testing.assert_allclose(a[~cold], b[~cold], rtol=3e-4, atol=3e-4)


def test_lds_limit_is_exactly_8192_flagged_lookups_per_table():
    """Forced eligibility on tables whose EVERY lookup is flagged (each row looked up twice: a permutation, repeated): 8192 lookups
    are finished in LDS, 8194 go through the key sort; both give the tables of the hybrid-off route bit for bit."""
    import param_amd

    D, R = 32, 100_000
    for B, expect_lds in ((4096, True), (4097, False)):
        L = 2
        g = torch.Generator(device=DEV).manual_seed(B)
        perm = torch.randperm(R, device=DEV, generator=g)[:B]
        idx = torch.stack([perm, perm.flip(0)], dim=1).reshape(-1).contiguous()         # bag b: rows perm[b], perm[B-1-b]: every row twice
        off = torch.arange(0, B * L + 1, L, device=DEV, dtype=torch.int64)
        grad = torch.randn((B, D), device=DEV, generator=g)
        out = {}
        for hyb in (2, 0):
            param_amd.set_hybrid_tuning(hyb)
            param_amd.set_hybrid_rest(1)
            m = _model([R], D, seed=1)
            m.scatter_add_(grad, idx, off, alpha=1.0, batch=B)
            if hyb:
                st = m.sort_status(idx, off, batch=B)
                assert st["hybrid_tables"] == 1, st
                if expect_lds:
                    assert st["lds_tables"] == 1 and st["lds_pairs"] == B * L and st["pairs_sorted"] == 0, st
                else:
                    assert st["lds_tables"] == 0 and st["pairs_sorted"] == B * L, st
            out[hyb] = m.weights.data.clone()
            del m
        assert torch.equal(out[2], out[0]), B


def test_lds_route_and_round5_route_alternate_on_one_workspace():
    """The route is a knob read at every apply, the verdicts are per request: LDS / round-5 / LDS on ONE module (one workspace)
    leave what three fresh modules leave."""
    import param_amd

    rows, D, B, L = [2_000_000] * 6, 64, 1024, 20
    idx, off = _request(rows, B, L, 0.0, 3)
    grad = torch.randn((B, len(rows) * D), device=DEV)
    param_amd.set_hybrid_tuning(1)
    m = _model(rows, D, seed=1)
    ref = _model(rows, D, seed=1)
    for rest in (1, 0, 1):
        param_amd.set_hybrid_rest(rest)
        m.scatter_add_(grad, idx, off, alpha=-0.01, batch=B)
        st = m.sort_status(idx, off, batch=B)
        assert st["hybrid_tables"] == 6 and st["lds_tables"] == (6 if rest else 0), (rest, st)
        fresh = _model(rows, D, seed=1)
        fresh.weights.data.copy_(ref.weights.data)
        fresh.scatter_add_(grad, idx, off, alpha=-0.01, batch=B)
        ref.weights.data.copy_(fresh.weights.data)
        del fresh
    assert torch.equal(m.weights.data, ref.weights.data)
