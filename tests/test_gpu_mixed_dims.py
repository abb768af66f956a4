"""Mixed embedding dims in ONE request (round 6) -- GPU parity tests (``pytest -m gpu``), through the C ABI.

BASELINE configs[4] names "mixed-dim tables"; the reference's hook is train/comms/pt/dlrm.py:384-385 (``mixed_dim`` ->
``torch.cat(ly, dim=1)``, dims from :506-557).  With ``pm_embbag_batch.min_dim`` (ABI v7) saying that a request's narrowest table
needs a smaller lane group than its widest, the forward runs the flat-walk kernel with the lane group chosen per table on the
device (sub-groups of 4 .. G lanes).  Bar: bit-exact against the C oracle and against the same request launched without the
hint (one lane-group width for every table) -- a bag is pooled by one sub-group, additions in index order from zero.
"""
import os

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


def _model(rows, dims, dtype=torch.float32, seed=0, hint=True):
    import param_amd

    m = param_amd.BatchedEmbeddingBagMI355(rows, dims, dtype=dtype, device=DEV, init="normal", layout="bd", seed=seed, fused_update=False)
    if not hint:                       # what a C caller that leaves min_dim at 0 gets: one lane-group width for every table
        m._tables().min_dim = 0
    return m


def _request(rows, B, pools, alpha, seed, idt=torch.int64):
    from param_amd.indices import tbe_request

    return tbe_request(rows, B, pools, alpha=alpha, device=DEV, seed=seed, index_dtype=idt)


_MIXED_SEEDS = int(os.environ.get("PARAM_AMD_MIXED_SEEDS", "16"))      # soak runs raise it


@pytest.mark.parametrize("seed", range(_MIXED_SEEDS))
def test_mixed_dim_requests_bit_exact_vs_oracle_and_vs_one_width(seed, coracle):
    """Random requests whose tables mix D in {8, 16, 32, 64, 128} (fp32; 16-bit tables: multiples of 8): fixed pooling or per-table
    pooling factors (1 .. 40), some weighted, int32 / int64 indices, uniform or Zipf rows, batch slices.  The forward with the hint
    (per-table lane groups) == without it (one width) == the C oracle, bit for bit."""
    from oracle.embbag_oracle import BF16, F16

    rng = np.random.default_rng(9100 + seed)
    T = int(rng.integers(2, 12))
    wdt = [torch.float32, torch.float32, torch.bfloat16, torch.float16][seed % 4]
    choices = [8, 16, 32, 64, 128] if wdt == torch.float32 else [8, 16, 32, 64, 128, 256]
    dims = [int(rng.choice(choices)) for _ in range(T)]
    if len(set(dims)) == 1:
        dims[0] = choices[0] if dims[0] != choices[0] else choices[-1]
    rows = [int(rng.choice([5, 300, 20_000, 400_000])) for _ in range(T)]
    B = int(rng.choice([64, 512, 2048]))
    if rng.random() < 0.5:
        pools = [int(rng.choice([1, 2, 20]))] * T                      # an even request: fixed pooling, mixed dims
    else:
        pools = [int(rng.choice([1, 1, 2, 3, 8, 40])) for _ in range(T)]
    idt = torch.int32 if rng.random() < 0.4 else torch.int64
    idx, off = _request(rows, B, pools, 1.05 if rng.random() < 0.4 else 0.0, 50 + seed, idt)
    psw = torch.rand(idx.numel(), device=DEV) if rng.random() < 0.3 else None
    b0 = int(rng.integers(0, B // 4)) if rng.random() < 0.3 else 0
    bc = B - b0 - (int(rng.integers(0, B // 4)) if b0 else 0)
    m = _model(rows, dims, dtype=wdt, seed=seed)
    m1 = _model(rows, dims, dtype=wdt, seed=seed, hint=False)
    out = torch.zeros(B, sum(dims), device=DEV)
    out1 = torch.zeros(B, sum(dims), device=DEV)
    m.lookup(idx, off, per_sample_weights=psw, out=out, batch=B, bag_begin=b0, bag_count=bc)
    m1.lookup(idx, off, per_sample_weights=psw, out=out1, batch=B, bag_begin=b0, bag_count=bc)
    torch.cuda.synchronize()
    assert torch.equal(out, out1), (seed, dims, pools)
    tabs = [m.table(t).float().cpu().numpy() if wdt == torch.float32 else m.table(t).view(torch.int16).cpu().numpy().view(np.uint16)
            for t in range(T)]
    ref = coracle.fwd_batched(tabs, idx.cpu().numpy(), off.cpu().numpy(), B, psw=None if psw is None else psw.cpu().numpy(),
                              dtype={torch.float32: None, torch.bfloat16: BF16, torch.float16: F16}[wdt])
    got = out.cpu().numpy()
    assert np.array_equal(got[b0:b0 + bc], ref[b0:b0 + bc]), (seed, dims, pools)
    assert not got[:b0].any() and not got[b0 + bc:].any()              # bags outside the slice are not written


def test_mixed_dim_backward_still_equals_the_oracle(coracle):
    """The sorted backward sizes its lane groups for the widest table (narrow tables leave lanes idle -- speed only): tables of
    D = 16 .. 128 in one request against the sequential oracle, bit for bit, plain update."""
    rows, dims, B = [50_000, 3_000, 200_000, 64, 9_000], [128, 16, 64, 32, 16], 1024
    pools = [6, 1, 3, 2, 9]
    idx, off = _request(rows, B, pools, 0.0, 5)
    m = _model(rows, dims, seed=4)
    tabs = [m.table(t).cpu().numpy() for t in range(len(rows))]
    grad = torch.randn(B, sum(dims), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    m.scatter_add_(grad, idx, off, alpha=-0.125, batch=B)
    idx_h, off_h, g_h = idx.cpu().numpy(), off.cpu().numpy(), grad.cpu().numpy()
    c0 = 0
    for t in range(len(rows)):
        s, e = off_h[t * B], off_h[(t + 1) * B]
        exp = coracle.bwd_f32(tabs[t].copy(), idx_h[s:e], off_h[t * B:(t + 1) * B] - s, np.ascontiguousarray(g_h[:, c0:c0 + dims[t]]), alpha=-0.125)
        assert np.array_equal(m.table(t).cpu().numpy(), exp), t
        c0 += dims[t]


def test_criteo_mixed_dims_full_size_vs_live_torch_rocm():
    """The 26 Criteo tables at full size with dims by table size (128 / 64 / 32 / 16), batch 8192, multi-hot pooling: the forward
    against torch-ROCm's own EmbeddingBag per table (a live second oracle at full scale; sequential fp32 sums on both sides for
    bags of up to 100 lookups: bit-exact is not promised by torch's kernel, 1e-5 relative is the contract's bar) and bit for bit
    against the launch without the hint."""
    import torch.nn.functional as F
    from param_amd.compute.pt import dataset as ds

    rows, pools = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot)
    dims = ds.criteo_v2_mixed_dims(rows)
    assert sorted(set(dims)) == [16, 32, 64, 128]
    B = 8192
    m = _model(rows, dims, seed=11)
    idx, off = _request(rows, B, pools, 1.05, 7)
    out = m.lookup(idx, off, batch=B)
    m._tables().min_dim = 0
    m._tables()._req = {}
    out1 = m.lookup(idx, off, batch=B)
    torch.cuda.synchronize()
    assert torch.equal(out, out1)
    c0 = 0
    for t in range(len(rows)):
        s, e = int(off[t * B]), int(off[(t + 1) * B])
        ref = F.embedding_bag(idx[s:e], m.table(t), off[t * B:(t + 1) * B] - s, mode="sum")
        mag = F.embedding_bag(idx[s:e], m.table(t).abs(), off[t * B:(t + 1) * B] - s, mode="sum")
        err = (out[:, c0:c0 + dims[t]] - ref).abs()
        assert bool((err <= 1e-5 * mag + 1e-30).all()), t
        c0 += dims[t]


@pytest.mark.parametrize("T", [300, 1100])
def test_flat_walk_launch_with_many_tables_empty_tables_and_slices(T, coracle):
    """The compact flat-walk launch keeps one prefix entry per table in LDS (requests of up to 1024 tables; larger ones keep the
    T x tiles grid): hundreds of one-hot / short-bag tables of mixed widths, some of them with EMPTY bags only and some ragged, a batch
    slice with bag_begin > 0, int32 indices -- the forward equals the C oracle bit for bit in every launch shape
    (pm_set_forward_tuning(flat_grid) 0 = round 3's grid, 1 = the library's, 64 = few workgroups walking many tiles each)."""
    import param_amd

    rng = np.random.default_rng(77 + T)
    dims = [int(rng.choice([8, 16, 32, 64])) for _ in range(T)]
    dims[0], dims[-1] = 64, 8
    rows = [int(rng.choice([3, 50, 4000])) for _ in range(T)]
    B = 96
    lens = []
    for t in range(T):
        kind = rng.random()
        if kind < 0.1:
            lens.append(np.zeros(B, dtype=np.int64))                         # a table nobody looks up
        elif kind < 0.6:
            lens.append(np.ones(B, dtype=np.int64))                          # one-hot
        else:
            lens.append(rng.integers(0, 4, B))                               # ragged, empty bags among them
    lens = np.concatenate(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([rng.integers(0, rows[t], int(lens[t * B:(t + 1) * B].sum())) for t in range(T)]).astype(np.int64)
    m = _model(rows, dims, seed=5)
    it, ot = torch.from_numpy(idx).to(DEV).to(torch.int32), torch.from_numpy(off).to(DEV).to(torch.int32)
    tabs = [m.table(t).cpu().numpy() for t in range(T)]
    ref = coracle.fwd_batched(tabs, idx, off, B)
    try:
        for grid in (1, 0, 64):
            param_amd.set_forward_tuning(flat_grid=grid)
            for b0, bc in ((0, B), (17, 60)):
                out = torch.full((B, sum(dims)), -7.0, device=DEV)
                m.lookup(it, ot, out=out, batch=B, bag_begin=b0, bag_count=bc)
                got = out.cpu().numpy()
                assert np.array_equal(got[b0:b0 + bc], ref[b0:b0 + bc]), (T, grid, b0)
                assert (got[:b0] == -7.0).all() and (got[b0 + bc:] == -7.0).all()       # bags outside the slice are not written
    finally:
        param_amd.set_forward_tuning()
