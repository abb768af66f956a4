"""Persistent forward (csrc/embbag_fwd_persist.hip, pm_set_forward_persist) -- GPU parity (``pytest -m gpu``).

The kernel's hand-offs (helper wave <-> pooling waves through LDS words) are the new thing; its arithmetic is the classic
kernel's.  Every case is therefore checked three ways: bit-identical to the CPU oracle (sequential fp32 sums), bit-identical
to embbag_fwd_kernel (persist off), and -- for batch slices -- untouched outside the slice.  Every launch runs under a
watchdog: a lost hand-off would be a hang, not a wrong number.
"""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    import param_amd

    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    param_amd.load_library()
    yield
    param_amd.set_forward_persist()
    param_amd.set_tuning()


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _both(m, idx, off, psw=None, **kw):
    """(persistent, classic) outputs of the same request"""
    import param_amd

    outs = []
    for mode in (2, 0):
        param_amd.set_forward_persist(mode, *_both.cfg)
        outs.append(m.lookup(_t(idx), _t(off), None if psw is None else _t(psw), **kw).cpu().numpy())
    torch.cuda.synchronize()
    return outs


_both.cfg = (0, 0, 0, 0)

CONFIGS = [(0, 0, 0, 0), (2, 1, 4, 0), (4, 2, 4, 0), (3, 1, 7, 0), (2, 2, 7, 3), (8, 1, 4, 1), (3, 4, 4, 6)]


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("wdt,D", [(torch.float32, 128), (torch.float32, 16), (torch.float32, 512), (torch.bfloat16, 128), (torch.float16, 256)])
def test_persistent_forward_bit_identical(coracle, cfg, wdt, D):
    """fixed-pooling requests, both layouts, with and without per-sample weights, tile tails (batch not a multiple of any tile),
    batch slices; ring depths 2 .. 8, 1 .. 4 bags per lane group and tile, 4 or 7 pooling waves, 1 .. 6 workgroups per CU"""
    from param_amd import BatchedEmbeddingBagMI355

    _both.cfg = cfg
    rng = np.random.default_rng(D + cfg[0])
    T, R, B, L = 5, 3000, 1531, 7
    for layout in ("bd", "tbd"):
        m = BatchedEmbeddingBagMI355([R] * T, D, dtype=wdt, device=DEV, init="normal", seed=2, layout=layout, fused_update=False)
        tabs = [m.table(t).float().cpu().numpy() for t in range(T)]
        idx = rng.integers(0, R, T * B * L).astype(np.int64)
        off = (np.arange(T * B + 1) * L).astype(np.int64)
        for psw in (None, rng.standard_normal(idx.size).astype(np.float32)):
            exp = coracle.fwd_batched(tabs, idx, off, B, psw=psw, layout=layout)
            per, cla = _both(m, idx, off, psw)
            assert np.array_equal(per, cla) and np.array_equal(per, exp), (layout, psw is not None, cfg)
        import param_amd

        param_amd.set_forward_persist(2, *cfg)
        o2 = torch.full(exp.shape, float("nan"), device=DEV)
        m.lookup(_t(idx), _t(off), out=o2, bag_begin=100, bag_count=1333)
        o2 = o2.cpu().numpy()
        ref = coracle.fwd_batched(tabs, idx, off, B, layout=layout)
        sel = (slice(100, 1433),) if layout == "bd" else (slice(None), slice(100, 1433))
        assert np.array_equal(o2[sel], ref[sel])
        assert np.isnan(np.delete(o2, np.arange(100, 1433), axis=0 if layout == "bd" else 1)).all()


@pytest.mark.parametrize("cfg", [(0, 0, 0, 0), (2, 2, 7, 0)])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_persistent_forward_ragged_but_divisible(coracle, cfg, idt):
    """eligibility is a host-visible fact (lookups divide evenly over the bags); a ragged request that happens to divide --
    most bags empty, some 80 or 160 lookups long -- overflows some tiles' index slots (direct-index path) and leaves lane
    groups of a wave with different bag lengths: same bits as the oracle and the classic kernel; int32 and int64 requests"""
    from param_amd import BatchedEmbeddingBagMI355

    _both.cfg = cfg
    rng = np.random.default_rng(77)
    T, R, D, B = 3, 4000, 128, 1024
    lens = np.zeros(T * B, dtype=np.int64)
    lens[::4] = 80
    lens[:64] = 0
    lens[64:96] = 160
    lens[96:128] = 0
    lens[-1] += (-int(lens.sum())) % (T * B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([rng.integers(0, R, int(lens[t * B:(t + 1) * B].sum())) for t in range(T)]).astype(np.int64)
    assert idx.size % (T * B) == 0
    for layout in ("bd", "tbd"):
        m = BatchedEmbeddingBagMI355([R] * T, D, device=DEV, init="normal", seed=5, layout=layout, fused_update=False)
        tabs = [m.table(t).cpu().numpy() for t in range(T)]
        exp = coracle.fwd_batched(tabs, idx, off, B, layout=layout)
        import param_amd

        outs = []
        for mode in (2, 0):
            param_amd.set_forward_persist(mode, *cfg)
            outs.append(m.lookup(_t(idx).to(idt), _t(off).to(idt), batch=B).cpu().numpy())
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], exp), layout


def test_persistent_forward_many_iterations_mixed_dims_and_default_mode(coracle):
    """a request large enough for mode 1 (>= 8 tiles per resident workgroup): 16 tables of mixed widths, each
    workgroup loops ~10 times (mode 1 = large requests only); checked against the classic kernel bit for bit (the oracle on a slice of the batch), and
    repeated launches on two streams give the same bits (no state survives a launch)"""
    import param_amd
    from param_amd import BatchedEmbeddingBagMI355

    T, B, L = 16, 16384, 10
    rows = [20000 + 1000 * t for t in range(T)]
    dims = [128, 64, 32, 128, 16, 128, 256, 64] * 2
    rng = np.random.default_rng(3)
    m = BatchedEmbeddingBagMI355(rows, dims, device=DEV, init="normal", seed=9, layout="bd", fused_update=False)
    idx = np.concatenate([rng.integers(0, r, B * L) for r in rows]).astype(np.int64)
    off = (np.arange(T * B + 1) * L).astype(np.int64)
    di, do = _t(idx), _t(off)
    param_amd.set_forward_persist(0)
    cla = m.lookup(di, do).cpu().numpy()
    param_amd.set_forward_persist(1)                     # mode 1: this request qualifies (>= 8 tiles per resident workgroup)
    per = m.lookup(di, do).cpu().numpy()
    assert np.array_equal(per, cla)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        per2 = m.lookup(di, do)
    per3 = m.lookup(di, do)
    torch.cuda.synchronize()
    assert np.array_equal(per2.cpu().numpy(), cla) and np.array_equal(per3.cpu().numpy(), cla)
    tabs = [m.table(t).cpu().numpy() for t in range(T)]
    sl = slice(5000, 5600)
    off_s = np.concatenate([[0], np.cumsum(np.full(T * 600, L))]).astype(np.int64)
    idx_s = np.concatenate([idx[(t * B + 5000) * L:(t * B + 5600) * L] for t in range(T)])
    exp = coracle.fwd_batched(tabs, idx_s, off_s, 600, layout="bd")
    assert np.array_equal(per[sl], exp)
