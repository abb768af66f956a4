"""BASELINE.json configs[3] and configs[4] AT THEIR STATED SHAPES on one MI355X (``pytest -m gpu``).

configs[3]: train/comms/pt DLRM all-to-all, 26 tables x 10 M rows x 128 (133 GB fp32), per-rank batch 8192, pooling 20.
configs[4]: the MLPerf DLRM-v2 Criteo tables (26 tables, 3 .. 40 M rows, 104.5 GB fp32), per-table multi-hot 1 .. 100.

Both run through the multi-GPU code path -- ``ShardedEmbeddingExchange`` (reference dlrm.py:858-878 forward exchange,
:204-214 gradient exchange) on a ONE-rank RCCL process group: the exchange is then a device-local copy, everything else
(lookup into the send layout, split lists, stream hand-offs, three batches in flight, sorted backward on the owner) is the
N > 1 code.  Checked:
  * forward of the request against torch-ROCm's own embedding_bag (live second oracle) on 3 tables, 1e-5 relative;
  * the received block equals the pooled send buffer bit for bit (1-rank all-to-all = identity);
  * one serial training step against an fp64 accumulation on hot / mid / once / never-touched rows of two tables;
  * five pipelined steps + drain leave the tables BIT-IDENTICAL to five serial steps from the same start (the sorted
    backward is deterministic; the gradients here do not depend on the stale lookups), by row checksums of every table and
    by the touched rows of three tables.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
D, B = 128, 8192


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    import param_amd

    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    param_amd.load_library()
    yield
    param_amd.set_tuning()


@pytest.fixture(scope="module")
def rccl_one_rank():
    """a 1-rank RCCL ("nccl" IS RCCL on ROCm) process group for the module"""
    import torch.distributed as dist

    if dist.is_initialized():
        yield dist
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=DEV)
    yield dist
    dist.destroy_process_group()


def _checksums(m):
    """two position-weighted 64-bit checksums of every table's BITS (a swapped pair of rows or columns changes them)"""
    out = []
    for t in range(len(m.rows)):
        bits = m.table(t).view(torch.int32)
        rs = bits.sum(1, dtype=torch.int64)
        w = torch.arange(rs.numel(), device=rs.device, dtype=torch.int64) % 1000003 + 1
        cs = bits.sum(0, dtype=torch.int64)
        wc = torch.arange(cs.numel(), device=cs.device, dtype=torch.int64) + 1
        out.append((int(rs.sum()), int((rs * w).sum()), int((cs * wc).sum())))
    return out


def _fp64_row_check(m, t, idx, off, grad, alpha, before_rows, sel, counts):
    """rows `sel` (sorted, unique) of table t after ONE scatter-add of the whole request against an fp64 accumulation"""
    s, e = int(off[t * B]), int(off[(t + 1) * B])
    it = idx[s:e]
    lens = (off[t * B + 1:(t + 1) * B + 1] - off[t * B:(t + 1) * B])
    bag_of = torch.repeat_interleave(torch.arange(B, device=DEV), lens)
    pos = torch.isin(it, sel).nonzero().squeeze(1)
    slot = torch.searchsorted(sel, it[pos])
    g = grad[:, t * D:(t + 1) * D].double()
    G = torch.zeros(sel.numel(), D, dtype=torch.float64, device=DEV).index_add_(0, slot, g[bag_of[pos]])
    Gabs = torch.zeros(sel.numel(), D, dtype=torch.float64, device=DEV).index_add_(0, slot, g[bag_of[pos]].abs())
    after = m.table(t)[sel]
    exp = before_rows.double() + alpha * G
    err = (after.double() - exp).abs()
    assert (err <= 1e-5 * (abs(alpha) * Gabs + before_rows.double().abs()) + 1e-30).all(), (t, float(err.max()))
    untouched = counts[sel] == 0
    assert torch.equal(after[untouched], before_rows[untouched]), t


def _pick_rows(counts):
    hot = torch.argsort(counts, descending=True)[:48]
    mid = (((counts >= 2) & (counts <= 256)).nonzero().squeeze(1))[:2000]
    once = ((counts == 1).nonzero().squeeze(1))[:2000]
    never = ((counts == 0).nonzero().squeeze(1))[:2000]
    return torch.unique(torch.cat([hot, mid, once, never]))


def _run_config(dist, rows, pools, check_tables, alpha_lr=-0.01, masked_cus=0):
    import param_amd
    from param_amd.comms.pt.pipeline import ShardedEmbeddingExchange
    from param_amd.indices import tbe_request

    free, _ = torch.cuda.mem_get_info()
    need = sum(rows) * D * 4 + (24 << 30)
    if free < need:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} available")
    T = len(rows)
    m = param_amd.BatchedEmbeddingBagMI355(rows, D, device=DEV, init="normal", seed=7, fused_update=False)
    n_batches = 5
    # batch 0 Zipf (the benchmark's skew), the others alternate; every batch has its own gradient
    reqs = [tbe_request(rows, B, pools, alpha=1.05 if k % 2 == 0 else 0.0, device=DEV, seed=100 + k) for k in range(n_batches)]
    gen = torch.Generator(device=DEV).manual_seed(11)
    grads = [torch.randn(B * T * D, device=DEV, generator=gen) for _ in range(n_batches)]
    idx0, off0 = reqs[0]
    m.check(idx0, off0)
    assert int(off0[-1]) == idx0.numel() == B * sum(pools)

    # ---- forward vs torch-ROCm's own kernel on 3 tables (different shapes under Criteo) ----------------------------
    out = m.lookup(idx0, off0, batch=B)
    for t in check_tables:
        s, e = int(off0[t * B]), int(off0[(t + 1) * B])
        lo = off0[t * B:(t + 1) * B] - s
        ref = torch.nn.functional.embedding_bag(idx0[s:e], m.table(t), lo, mode="sum")
        mag = torch.nn.functional.embedding_bag(idx0[s:e], m.table(t).abs(), lo, mode="sum")
        assert ((out[:, t * D:(t + 1) * D] - ref).abs() <= 1e-5 * mag + 1e-30).all(), t

    k_grad = [0]

    def make_grad(recv, grad_in):            # the dense part's stand-in: a fixed gradient per batch (independent of the lookup)
        grad_in.copy_(grads[k_grad[0] % n_batches])
        k_grad[0] += 1

    def hip_lookup(i, o, out_t):
        m.lookup(i, o, out=out_t, batch=B)

    def hip_backward(g, i, o):
        m.scatter_add_(g, i, o, alpha=alpha_lr, batch=B)

    def fresh_exchange():
        k_grad[0] = 0
        return ShardedEmbeddingExchange(hip_lookup, hip_backward, 1, 0, B, [T * D], DEV, make_grad=make_grad)

    # ---- serial: step 0 checked against fp64 on row slices, then four more ---------------------------------------------
    ex = fresh_exchange()
    sel, before, counts = {}, {}, {}
    for t in check_tables[:2]:
        s, e = int(off0[t * B]), int(off0[(t + 1) * B])
        counts[t] = torch.bincount(idx0[s:e], minlength=rows[t])
        sel[t] = _pick_rows(counts[t])
        before[t] = m.table(t)[sel[t]].clone()
    ex.step_serial(idx0, off0)
    torch.cuda.synchronize()
    assert torch.equal(ex.recv_block(0, 0), ex.pooled[0])                       # 1-rank exchange = identity, bit for bit
    assert torch.equal(ex.pooled[0], out)                                       # the send buffer IS the lookup output
    assert torch.equal(ex.grad[0].view(-1), grads[0])                           # gradient exchange likewise
    for t in check_tables[:2]:
        _fp64_row_check(m, t, idx0, off0, grads[0].view(B, T * D), alpha_lr, before[t], sel[t], counts[t])
    for k in range(1, n_batches):
        ex.step_serial(*reqs[k])
    torch.cuda.synchronize()
    serial_sums = _checksums(m)
    last_i, last_o = reqs[-1]
    touched = {}
    for t in check_tables:
        s, e = int(last_o[t * B]), int(last_o[(t + 1) * B])
        touched[t] = (last_i[s:e].clone(), m.table(t)[last_i[s:e]].clone())
    del ex

    # ---- pipelined (three batches in flight) from the same start -------------------------------------------------------
    m.reset_parameters("normal", 7)                                             # counter-based fill: the same bits again
    if masked_cus:
        # the N > 1 default of bench.py: lookups and backward on a CU-masked HIP stream (hipExtStreamCreateWithCUMask), the rest of
        # the chip left to RCCL's kernels; c10d's stream hand-offs key on torch's current stream, which is the masked one here
        from bench import masked_stream

        ms = masked_stream(masked_cus, torch.device(DEV))
        ms.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ms):
            ex = fresh_exchange()
            for k in range(n_batches):
                ex.step(*reqs[k])
            ex.drain()
        torch.cuda.current_stream().wait_stream(ms)
    else:
        ex = fresh_exchange()
        for k in range(n_batches):
            ex.step(*reqs[k])
        ex.drain()
    torch.cuda.synchronize()
    assert _checksums(m) == serial_sums, "pipelined and serial training steps left different tables"
    for t in check_tables:
        ids, vals = touched[t]
        assert torch.equal(m.table(t)[ids], vals), t
    # the layout the sort chose for this request (ragged multi-hot requests must get table segments too)
    from param_amd.embedding_bag import sort_plan

    plan = sort_plan(m._tables(), idx0, off0, B)
    assert "segmented=1" in plan, plan
    del ex, m, grads, reqs
    torch.cuda.empty_cache()


def test_configs3_26_tables_10M_rows_batch_8192_through_the_exchange(rccl_one_rank):
    """BASELINE configs[3] at its stated shape: 26 x 10 M x 128 fp32, per-rank batch 8192, pooling 20"""
    _run_config(rccl_one_rank, [10_000_000] * 26, [20] * 26, check_tables=[0, 13, 25])


def test_configs4_criteo_tables_multi_hot_through_the_exchange(rccl_one_rank):
    """BASELINE configs[4] at its stated shape: the real Criteo table sizes and multi-hot pooling factors"""
    from param_amd.compute.pt import dataset as ds

    assert ds.criteo_v2_dim == D
    # table 20: 40 M rows x 100-hot (the heaviest), table 0: 40 M rows x 3-hot, table 5: 3 rows (every row hot)
    _run_config(rccl_one_rank, list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot), check_tables=[20, 0, 5])


def test_configs3_through_the_exchange_with_the_compute_stream_on_224_cus(rccl_one_rank):
    """the same shape with the pipelined steps on a 224-CU-masked compute stream (what ``bench.py --gpus N`` runs by default for
    N > 1): serial (unmasked) and pipelined (masked) training steps must leave the same tables, bit for bit"""
    _run_config(rccl_one_rank, [10_000_000] * 26, [20] * 26, check_tables=[0, 13, 25], masked_cus=224)
