#!/usr/bin/env python3
"""bench.py -- EmbeddingBag lookups/s (+ achieved HBM GB/s) on MI355X, 1..8 GPUs.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 under torch.distributed.run,
one rank per GPU).  W untimed warm-up steps, then exactly K steps bracketed by barrier +
``torch.cuda.synchronize()``, MAX over ranks, rank 0 prints ONE JSON line.

A *step* is one pass of the hot path over one synthetic batch:
  N == 1 : one batched EmbeddingBag forward launch over all local tables (BASELINE.json
           configs[1]: 64 tables x 10M rows x 128 dim, fp32, Zipf indices -- 64 fp32 tables are
           327.68 GB > 288 GB HBM, so one GPU runs the largest table count that fits, 48).
  N  > 1 : the 64 tables are sharded table-wise (64/N per GPU); every rank looks up the GLOBAL
           batch (N x 8192 bags) for its tables and the pooled embeddings go back to
           batch-parallel layout with ONE RCCL all-to-all per table group, issued per group so
           that it overlaps the next group's lookup (reference dlrm.py:858-878 /
           pytorch_dist_backend.py:214-234).  Per-GPU work is fixed as N grows ("weak").

``value`` = lookups of all ranks / wall time, inputs resident in HBM.  ``roofline`` is the
forward kernel's ALGORITHMIC bytes (546 B/lookup at D=128 fp32 L=20) over its
average launch duration measured with HIP events on the launch stream.  ``cpu_baseline`` times
the reference's CPU engine (torch.nn.EmbeddingBag, pytorch_emb.py:37-45 protocol) and the
1-core C oracle on a bounded sample on the box's host cores (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import param_amd  # noqa: E402
from param_amd.compute.pt.pytorch_emb import algorithmic_bytes  # noqa: E402
from param_amd.indices import tbe_request  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
_DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", choices=["uniform-tables", "criteo"], default="uniform-tables",
                   help="criteo: the 26 MLPerf DLRM-v2 tables with per-table multi-hot pooling (BASELINE.json configs[4]), N=1 only")
    p.add_argument("--tables", type=int, default=64, help="logical table count of the workload")
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--batch", type=int, default=8192, help="bags per table per rank")
    p.add_argument("--pooling", type=int, default=20)
    p.add_argument("--alpha", type=float, default=1.05, help="Zipf exponent of the headline run (0 = uniform)")
    p.add_argument("--dtype", choices=sorted(_DT), default="fp32")
    p.add_argument("--a2a-groups", type=int, default=1, help="N>1: table groups pipelined against the all-to-all inside a step")
    p.add_argument("--pipeline-depth", type=int, default=2,
                   help="N>1: steps in flight; 2 = step k's exchange completes under step k+1's lookups (double-buffered)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--dist-debug", action="store_true", help="run the N>1 (pipeline + RCCL) code path even at world size 1")
    p.add_argument("--no-uniform", action="store_true", help="skip the extra uniform-index (pure HBM) measurement")
    p.add_argument("--bwd", action="store_true", help="also time the scatter-add backward (extra field)")
    p.add_argument("--unroll", type=int, default=0)
    p.add_argument("--bags-per-block", type=int, default=0)
    p.add_argument("--xcd-affine", type=int, default=-1)
    p.add_argument("--nt-loads", type=int, default=-1)
    return p.parse_args()


def time_steps(fn, steps, warmup, barrier):
    """W warm-ups, then K steps between (barrier + device sync); HIP events on the launch stream
    give the average device time per step."""
    for _ in range(warmup):
        fn()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) * 1e-3 / steps


def cpu_baseline(table0: torch.Tensor, idx0: torch.Tensor, B: int, L: int, budget_s: float = 12.0):
    """Reference CPU engine on a bounded sample: ONE table of the workload (same rows/dim/indices
    as table 0 on the GPU), measure_cpu protocol of pytorch_emb.py:37-45."""
    from param_amd.compute.pt.pytorch_emb import measure_cpu

    W = table0.float().cpu()
    idx = idx0.cpu()
    off = torch.arange(B, dtype=torch.int64) * L
    emb = torch.nn.EmbeddingBag(W.shape[0], W.shape[1], mode="sum", _weight=W)
    res = {}
    nthreads = torch.get_num_threads()
    lookups = B * L

    def run(tag, threads, no_grad, budget):
        torch.set_num_threads(threads)
        ctx = torch.no_grad() if no_grad else torch.enable_grad()
        with ctx:
            t1, _ = measure_cpu(1, 1, emb, idx, off)
            steps = max(2, min(200, int(budget / max(t1, 1e-4))))
            el, _ = measure_cpu(1, steps, emb, idx, off)
        res[tag] = {"lookups_per_s": lookups * steps / el, "s_per_step": el / steps, "threads": threads, "steps": steps}

    with torch.no_grad():  # discarded: first touch of the table pages + thread-pool spin-up
        measure_cpu(1, 3, emb, idx, off)
    run("all_threads_no_grad", nthreads, True, budget_s * 0.4)
    run("all_threads_grad_on_param_default", nthreads, False, budget_s * 0.2)
    run("one_thread_no_grad", 1, True, budget_s * 0.2)
    torch.set_num_threads(nthreads)
    # 1-core C oracle ("port") on a smaller slice of the same request
    try:
        from oracle.embbag_oracle import COracle

        nb = min(B, 2048)
        orc = COracle()
        Wn, In, On = W.numpy(), idx[: nb * L].numpy(), off[:nb].numpy()
        t0 = time.perf_counter()
        orc.fwd(Wn, In, On)
        res["c_oracle_1core"] = {"lookups_per_s": nb * L / (time.perf_counter() - t0), "bags": nb}
    except Exception as exc:  # the checker is optional for the baseline leg
        res["c_oracle_1core"] = {"error": str(exc)}
    best = res["all_threads_no_grad"]
    return {
        "value": best["lookups_per_s"], "unit": "lookups/s", "cores": nthreads, "kind": "port",
        "sample": (f"torch.nn.EmbeddingBag(sum) on host, 1 table {W.shape[0]}x{W.shape[1]} fp32, batch {B}, "
                   f"pool {L}, same indices as GPU table 0, {best['steps']} steps after 1 warm-up, "
                   f"{nthreads} threads, no_grad (reference engine + its measure_cpu protocol)"),
        "host_cpu_count": os.cpu_count(), "modes": res,
    }


def main():
    a = parse()
    # stdout must carry exactly ONE JSON line: RCCL prints a version banner to the C-level stdout and
    # torch may warn there too, so fd 1 is pointed at stderr for the whole run and the JSON line goes
    # to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or a.dist_debug  # --dist-debug: exercise the N>1 code path on one GPU (1-rank RCCL group)
    if multi:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" IS RCCL on ROCm

    def barrier():
        if dist is not None:
            dist.barrier()

    param_amd.load_library()  # fail loudly before allocating anything
    param_amd.set_tuning(a.unroll, a.bags_per_block, a.xcd_affine, a.nt_loads)
    dtype = _DT[a.dtype]
    esize = torch.empty(0, dtype=dtype).element_size()
    D, R, L, B_local = a.dim, a.rows, a.pooling, a.batch

    # ---- shard the tables -------------------------------------------------------------------
    free, total = torch.cuda.mem_get_info()
    table_bytes = R * D * esize
    if world == 1:
        fit = int((free - (40 << 30)) // table_bytes)  # headroom for outputs, gradients, sort scratch
        T_loc = min(a.tables, fit)
        if T_loc >= 8:
            T_loc = T_loc // 8 * 8  # keep the XCD-affine mapping (table t -> XCD t % 8)
    else:
        assert a.tables % world == 0, "table count must divide over ranks"
        T_loc = a.tables // world
    assert T_loc >= 1, "not enough HBM for one table"
    B_glob = B_local * world  # table-wise sharding: every rank serves the global batch for its tables
    rows_list, pool_list = [R] * T_loc, [L] * T_loc
    if a.workload == "criteo":
        from param_amd.compute.pt import dataset as ds

        assert not multi, "--workload criteo is a single-GPU configuration of bench.py"
        rows_list, pool_list, D = list(ds.criteo_v2_rows), list(ds.criteo_v2_multi_hot), ds.criteo_v2_dim
        T_loc, a.tables = len(rows_list), len(rows_list)

    model = param_amd.BatchedEmbeddingBagMI355(rows_list, D, dtype=dtype, device=dev, init="normal",
                                               seed=1000 + rank, fused_update=False)
    groups = 1 if not multi else max(1, min(a.a2a_groups, T_loc))
    while T_loc % groups:
        groups -= 1
    Tg = T_loc // groups

    def make_request(alpha, seed):
        return tbe_request(rows_list, B_glob, pool_list if a.workload == "criteo" else L, alpha=alpha, device=dev,
                           seed=seed + 17 * rank)

    idx, off = make_request(a.alpha, 1)
    lookups_step_rank = B_glob * sum(pool_list)
    # SURVEY 8d per-unit figures summed over tables: per lookup D*e + 8 read, per bag 8 read + D*4 written
    alg_bytes = sum(algorithmic_bytes(1, B_glob, Lt, D, esize) for Lt in pool_list)

    # ---- the step ---------------------------------------------------------------------------
    if not multi:
        out = torch.empty((B_glob, T_loc * D), dtype=torch.float32, device=dev)

        def step(i=idx, o=off):
            model.lookup(i, o, out=out, batch=B_glob)
    else:
        from param_amd.comms.pt.pipeline import LookupAllToAll, split_request_by_group
        from param_amd.embedding_bag import _TableSet, _fwd

        tsets = [_TableSet([model.table(g * Tg + t) for t in range(Tg)], "bd") for g in range(groups)]

        def hip_lookup(g, ig, og, out_g):  # the HIP batched forward of table group g
            _fwd(tsets[g], ig, og, B_glob, out=out_g)

        # lookup(g) -> RCCL all_to_all(g) on the process group's own stream, under lookup(g+1)
        pipe = LookupAllToAll(hip_lookup, world, B_local, [Tg * D] * groups, dev, depth=a.pipeline_depth)

        def split_request(i, o):
            return split_request_by_group(i, o, T_loc, groups, B_glob)

        reqs = split_request(idx, off)

        def step(rq=None):
            pipe.step(reqs if rq is None else rq)

    wall, dev_s = time_steps(step, a.steps, a.warmup, barrier)   # the closing device sync covers exchanges still in flight
    if multi:
        pipe.flush()
    if dist is not None:
        t = torch.tensor([wall, dev_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_s = t.tolist()

    result = {
        "metric": "EmbeddingBag lookups/s + achieved HBM GB/s; all-to-all bus-BW at 1/2/4/8 GPU",
        "value": lookups_step_rank * world * a.steps / wall,
        "unit": "lookups/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[a.dtype],
        "data": "synthetic",
        "config": {
            "workload": ((f"batched EmbeddingBag(sum) fwd, {a.tables} tables x {R} rows x {D} dim {a.dtype}, "
                          f"batch {B_local}/rank, pool {L}, " if a.workload != "criteo" else
                          f"batched EmbeddingBag(sum) fwd, MLPerf DLRM-v2 Criteo tables (26 tables, {sum(rows_list)} rows, "
                          f"dim {D}, {a.dtype}), batch {B_local}, multi-hot pooling {sum(pool_list)} lookups/sample, ")
                         + f"Zipf alpha={a.alpha} (reference pmf, per-bag dedupe)"
                         + (f"; 1 GPU holds {T_loc} of {a.tables} tables ({T_loc * table_bytes / 1e9:.1f} GB): "
                            f"{a.tables} x {table_bytes / 1e9:.2f} GB exceeds 288 GB HBM" if world == 1 and T_loc < a.tables else "")
                         + (f"; table-wise sharded {T_loc}/GPU, global batch {B_glob}, pooled all-to-all in "
                            f"{groups} table group(s), {a.pipeline_depth} step(s) in flight (exchange of step k under the "
                            f"lookups of step k+1)" if world > 1 else "")),
            "tables_total": a.tables if world > 1 else T_loc, "tables_per_gpu": T_loc,
            "rows": R if a.workload != "criteo" else "criteo_v2 (3 .. 40M rows, 204.2 M total)", "dim": D,
            "batch_per_rank": B_local, "global_batch": B_glob, "pooling": L, "alpha": a.alpha,
            "index_dtype": "int64", "parallelism": "1gpu" if world == 1 else f"table-wise x{world} + all-to-all",
            "lookups_per_step": lookups_step_rank * world,
        },
    }

    # roofline of the dominant kernel (forward lookup): algorithmic bytes / avg launch duration
    if not multi:
        kern_s = dev_s
    else:  # time the lookups alone (no a2a) for the kernel roofline
        _, kern_s = time_steps(lambda: pipe.lookups_only(reqs), max(5, a.steps // 2), 2, barrier)
        if dist is not None:
            t = torch.tensor([kern_s], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            kern_s = t.item()
    ach = alg_bytes / kern_s / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world == 1 and os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            key = f"T{T_loc}_R{R}_D{D}_B{B_local}_L{L}_a{a.alpha}_{a.dtype}" if a.workload != "criteo" else "criteo"
            traffic = pm.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    result["roofline"] = {
        "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
        "traffic": traffic, "kernel": "embbag_fwd_kernel", "algorithmic_bytes_per_launch": alg_bytes,
        "bytes_per_lookup": alg_bytes / lookups_step_rank, "avg_launch_s": kern_s,
        "lookups_per_s_kernel": lookups_step_rank / kern_s,
    }

    if multi:
        a2a_bytes = world * B_local * T_loc * D * 4  # output tensor bytes per rank (reference memSize)
        _, a2a_s = time_steps(pipe.all_to_all_only, max(5, a.steps // 2), 2, barrier)
        t = torch.tensor([a2a_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        a2a_s = t.item()
        alg_bw = a2a_bytes / a2a_s / 1e9
        result["all_to_all"] = {
            "bytes_per_rank": a2a_bytes, "avg_s": a2a_s, "algbw_GBps": alg_bw,
            "busbw_GBps": alg_bw * (world - 1) / world,  # pytorch_backend_utils.py:221-234
            "xgmi_bound_GBps": (world - 1) * 153.0, "overlap_step_s": dev_s, "lookup_only_s": kern_s,
        }

    # extra: uniform indices = no cache help, the pure-HBM run (SURVEY.md 8d "roofline-defining run")
    if not a.no_uniform and a.alpha != 0.0:
        ui, uo = make_request(0.0, 2)
        if not multi:
            _, us = time_steps(lambda: step(ui, uo), max(5, a.steps // 2), 2, barrier)
        else:
            ur = split_request(ui, uo)
            _, us = time_steps(lambda: pipe.lookups_only(ur), max(5, a.steps // 2), 2, barrier)
        if dist is not None:
            t = torch.tensor([us], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            us = t.item()
        result["uniform"] = {"lookups_per_s": lookups_step_rank * world / us, "achieved_GBps": alg_bytes / us / 1e9,
                             "frac": alg_bytes / us / 1e9 / HBM_PEAK_GBPS, "avg_launch_s": us}
        del ui, uo

    if a.bwd and world == 1:
        grad = torch.randn((B_glob, T_loc * D), dtype=torch.float32, device=dev)
        bwd_bytes = lookups_step_rank * (2 * D * esize + 8) + T_loc * B_glob * (D * 4 + 8)
        n_b = max(5, a.steps // 2)
        _, bs = time_steps(lambda: model.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B_glob), n_b, 2, barrier)
        model.sort_indices(idx, off, batch=B_glob)
        _, ba = time_steps(lambda: model.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B_glob, presorted=True),
                           n_b, 2, barrier)
        _, bt = time_steps(lambda: model.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B_glob, method="atomic"),
                           3, 1, barrier)
        result["bwd_scatter_add"] = {
            "method": "sorted (stable radix sort of (table,row) keys + one read-modify-write per touched row, no atomics)",
            "lookups_per_s": lookups_step_rank / bs, "achieved_GBps": bwd_bytes / bs / 1e9,
            "frac": bwd_bytes / bs / 1e9 / HBM_PEAK_GBPS, "avg_s_sort_plus_apply": bs, "avg_s_apply_only": ba,
            "apply_only_GBps": bwd_bytes / ba / 1e9, "apply_only_frac": bwd_bytes / ba / 1e9 / HBM_PEAK_GBPS,
            "bytes_per_lookup": bwd_bytes / lookups_step_rank,
            "atomic_kernel_s": bt, "atomic_kernel_frac": bwd_bytes / bt / 1e9 / HBM_PEAK_GBPS}
        # the optimizer the reference configures for its TBE ops (EXACT_ROWWISE_ADAGRAD): same kernels, fused epilogue,
        # + 4 B of state read-modify-write per touched row
        try:
            model.learning_rate = 1e-6
            _, ag = time_steps(lambda: model.adagrad_step_(grad, idx, off, batch=B_glob, presorted=True), n_b, 2, barrier)
            result["bwd_rowwise_adagrad"] = {"avg_s_apply_only": ag, "apply_only_GBps": bwd_bytes / ag / 1e9,
                                             "apply_only_frac": bwd_bytes / ag / 1e9 / HBM_PEAK_GBPS,
                                             "avg_s_sort_plus_apply": ag + (bs - ba)}
        except Exception as exc:  # wide rows (> 64 lanes x vector) have no fused Adagrad
            result["bwd_rowwise_adagrad"] = {"error": str(exc)}
        # BASELINE configs[2]: one fwd + bwd training step.  The key sort needs only the request, so it runs on a
        # second HIP stream UNDER the forward lookup; the apply kernels wait for both.
        side = torch.cuda.Stream(device=dev)
        ev_sorted = torch.cuda.Event()
        out_fb = torch.empty((B_glob, T_loc * D), dtype=torch.float32, device=dev)

        def fwd_bwd_step():
            main = torch.cuda.current_stream()
            side.wait_stream(main)                       # the previous step's apply must be done with the scratch
            with torch.cuda.stream(side):
                model.sort_indices(idx, off, batch=B_glob)
                ev_sorted.record(side)
            model.lookup(idx, off, out=out_fb, batch=B_glob)
            main.wait_event(ev_sorted)
            model.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B_glob, presorted=True)

        def fwd_bwd_serial():
            model.lookup(idx, off, out=out_fb, batch=B_glob)
            model.scatter_add_(grad, idx, off, alpha=-1e-6, batch=B_glob)

        _, fb = time_steps(fwd_bwd_step, n_b, 2, barrier)
        _, fs = time_steps(fwd_bwd_serial, n_b, 2, barrier)
        fb_bytes = alg_bytes + bwd_bytes
        result["fwd_bwd_step"] = {"avg_s_sort_overlapped": fb, "avg_s_serial": fs, "lookups_per_s": lookups_step_rank / fb,
                                  "algorithmic_GBps": fb_bytes / fb / 1e9, "frac": fb_bytes / fb / 1e9 / HBM_PEAK_GBPS,
                                  "bytes_per_lookup": fb_bytes / lookups_step_rank}
        del grad, out_fb

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            L0 = pool_list[0]  # table 0 of the workload: same rows / dim / indices as on the GPU
            result["cpu_baseline"] = cpu_baseline(model.table(0), idx[: B_glob * L0], B_glob, L0)
        except Exception as exc:
            result["cpu_baseline"] = {"value": None, "unit": "lookups/s", "cores": torch.get_num_threads(),
                                      "kind": "port", "sample": f"failed: {exc}"}
    elif rank == 0:
        result["cpu_baseline"] = None if a.no_cpu_baseline else {
            "value": None, "unit": "lookups/s", "cores": None, "kind": "port", "sample": "timed at N=1 only"}

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
