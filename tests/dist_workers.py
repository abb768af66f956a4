"""Worker bodies for the world_size-2 gloo/CPU tests (spawned by tests/test_dist_gloo.py).

The collectives run for real over loopback; the embedding lookup is a TEST STUB
(torch.nn.functional.embedding_bag) injected into the pipeline -- the product lookup is the HIP
kernel and never runs on CPU."""
import os
import socket
import types

import torch
import torch.distributed as dist


def free_port() -> int:
    """A TCP port for a rendezvous: below the kernel's ephemeral range (32768+), so it cannot collide with the local port
    of some earlier store client's connection still in TIME_WAIT (seen as EADDRINUSE once in a long GPU suite run), never
    handed out twice by this process, and bindable right now."""
    import itertools
    import os as _os

    global _PORTS
    try:
        _PORTS
    except NameError:
        _PORTS = itertools.count(12000 + (_os.getpid() * 37) % 18000)
    for p in _PORTS:
        p = 12000 + (p - 12000) % 20000
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p


def _env(rank, world, port):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "LOCAL_SIZE": str(world)})
    torch.set_num_threads(1)


def _torch_host_row_codec(bits):
    """torch's CPU operators for the row-wise quantised formats (the operators oracle/rowquant.py is pinned to): what the gloo
    workers inject as ``MI355XBackend.host_row_codec`` -- the product backend has no host codec of its own"""
    q = torch.ops.quantized
    return {8: (q.embedding_bag_byte_prepack, q.embedding_bag_byte_unpack),
            4: (q.embedding_bag_4bit_prepack, q.embedding_bag_4bit_unpack),
            2: (q.embedding_bag_2bit_prepack, q.embedding_bag_2bit_unpack)}[bits]


def _backend(rank, world, port):
    from param_amd.comms.pt import comms_utils
    from param_amd.comms.pt.mi355_backend import MI355XBackend

    MI355XBackend.host_row_codec = staticmethod(_torch_host_row_codec)      # (class-level: the sweep driver builds its own instance)
    env = comms_utils.read_comms_env_vars()
    assert env == {"world_size": world, "local_size": world, "global_rank": rank, "local_rank": rank}
    info = comms_utils.bootstrap_info_holder("127.0.0.1", str(port), 0, env)
    params = types.SimpleNamespace(device="cpu", backend="gloo")
    bf = MI355XBackend(info, params)
    bf.initialize_backend("127.0.0.1", str(port), backend="gloo")
    return bf


def backend_collectives(rank, world, port):
    from param_amd.comms.pt.pytorch_backend_utils import collectiveArgsHolder

    _env(rank, world, port)
    bf = _backend(rank, world, port)
    try:
        # get_next_group() walks the process groups round-robin (reference pytorch_dist_backend.py:1200,1251): one group
        # after initialize_backend, both after initialize_groups
        assert bf.get_next_group() is bf.get_default_group() and bf.get_next_group() is bf.get_default_group()
        bf.initialize_groups({0: list(range(world)), 1: list(range(world))}, backend="gloo", force_new_group=True)
        seen = [bf.get_next_group() for _ in range(4)]
        assert seen[0] is not seen[1] and seen[0] is seen[2] and seen[1] is seen[3] and bf.get_num_pgs() == 2
        ca = collectiveArgsHolder()
        ca.world_size, ca.global_rank, ca.group, ca.device = world, rank, bf.get_default_group(), bf.get_device()
        assert bf.get_world_size() == world and bf.get_global_rank() == rank and bf.get_device().type == "cpu"
        # all_to_allv with uneven splits: rank r sends (r+1) elements to rank 0 and 2 to rank 1
        send_counts = [[1, 2], [2, 2]][rank]
        recv_counts = [[1, 2], [2, 2]]
        recv_counts = [recv_counts[src][rank] for src in range(world)]
        ca.ipTensor = torch.arange(sum(send_counts), dtype=torch.float32) + 100 * rank
        ca.opTensor = torch.full((sum(recv_counts),), -1.0)
        ca.ipTensor_split, ca.opTensor_split, ca.asyncOp = send_counts, recv_counts, True
        w = bf.all_to_allv(ca, retFlag=True)
        assert w is not None and ca.waitObj == [w]          # async: queued AND returned
        bf.complete_accel_ops(ca)
        assert ca.waitObj == []
        expect = {0: [0.0, 100.0, 101.0], 1: [1.0, 2.0, 102.0, 103.0]}[rank]
        assert ca.opTensor.tolist() == expect, (rank, ca.opTensor.tolist())
        assert bf.get_mem_size(ca) == 4 * len(expect)
        # list-form all_to_all (gloo has none: the backend flattens it)
        ca.asyncOp = False
        ca.ipTensor = [torch.full((3,), float(10 * rank + j)) for j in range(world)]
        ca.opTensor = [torch.empty(3) for _ in range(world)]
        assert bf.all_to_all(ca, retFlag=True) is None and ca.waitObj == []
        for src in range(world):
            assert ca.opTensor[src].tolist() == [float(10 * src + rank)] * 3
        # equal-split single form, all_reduce, barrier
        ca.ipTensor, ca.opTensor = torch.arange(4.0) + 10 * rank, torch.empty(4)
        ca.ipTensor_split = ca.opTensor_split = [2, 2]
        bf.all_to_all_single(ca)
        assert ca.opTensor.tolist() == [[0.0, 1.0, 10.0, 11.0], [2.0, 3.0, 12.0, 13.0]][rank]
        ca.ipTensor = torch.ones(5) * (rank + 1)
        ca.op = bf.get_reduce_op("sum")
        bf.all_reduce(ca)
        assert ca.ipTensor.tolist() == [3.0] * 5
        bf.sync_barrier(ca)
        assert bf.getBusBW("all_to_allv", 10.0, ca) == 5.0 and bf.getBusBW("all_reduce", 10.0, ca) == 10.0
    finally:
        bf.shutdown()


def comms_sweep(rank, world, port, outdir):
    import contextlib
    import io
    import json

    from param_amd.comms.pt import comms

    _env(rank, world, port)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = comms.main(["--master-ip", "127.0.0.1", "--master-port", str(port), "--b", "64", "--e", "1024", "--f", "4",
                          "--n", "3", "--w", "1", "--z", "1", "--c", "1", "--collective",
                          "all_to_allv,all_to_all,all_reduce", "--backend", "gloo", "--device", "cpu"])
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"results": res, "stdout": buf.getvalue()}, f)


def _stub_tables(rows, D):
    """deterministic per-GLOBAL-table weights so every rank can compute any table's expected output"""
    return [torch.randn(r, D, generator=torch.Generator().manual_seed(1000 + r)) for r in rows]


def dlrm_sparse_path(rank, world, port):
    """DLRMSparsePath over gloo with a torch STUB lookup: after lengths/indices exchange, lookup and the
    pooled all-to-all, every rank holds [B, sum_t D] pooled embeddings of ITS samples for ALL tables."""
    from param_amd.comms.pt import dlrm as D_
    from param_amd.comms.pt.pytorch_backend_utils import collectiveArgsHolder

    _env(rank, world, port)
    bf = _backend(rank, world, port)
    try:
        ln_emb, D, B, L = [40, 50, 60], 4, 5, 3          # 3 tables over 2 ranks -> [2, 1]
        _, per_rank = D_.get_split_lengths_by_len(len(ln_emb), rank, world)
        assert per_rank == [2, 1]
        ca = collectiveArgsHolder()
        ca.world_size, ca.global_rank, ca.group, ca.device = world, rank, bf.get_default_group(), bf.get_device()
        all_tables = _stub_tables(ln_emb, D)
        mine = all_tables[D_.get_slice_sparse(rank, per_rank, world)]
        n_glob = world * B

        def lookup(idx, off, out):
            for t, Wt in enumerate(mine):
                s, e = int(off[t * n_glob]), int(off[(t + 1) * n_glob])
                out[:, t * D:(t + 1) * D] = torch.nn.functional.embedding_bag(
                    idx[s:e], Wt, off[t * n_glob:(t + 1) * n_glob] - s, mode="sum")

        path = D_.DLRMSparsePath(bf, ca, per_rank, D, B, lookup, None)
        gen = torch.Generator().manual_seed(50 + rank)
        lengths, indices = D_.generate_sparse_batch(ln_emb, B, L, False, torch.device("cpu"), gen)
        assert lengths.shape == (len(ln_emb) * B,) and int(lengths.sum()) == indices.numel() and int(lengths.min()) >= 1
        idx_tbe, off_tbe = path.sparse_data_dist(lengths, indices)
        assert off_tbe.numel() == per_rank[rank] * n_glob + 1 and int(off_tbe[-1]) == idx_tbe.numel()
        ly = path.apply_emb(idx_tbe, off_tbe)
        pooled, out_split, in_split = path.alltoallv_fwd(ly)
        assert pooled.shape == (B, len(ln_emb) * D)
        # expected: my sample b, global table t -> sum of that table's rows at MY indices
        offs = torch.cumsum(lengths, 0) - lengths
        for t in range(len(ln_emb)):
            for b in range(B):
                s = int(offs[t * B + b])
                rows = indices[s:s + int(lengths[t * B + b])]
                assert torch.allclose(pooled[b, t * D:(t + 1) * D], all_tables[t][rows].sum(0), atol=1e-5), (rank, t, b)
        # backward exchange: gradient of pooled goes back to the table owners in [N_global, E_local] layout
        grad = torch.arange(pooled.numel(), dtype=torch.float32).view_as(pooled) + 1000 * rank
        g_loc = path.alltoallv_bwd(grad, out_split, in_split)
        assert g_loc.shape == (n_glob, per_rank[rank] * D)
        col0 = sum(per_rank[:rank]) * D
        for src in range(world):
            exp = (torch.arange(B * len(ln_emb) * D, dtype=torch.float32).view(B, -1) + 1000 * src)[:, col0:col0 + per_rank[rank] * D]
            assert torch.equal(g_loc[src * B:(src + 1) * B], exp), (rank, src)
        kinds = [c["comms"] for c in path.commDetails]
        assert kinds == ["all_to_all"] * 4 and path.commDetails[2]["out_split"] == [B * 2 * D, B * 1 * D]
    finally:
        bf.shutdown()


def dlrm_driver(rank, world, port, outdir, extra_json="[]", num_batches="4", flags_json=""):
    import contextlib
    import io
    import json

    from param_amd.comms.pt import dlrm as D_

    _env(rank, world, port)
    os.chdir(outdir)

    def factory(rows, D, dev, dtype):
        tabs = _stub_tables(rows, D)

        def lookup(idx, off, out):
            n = out.shape[0]
            for t, Wt in enumerate(tabs):
                s, e = int(off[t * n]), int(off[(t + 1) * n])
                out[:, t * D:(t + 1) * D] = torch.nn.functional.embedding_bag(idx[s:e], Wt, off[t * n:(t + 1) * n] - s, mode="sum")
        return lookup, None

    bench = D_.commsDLRMBench()
    import argparse
    if flags_json:          # the flag list of another fixture (tests/golden/dlrm_np4/flags.json), as the reference was given it
        args = bench.readArgs(argparse.ArgumentParser(), ["--master-ip", "127.0.0.1", "--master-port", str(port)]
                              + __import__("json").loads(flags_json) + ["--data-generation", "random"])
    else:
        args = bench.readArgs(argparse.ArgumentParser(), [
            "--master-ip", "127.0.0.1", "--master-port", str(port), "--backend", "gloo", "--device", "cpu",
            "--mini-batch-size", "8", "--num-batches", num_batches, "--warmup-batches", "1", "--arch-mlp-bot", "16-8",
            "--arch-mlp-top", "8-1", "--arch-sparse-feature-size", "8", "--arch-embedding-size", "100-200-300-400",
            "--num-indices-per-lookup", "5", "--print-comms", "--data-generation", "random"] + __import__("json").loads(extra_json))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rep = bench.run(args, lookup_factory=factory)
    with open(os.path.join(outdir, f"report{rank}.json"), "w") as f:
        json.dump({"report": rep, "stdout": buf.getvalue()}, f)


def pipeline_layout(rank, world, port):
    from param_amd.comms.pt.pipeline import LookupAllToAll, split_request_by_group

    _env(rank, world, port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T_loc, groups, D, B_local, L, R = 4, 2, 8, 3, 5, 50
        Tg, B_glob = T_loc // groups, world * B_local
        g = torch.Generator().manual_seed(100 + rank)
        tables = [torch.randn(R, D, generator=g) for _ in range(T_loc)]          # this rank's tables
        # every rank must look up the SAME global batch: generate it from a shared seed per owner rank
        def request_for(owner):
            gg = torch.Generator().manual_seed(7 + owner)
            idx = torch.randint(0, R, (T_loc * B_glob * L,), generator=gg)
            off = torch.arange(T_loc * B_glob + 1) * L
            return idx, off
        idx, off = request_for(rank)
        reqs = split_request_by_group(idx, off, T_loc, groups, B_glob)
        assert [r[1].numel() for r in reqs] == [Tg * B_glob + 1] * groups and int(reqs[1][1][0]) == 0

        def stub_lookup(gi, ig, og, out):  # TEST STUB for the HIP kernel: [B_glob, Tg*D]
            for t in range(Tg):
                s, e = int(og[t * B_glob]), int(og[(t + 1) * B_glob])
                out[:, t * D:(t + 1) * D] = torch.nn.functional.embedding_bag(
                    ig[s:e], tables[gi * Tg + t], og[t * B_glob:(t + 1) * B_glob] - s, mode="sum")

        pipe = LookupAllToAll(stub_lookup, world, B_local, [Tg * D] * groups, torch.device("cpu"))
        recv = pipe.step(reqs)
        assert pipe.bytes_per_rank() == world * B_local * T_loc * D * 4
        # check: recv[g][src, b, t*D:(t+1)*D] == pooled embedding of MY local sample b for src's table g*Tg+t
        for src in range(world):
            gs = torch.Generator().manual_seed(100 + src)
            src_tables = [torch.randn(R, D, generator=gs) for _ in range(T_loc)]
            s_idx, s_off = request_for(src)
            for gi in range(groups):
                for t in range(Tg):
                    tt = gi * Tg + t
                    for b in range(B_local):
                        bag = tt * B_glob + rank * B_local + b      # my rows of the global batch
                        rows = s_idx[int(s_off[bag]):int(s_off[bag + 1])]
                        exp = src_tables[tt][rows].sum(0)
                        assert torch.allclose(recv[gi][src, b, t * D:(t + 1) * D], exp, atol=1e-5), (src, gi, t, b)
        pipe.lookups_only(reqs)
        pipe.all_to_all_only()

        # two steps in flight (double-buffered): three steps with DIFFERENT requests; after flush() the tensors returned
        # by each of the last two steps hold that step's exchange, untouched by the steps issued after it
        def request_k(owner, k):
            gg = torch.Generator().manual_seed(1000 * k + 7 + owner)
            return torch.randint(0, R, (T_loc * B_glob * L,), generator=gg), torch.arange(T_loc * B_glob + 1) * L

        pipe2 = LookupAllToAll(stub_lookup, world, B_local, [Tg * D] * groups, torch.device("cpu"), depth=2)
        outs = []
        for k in range(3):
            ik, ok = request_k(rank, k)
            outs.append(pipe2.step(split_request_by_group(ik, ok, T_loc, groups, B_glob)))
        pipe2.flush()
        assert outs[0][0].data_ptr() == outs[2][0].data_ptr() != outs[1][0].data_ptr()     # slots alternate
        for k in (1, 2):
            for src in range(world):
                gs = torch.Generator().manual_seed(100 + src)
                src_tables = [torch.randn(R, D, generator=gs) for _ in range(T_loc)]
                s_idx, s_off = request_k(src, k)
                for gi in range(groups):
                    for t in range(Tg):
                        tt = gi * Tg + t
                        for b in range(B_local):
                            bag = tt * B_glob + rank * B_local + b
                            exp = src_tables[tt][s_idx[int(s_off[bag]):int(s_off[bag + 1])]].sum(0)
                            assert torch.allclose(outs[k][gi][src, b, t * D:(t + 1) * D], exp, atol=1e-5), (k, src, gi, t, b)
    finally:
        dist.destroy_process_group()


def trace_replay(rank, world, port, outdir, golden_dir, blocking):
    """replay the collective entries of tests/golden/basic_trace.json on 2 gloo ranks.  The recorded index exchange is
    rank 0's ([39,34] -> [39,44]); rank 1 gets the mirror image so the two ranks' splits agree."""
    import json

    from param_amd.comms.pt import commsTraceReplay

    _env(rank, world, port)
    trace = [e for e in json.load(open(os.path.join(golden_dir, "basic_trace.json"))) if "comms" in e]
    if rank == 1:
        for e in trace:
            if e.get("in_split") == [39, 34]:
                e["in_split"], e["out_split"], e["in_msg_size"], e["out_msg_size"] = [44, 40], [34, 40], 84, 74
    tdir = os.path.join(outdir, "traces")
    os.makedirs(tdir, exist_ok=True)
    json.dump(trace, open(os.path.join(tdir, f"{rank}.json"), "w"))
    argv = ["--trace-path", tdir, "--backend", "gloo", "--device", "cpu", "--master-ip", "127.0.0.1",
            "--master-port", str(port), "--num-replays", "2", "--do-warm-up", "--output-path", os.path.join(outdir, "perf"),
            "--z", "1" if blocking else "0"]
    if blocking:
        argv += ["--c", "1", "--reuse-tensors"]
    import contextlib
    import io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench = commsTraceReplay.main(argv)
    with open(os.path.join(outdir, f"summary{rank}.json"), "w") as f:
        json.dump({"collLat": {k: len(v) for k, v in bench.collLat.items()}, "stdout": buf.getvalue(),
                   "total_us": bench.totalTraceLatency}, f)


def trace_replay_one_reader(rank, world, port, outdir, golden_dir):
    """``--use-one-trace --disable-parallel-read --enable-profiler``: only rank 0 opens the trace file (rank 1's ``open`` refuses
    the path), both replay the same symmetric operations, each writes a profiler trace"""
    import builtins
    import contextlib
    import io
    import json

    from param_amd.comms.pt import commsTraceReplay

    _env(rank, world, port)
    trace = [e for e in json.load(open(os.path.join(golden_dir, "basic_trace.json")))
             if e.get("comms") in ("all_reduce", "barrier") or (e.get("comms") == "all_to_all" and not e.get("in_split"))]
    assert len(trace) >= 2
    tdir = os.path.join(outdir, "one")
    if rank == 0:
        os.makedirs(tdir, exist_ok=True)
        json.dump(trace, open(os.path.join(tdir, "0.json"), "w"))
    dist_file = os.path.join(tdir, "0.json")
    import time as _t
    while not os.path.exists(dist_file):
        _t.sleep(0.05)
    real_open = builtins.open

    def guarded(path, *a, **k):
        if rank != 0 and os.path.abspath(str(path)) == os.path.abspath(dist_file):
            raise AssertionError("rank 1 opened the trace file although --disable-parallel-read was given")
        return real_open(path, *a, **k)

    argv = ["--trace-path", tdir, "--backend", "gloo", "--device", "cpu", "--master-ip", "127.0.0.1", "--master-port", str(port),
            "--num-replays", "3", "--use-one-trace", "--disable-parallel-read", "--enable-profiler", "--profiler-num-replays-start", "1",
            "--profiler-num-replays", "5", "--output-path", os.path.join(outdir, "perf1"), "--z", "1"]
    builtins.open = guarded
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            bench = commsTraceReplay.main(argv)
    finally:
        builtins.open = real_open
    with open(os.path.join(outdir, f"one{rank}.json"), "w") as f:
        json.dump({"collLat": {k: len(v) for k, v in bench.collLat.items()}, "n": len(bench.comms_trace)}, f)
    # the flag without --use-one-trace is refused
    with pytest_raises(ValueError, "--disable-parallel-read is valid only when --use-one-trace is used."):
        b2 = commsTraceReplay.commsTraceReplayBench()
        import argparse
        b2.checkArgs(b2.readArgs(argparse.ArgumentParser(), ["--trace-path", tdir, "--disable-parallel-read", "--device", "cpu",
                                                             "--backend", "gloo"]))


def plugin_table_collectives(rank, world, port):
    """The rest of the reference ABC's collective table (all_gather ... scatter, point-to-point, the pair-mode twins and
    the reference-driver call forms ``sayHello()`` with no arguments / list work handles in ``waitObj``): data checks."""
    import contextlib
    import io

    from param_amd.comms.pt.pytorch_backend_utils import collectiveArgsHolder

    _env(rank, world, port)
    bf = _backend(rank, world, port)
    try:
        ca = collectiveArgsHolder()
        ca.world_size, ca.global_rank, ca.group, ca.device = world, rank, bf.get_default_group(), bf.get_device()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bf.sayHello()                                # the reference drivers pass nothing (comms.py:1533)
            bf.sayHello(rank, rank, world, "127.0.0.1")  # the ABC's declared form still works
        if rank == 0:
            assert buf.getvalue().count("Hello from Rank 1: [Rank   1]") == 2, buf.getvalue()
        for name in ("all_gather", "all_gather_base", "reduce_scatter", "reduce_scatter_base", "broadcast", "gather", "scatter",
                     "all_to_all", "all_to_allv", "all_to_all_single", "all_reduce", "reduce", "barrier", "wait", "send", "recv"):
            assert name in bf.collectiveFunc, name
        me = torch.arange(3.0) + 10 * rank
        ca.asyncOp = False
        # all_gather (what comms.py:951 / dlrm.py:1220 use for their reports)
        ca.ipTensor, ca.opTensor = me.clone(), [torch.empty(3) for _ in range(world)]
        ca.collective = "all_gather"
        bf.collectiveFunc["all_gather"](ca)
        assert [t.tolist() for t in ca.opTensor] == [[0.0, 1.0, 2.0], [10.0, 11.0, 12.0]]
        assert bf.get_mem_size(ca) == 4 * 3 * world
        ca.opTensor = torch.empty(3 * world)
        bf.all_gather_base(ca)
        assert ca.opTensor.tolist() == [0.0, 1.0, 2.0, 10.0, 11.0, 12.0]
        # reduce_scatter (list) / reduce_scatter_base: get_mem_size counts the INPUT (reference :866-877)
        ca.op = bf.get_reduce_op("sum")
        ca.ipTensor, ca.opTensor = [torch.ones(2) * (rank + 1 + j) for j in range(world)], torch.empty(2)
        ca.collective = "reduce_scatter"
        bf.reduce_scatter(ca)
        assert ca.opTensor.tolist() == [[3.0, 3.0], [5.0, 5.0]][rank] and bf.get_mem_size(ca) == 16
        ca.ipTensor, ca.opTensor = torch.arange(4.0) * (rank + 1), torch.empty(2)
        ca.collective = "reduce_scatter_base"
        bf.reduce_scatter_base(ca)
        assert ca.opTensor.tolist() == [[0.0, 3.0], [6.0, 9.0]][rank] and bf.get_mem_size(ca) == 16
        # broadcast / gather / scatter
        ca.collective, ca.srcOrDst = "broadcast", 1
        ca.opTensor = me.clone()
        bf.broadcast(ca)
        assert ca.opTensor.tolist() == [10.0, 11.0, 12.0]
        ca.collective, ca.srcOrDst = "gather", 0
        ca.ipTensor, ca.opTensor = me.clone(), [torch.empty(3) for _ in range(world)]
        bf.gather(ca)
        if rank == 0:
            assert ca.opTensor[1].tolist() == [10.0, 11.0, 12.0]
        ca.collective = "scatter"
        ca.ipTensor, ca.opTensor = [torch.full((2,), float(j)) for j in range(world)], torch.empty(2)
        bf.scatter(ca)
        assert ca.opTensor.tolist() == [float(rank)] * 2
        # point to point, blocking and batched
        ca.ipTensor, ca.opTensor = me.clone(), torch.empty(3)
        ca.dst_rank = ca.src_rank = 1 - rank
        if rank == 0:
            bf.send(ca)
            bf.recv(ca)
        else:
            bf.recv(ca)
            bf.send(ca)
        assert ca.opTensor.tolist() == (torch.arange(3.0) + 10 * (1 - rank)).tolist()
        ca.p2pOps, ca.opTensor = [], torch.empty(3)
        ca.collective = "isend"
        bf.P2POp(ca)
        ca.collective = "irecv"
        bf.P2POp(ca)
        bf.batch_isend_irecv(ca)
        assert len(ca.waitObj) == 2 and ca.p2pOps == []
        bf.complete_accel_ops(ca)
        assert ca.opTensor.tolist() == (torch.arange(3.0) + 10 * (1 - rank)).tolist() and ca.waitObj == []
        # pair mode (comms.py --pair): the twins' tensors are used, get_mem_size reads the pair's output
        ca.collective = "all_to_allv"
        ca.ipTensor_pair, ca.opTensor_pair = [torch.arange(4.0) + 100 * rank], [torch.empty(4)]
        ca.ipTensor_split_pair, ca.opTensor_split_pair = [[2, 2]], [[2, 2]]
        bf.all_to_allv(ca, pair=True, pairIdx=0)
        assert ca.opTensor_pair[0].tolist() == [[0.0, 1.0, 100.0, 101.0], [2.0, 3.0, 102.0, 103.0]][rank]
        assert bf.get_mem_size(ca, pair=True, pairIdx=0) == 16
        # a list of work handles in waitObj / waitObjIds (what a pipelined call returns) is waited element-wise
        ca.asyncOp = True
        ca.ipTensor, ca.opTensor = torch.ones(2), torch.empty(2)
        ca.ipTensor_split = ca.opTensor_split = [1, 1]
        w1 = bf.all_to_allv(ca, retFlag=True)
        ca.waitObj.clear()
        ca.waitObj.append([w1])
        bf.wait(ca)
        assert ca.waitObj == []
        bf.sync_barrier(ca)
        # get_new_pg (trace replay's group creation, pytorch_dist_backend.py:1132-1138) and the store hand-off used by
        # --disable-parallel-read
        ca.group = bf.get_new_pg(list(range(world)), "gloo")
        ca.asyncOp, ca.ipTensor = False, torch.full((3,), float(rank + 1))
        ca.opTensor = ca.ipTensor
        bf.all_reduce(ca)
        assert ca.ipTensor.tolist() == [float(sum(range(1, world + 1)))] * 3
        if rank == 0:
            bf.store_set("k", "v1")
        assert bf.store_get("k") == b"v1"
    finally:
        bf.shutdown()


def comms_sweep_plugin_fixture(rank, world, port, outdir, argv_json):
    """this build's comms.py with the argument list of the reference-driver run recorded in ref_plugin_rows.json"""
    import contextlib
    import io
    import json

    from param_amd.comms.pt import comms

    _env(rank, world, port)
    argv = json.loads(argv_json)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        comms.main(["--master-ip", "127.0.0.1", "--master-port", str(port)] + argv)
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(buf.getvalue())


def comms_surface_case(rank, world, port, outdir, case_json):
    """this build's comms.py with one argument list of tests/golden/comms_surface.json (what the REFERENCE printed for it)"""
    import contextlib
    import io
    import json

    from param_amd.comms.pt import comms

    _env(rank, world, port)
    case = json.loads(case_json)
    argv = ["--master-ip", "127.0.0.1", "--master-port", str(port), "--n", "3", "--w", "1", "--backend", "gloo", "--device", "cpu"]
    argv += case["argv"] + (case["per_rank"][rank] if "per_rank" in case else [])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = comms.main(argv)
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"results": res, "stdout": buf.getvalue()}, f)


def sharded_exchange(rank, world, port, n_tables=3):
    """ShardedEmbeddingExchange on an UNEVEN table split (3 tables of mixed dims over 2 ranks -> [2, 1]; 26 mixed tables over
    4 / 8 ranks -> the reference's [7, 7, 6, 6] / [4, 4, 3, 3, 3, 3, 3, 3], dlrm.py:390-398): forward receive blocks, the
    exchange self-check, the gradient's way back (splits swapped) and the pipelined step's bookkeeping (3 batches in
    flight), against a single-process restatement.  Lookup / backward are torch stand-ins (the HIP kernels need a GPU)."""
    from param_amd.comms.pt.pipeline import ShardedEmbeddingExchange, table_split

    _env(rank, world, port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert table_split(26, 8) == [4, 4, 3, 3, 3, 3, 3, 3] and table_split(64, 8) == [8] * 8 and table_split(3, 2) == [2, 1]
        assert table_split(26, 4) == [7, 7, 6, 6]
        if n_tables == 3:
            rows, dims, pools, B_local = [40, 50, 60], [8, 4, 12], [3, 1, 5], 3
        else:   # Criteo-like: mixed dims and pooling factors (incl. tiny tables), a small per-rank batch
            rows = [[40, 7, 3, 90, 11][t % 5] + t for t in range(n_tables)]
            dims = [[8, 4, 12, 4][t % 4] for t in range(n_tables)]
            pools = [[3, 1, 5, 2, 1, 7][t % 6] for t in range(n_tables)]
            B_local = 2
        split = table_split(len(rows), world)
        if n_tables == 26 and world == 8:
            assert split == [4, 4, 3, 3, 3, 3, 3, 3]
        first = [sum(split[:r]) for r in range(world)]
        mine = list(range(first[rank], first[rank] + split[rank]))
        widths = [sum(dims[first[r]:first[r] + split[r]]) for r in range(world)]
        B_glob = world * B_local
        tables = {t: torch.randn(rows[t], dims[t], generator=torch.Generator().manual_seed(500 + t)) for t in range(len(rows))}

        def request(owner, k):   # the GLOBAL batch for owner's tables at step k: every rank can regenerate it
            gg = torch.Generator().manual_seed(1000 * k + owner)
            own = list(range(first[owner], first[owner] + split[owner]))
            idx = torch.cat([torch.randint(0, rows[t], (B_glob * pools[t],), generator=gg) for t in own])
            lens = torch.cat([torch.full((B_glob,), pools[t], dtype=torch.int64) for t in own])
            off = torch.zeros(len(own) * B_glob + 1, dtype=torch.int64)
            off[1:] = torch.cumsum(lens, 0)
            return idx, off

        def pooled_of(owner, k):  # [B_glob, widths[owner]]
            idx, off = request(owner, k)
            own = list(range(first[owner], first[owner] + split[owner]))
            cols = []
            for i, t in enumerate(own):
                s, e = int(off[i * B_glob]), int(off[(i + 1) * B_glob])
                cols.append(torch.nn.functional.embedding_bag(idx[s:e], tables[t], off[i * B_glob:(i + 1) * B_glob] - s, mode="sum"))
            return torch.cat(cols, dim=1)

        acc = {t: torch.zeros(rows[t], dims[t]) for t in mine}     # what the backward stand-in accumulates
        applied = []

        def lookup(indices, offsets, out):
            col = 0
            for i, t in enumerate(mine):
                s, e = int(offsets[i * B_glob]), int(offsets[(i + 1) * B_glob])
                out[:, col:col + dims[t]] = torch.nn.functional.embedding_bag(
                    indices[s:e], tables[t], offsets[i * B_glob:(i + 1) * B_glob] - s, mode="sum")
                col += dims[t]

        def backward(grad, indices, offsets):
            applied.append(int(indices.sum()))
            col = 0
            for i, t in enumerate(mine):
                for b in range(B_glob):
                    for j in range(int(offsets[i * B_glob + b]), int(offsets[i * B_glob + b + 1])):
                        acc[t][indices[j]] += grad[b, col:col + dims[t]]
                col += dims[t]

        def make_grad(recv, grad_in):      # the "dense part": a different scale per destination rank, so routing errors show
            grad_in.copy_(recv * float(rank + 2))

        ex = ShardedEmbeddingExchange(lookup, backward, world, rank, B_local, widths, torch.device("cpu"), make_grad=make_grad)
        assert ex.fwd_recv_splits == [B_local * w for w in widths] and ex.bytes_per_rank() == B_local * sum(widths) * 4
        steps = 5
        for k in range(steps):
            ex.step(*request(rank, k))
            if k >= 2:
                assert len(applied) == k - 1                       # backward(k-2) ran inside step k
            if k == 1:                                             # batch 0's exchange is complete now: check the blocks
                for src in range(world):
                    exp = pooled_of(src, 0)[rank * B_local:(rank + 1) * B_local]
                    assert torch.allclose(ex.recv_block(0, src), exp, atol=1e-6), src
                # the self-check bench.py runs after its timed region: every peer's first table recomputed locally
                d0 = lambda src: dims[first[src]]   # noqa: E731
                chk = ex.selfcheck(0, lambda src: pooled_of(src, 0)[rank * B_local:(rank + 1) * B_local, :d0(src)], exact=False)
                assert chk["a2a_selfcheck"] == "ok" and chk["peers_checked"] == world and chk["ranks"] == world, chk
                # a payload that is NOT what the peer sent is reported by every rank (rank 1 plants the fault)
                if rank == 1:
                    ex.recv[0][0] += 1.0
                bad = ex.selfcheck(0, lambda src: pooled_of(src, 0)[rank * B_local:(rank + 1) * B_local, :d0(src)], exact=False)
                assert bad["a2a_selfcheck"] == "MISMATCH" and bad["this_rank_ok"] == (rank != 1), bad
                if rank == 1:
                    ex.recv[0][0] -= 1.0
        ex.drain()
        assert applied == [int(request(rank, k)[0].sum()) for k in range(steps)]      # every batch once, in order
        # expected accumulators: grad(owner=me)[b_glob rows of rank j] = (j + 2) * pooled_me[those rows]
        exp_acc = {t: torch.zeros(rows[t], dims[t]) for t in mine}
        for k in range(steps):
            idx, off = request(rank, k)
            g = pooled_of(rank, k).clone()
            for j in range(world):
                g[j * B_local:(j + 1) * B_local] *= float(j + 2)
            col = 0
            for i, t in enumerate(mine):
                for b in range(B_glob):
                    for jx in range(int(off[i * B_glob + b]), int(off[i * B_glob + b + 1])):
                        exp_acc[t][idx[jx]] += g[b, col:col + dims[t]]
                col += dims[t]
        for t in mine:
            assert torch.allclose(acc[t], exp_acc[t], atol=1e-4), t
        # the un-overlapped form gives the same arithmetic for one more batch
        before = {t: acc[t].clone() for t in mine}
        ex.step_serial(*request(rank, 99))
        assert len(applied) == steps + 1 and any(not torch.equal(before[t], acc[t]) for t in mine)
        # without a dense stand-in the received embeddings themselves travel back (no extra buffer)
        ex2 = ShardedEmbeddingExchange(lookup, backward, world, rank, B_local, widths, torch.device("cpu"))
        assert ex2.grad_in is ex2.recv
        ex2.step_serial(*request(rank, 7))
        assert torch.allclose(ex2.grad[0], pooled_of(rank, 7), atol=1e-6)    # went to the peers and came back unchanged
    finally:
        dist.destroy_process_group()


def sharded_exchange_blocked(rank, world, port):
    """ShardedEmbeddingExchange with the BLOCKED send layout ([W][T_loc][B_local][D], ABI v6) against the same exchange in the
    default [W * B_local, T_loc * D] layout: 3 tables per rank, one dim, a power-of-two per-rank batch.  Torch stand-ins write / read
    each layout; what every rank RECEIVES (per source, per table), the self-check and what the backward accumulates must agree."""
    from param_amd.comms.pt.pipeline import ShardedEmbeddingExchange

    _env(rank, world, port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T_loc, D, Bl, L, R = 3, 4, 4, 2, 30
        B_glob = world * Bl
        widths = [T_loc * D] * world
        tables = {(r, t): torch.randn(R, D, generator=torch.Generator().manual_seed(50 * r + t)) for r in range(world) for t in range(T_loc)}

        def request(owner, k):
            gg = torch.Generator().manual_seed(100 * k + owner)
            idx = torch.randint(0, R, (T_loc * B_glob * L,), generator=gg)
            off = torch.arange(T_loc * B_glob + 1, dtype=torch.int64) * L
            return idx, off

        def pooled_tbd(owner, k):      # [T_loc, B_glob, D]
            idx, off = request(owner, k)
            return torch.stack([torch.nn.functional.embedding_bag(idx[t * B_glob * L:(t + 1) * B_glob * L], tables[(owner, t)],
                                                                  off[:B_glob], mode="sum") for t in range(T_loc)])

        results = {}
        for layout in ("bd", "blocked"):
            acc = {t: torch.zeros(R, D) for t in range(T_loc)}

            def lookup(indices, offsets, out, layout=layout):
                p = pooled_tbd(rank, lookup.k)
                if layout == "bd":
                    out.copy_(p.permute(1, 0, 2).reshape(B_glob, T_loc * D))
                else:
                    out.view(world, T_loc, Bl, D).copy_(p.view(T_loc, world, Bl, D).permute(1, 0, 2, 3))

            def backward(grad, indices, offsets, layout=layout, acc=acc):
                g = (grad.view(B_glob, T_loc, D).permute(1, 0, 2) if layout == "bd"
                     else grad.view(world, T_loc, Bl, D).permute(1, 0, 2, 3).reshape(T_loc, B_glob, D))
                for t in range(T_loc):
                    for b in range(B_glob):
                        for j in range(L):
                            acc[t][indices[(t * B_glob + b) * L + j]] += g[t, b]

            def make_grad(recv, grad_in):
                grad_in.copy_(recv * float(rank + 2))

            ex = ShardedEmbeddingExchange(lookup, backward, world, rank, Bl, widths, torch.device("cpu"), make_grad=make_grad,
                                          layout=layout, dim=D)
            recvd = []
            for k in range(3):
                lookup.k = k
                ex.step_serial(*request(rank, k))
                blocks = []
                for src in range(world):
                    blk = ex.recv_block(0, src)
                    blocks.append(blk.clone() if layout == "blocked" else blk.view(Bl, T_loc, D).permute(1, 0, 2).clone())    # -> [T_src, Bl, D]
                    exp = pooled_tbd(src, k)[:, rank * Bl:(rank + 1) * Bl]
                    assert torch.allclose(blocks[-1], exp, atol=1e-6), (layout, k, src)
                recvd.append(torch.stack(blocks))
                chk = ex.selfcheck(0, lambda src: pooled_tbd(src, k)[0, rank * Bl:(rank + 1) * Bl], exact=False)
                assert chk["a2a_selfcheck"] == "ok" and chk["peers_checked"] == world, (layout, chk)
            results[layout] = (torch.stack(recvd), {t: a.clone() for t, a in acc.items()})
        assert torch.equal(results["bd"][0], results["blocked"][0])
        for t in range(T_loc):
            assert torch.allclose(results["bd"][1][t], results["blocked"][1][t], atol=1e-5), t
            assert results["bd"][1][t].abs().sum() > 0
        # the blocked layout needs one dim dividing every width and fp32 payloads
        try:
            ShardedEmbeddingExchange(lambda *a: None, lambda *a: None, world, rank, Bl, [10] * world, torch.device("cpu"), layout="blocked", dim=4)
            raise AssertionError("a width that is not a whole number of tables was accepted")
        except ValueError:
            pass
    finally:
        dist.destroy_process_group()


def quantized_collectives(rank, world, port, outdir):
    """``--bitwidth < 32`` on 2 gloo ranks with host tensors: the quantised all_to_allv (uneven per-peer row counts) and the
    list-form all_to_all for every bit width against the numpy oracle of the row formats, the downcast all_reduce, the
    threshold, pair mode untouched, and the sweep driver's -QUANT report."""
    import contextlib
    import io
    import json

    import numpy as np

    from oracle import rowquant as orq
    from param_amd.comms.pt import comms, comms_utils
    from param_amd.comms.pt.pytorch_backend_utils import collectiveArgsHolder

    _env(rank, world, port)
    bf = _backend(rank, world, port)
    try:
        dim = 32
        rows_to = [[2, 3], [1, 4]]                        # rows_to[src][dst]
        chunk = lambda src, dst: torch.randn(rows_to[src][dst], dim, generator=torch.Generator().manual_seed(10 * src + dst)) * (3 + src)
        for bits in (16, 8, 4, 2):
            ca = collectiveArgsHolder()
            ca.group, ca.asyncOp, ca.world_size, ca.global_rank = bf.get_default_group(), False, world, rank
            comms_utils.initQuantCommCtx(ca, types.SimpleNamespace(bitwidth=bits, quant_a2a_embedding_dim=dim))
            ca.ipTensor = torch.cat([chunk(rank, d) for d in range(world)]).reshape(-1)
            ca.ipTensor_split = [rows_to[rank][d] * dim for d in range(world)]
            ca.opTensor_split = [rows_to[s][rank] * dim for s in range(world)]
            ca.opTensor = torch.full((sum(ca.opTensor_split),), -1.0)
            before = ca.ipTensor.clone()
            bf.all_to_allv(ca)
            want = np.concatenate([orq.dequantize_rows(orq.quantize_rows(chunk(s, rank).numpy(), bits), dim, bits)
                                   for s in range(world)]).reshape(-1)
            assert np.array_equal(ca.opTensor.numpy(), want), bits
            assert torch.equal(ca.ipTensor, before) and not ca.waitObj
            assert ca.quant_time.getTimeUS() > 0 and ca.dequant_time.getTimeUS() > 0
            # a NON-CONTIGUOUS output tensor (every other element of a wider buffer) receives the restored values too
            wide = torch.full((2 * sum(ca.opTensor_split),), -1.0)
            ca.opTensor = wide[::2]
            assert not ca.opTensor.is_contiguous()
            bf.all_to_allv(ca)
            assert np.array_equal(wide[::2].numpy(), want) and bool((wide[1::2] == -1.0).all()), bits
            ca.opTensor = torch.full((sum(ca.opTensor_split),), -1.0)
            # list form (equal chunks), same formats
            ca.ipTensor = [chunk(rank, 0)[:1].reshape(-1) + d for d in range(world)]
            ca.opTensor = [torch.zeros(dim) for _ in range(world)]
            bf.all_to_all(ca)
            for s in range(world):
                src_chunk = (torch.randn(rows_to[s][0], dim, generator=torch.Generator().manual_seed(10 * s)) * (3 + s))[:1] + rank
                assert np.array_equal(ca.opTensor[s].numpy(),
                                      orq.dequantize_rows(orq.quantize_rows(src_chunk.numpy(), bits), dim, bits).reshape(-1)), (bits, s)
            # below the threshold: the plain exchange, bit-exact payload
            ca.quant_threshold = 1 << 30
            ca.ipTensor = torch.cat([chunk(rank, d) for d in range(world)]).reshape(-1)
            ca.opTensor = torch.zeros(sum(ca.opTensor_split))
            bf.all_to_allv(ca)
            assert torch.equal(ca.opTensor, torch.cat([chunk(s, rank) for s in range(world)]).reshape(-1))
            ca.quant_threshold = 0
            # chunks that are not whole rows are refused, loudly
            ca.ipTensor_split = [dim + 1, ca.ipTensor.numel() - dim - 1]
            with pytest_raises(ValueError, "whole number of rows"):
                bf.all_to_allv(ca)
            # all_to_all_single: the reference warns and does nothing under quantisation
            ca.opTensor.fill_(7.0)
            assert bf.all_to_all_single(ca) is None and bool((ca.opTensor == 7.0).all())
        # all_reduce on a downcast copy: result handed back, ipTensor untouched (reference semantics)
        ca = collectiveArgsHolder()
        ca.group, ca.asyncOp, ca.world_size, ca.op = bf.get_default_group(), False, world, None
        comms_utils.initQuantCommCtx(ca, types.SimpleNamespace(bitwidth=16, quant_a2a_embedding_dim=dim))
        x = torch.arange(10, dtype=torch.float32) * 0.1 + rank
        ca.ipTensor = x.clone()
        got = bf.all_reduce(ca, retFlag=True)
        want = sum((torch.arange(10, dtype=torch.float32) * 0.1 + r).to(torch.float16) for r in range(world)).to(torch.float32)
        assert torch.equal(got, want) and torch.equal(ca.ipTensor, x)
        ca.asyncOp = True
        fut = bf.all_reduce(ca, retFlag=True)
        assert ca.waitObj == [fut]
        bf.complete_accel_ops(ca)
        assert torch.equal(fut.value(), want)
        ca.ipTensor = torch.arange(4, dtype=torch.int32)                   # not float32: the plain collective
        ca.asyncOp = False
        bf.all_reduce(ca)
        assert ca.ipTensor.tolist() == [0, 2, 4, 6]
    finally:
        bf.shutdown()
    # the sweep driver
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = comms.main(["--master-ip", "127.0.0.1", "--master-port", str(port + 1 if port < 65000 else port - 1), "--b", "8",
                          "--e", "4096", "--f", "4", "--n", "3", "--w", "1", "--z", "1", "--c", "1", "--collective",
                          "all_to_allv,all_to_all", "--backend", "rccl_xgmi", "--device", "cpu", "--bitwidth", "8",
                          "--quant-a2a-embedding-dim", "64"])
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"results": res, "stdout": buf.getvalue()}, f)


class pytest_raises:
    """minimal ``pytest.raises`` for worker processes"""

    def __init__(self, exc, match):
        self.exc, self.match = exc, match

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, self.exc) and self.match in str(ev), (et, ev)
        return True


def sharded_exchange_quantized(rank, world, port):
    """ShardedEmbeddingExchange with quantised payloads (forward 8-bit rows, gradient fp16; then 4-bit / fp32) on an uneven
    split: the receive blocks hold restore(quantise(pooled)) row by row, the gradient arriving at the owner is
    restore(quantise(gradient sent)), byte splits follow the table split.  numpy stand-ins for the HIP quantisers."""
    import numpy as np

    from oracle import rowquant as orq
    from param_amd.comms.pt.pipeline import RowQuant, ShardedEmbeddingExchange, table_split

    _env(rank, world, port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, rows, pools, B_local = 8, [40, 50, 60], [3, 1, 5], 4
        split = table_split(len(rows), world)                      # [2, 1]
        first = [sum(split[:r]) for r in range(world)]
        widths = [split[r] * D for r in range(world)]
        B_glob = world * B_local
        tables = {t: torch.randn(rows[t], D, generator=torch.Generator().manual_seed(700 + t)) * (t + 1) for t in range(len(rows))}

        def request(owner, k):
            gg = torch.Generator().manual_seed(1000 * k + owner)
            own = list(range(first[owner], first[owner] + split[owner]))
            idx = torch.cat([torch.randint(0, rows[t], (B_glob * pools[t],), generator=gg) for t in own])
            lens = torch.cat([torch.full((B_glob,), pools[t], dtype=torch.int64) for t in own])
            off = torch.zeros(len(own) * B_glob + 1, dtype=torch.int64)
            off[1:] = torch.cumsum(lens, 0)
            return idx, off

        def pooled_of(owner, k):
            idx, off = request(owner, k)
            own = list(range(first[owner], first[owner] + split[owner]))
            cols = []
            for i, t in enumerate(own):
                s, e = int(off[i * B_glob]), int(off[(i + 1) * B_glob])
                cols.append(torch.nn.functional.embedding_bag(idx[s:e], tables[t], off[i * B_glob:(i + 1) * B_glob] - s, mode="sum"))
            return torch.cat(cols, dim=1)

        def lookup(indices, offsets, out):
            out.copy_(pooled_of(rank, lookup.k))

        grads_home = []

        def backward(grad, indices, offsets):
            grads_home.append(grad.clone())

        def quantize(src, bits, out):
            out.copy_(torch.from_numpy(orq.quantize_rows(src.reshape(-1, D).numpy(), bits)).reshape(-1))

        def dequantize(src, bits, out):
            out.view(-1).copy_(torch.from_numpy(orq.dequantize_rows(src.numpy().reshape(-1, orq.row_bytes(D, bits)), D, bits)).reshape(-1))

        rt = lambda x, bits: torch.from_numpy(orq.dequantize_rows(orq.quantize_rows(x.reshape(-1, D).numpy(), bits), D, bits)).reshape(x.shape)  # noqa: E731
        for fwd_bits, bwd_bits in ((8, 16), (4, 0), (0, 2)):
            calls = {"fused": 0}

            def lookup_q(indices, offsets, out_q):                  # a lookup that writes quantised rows itself
                calls["fused"] += 1
                quantize(pooled_of(rank, lookup.k), fwd_bits, out_q)

            q = RowQuant(D, fwd_bits, bwd_bits, quantize, dequantize, lookup_quantized=lookup_q if fwd_bits == 4 else None)
            ex = ShardedEmbeddingExchange(lookup, backward, world, rank, B_local, widths, torch.device("cpu"),
                                          make_grad=lambda recv, gin: gin.copy_(recv * float(rank + 2)), quant=q)
            fb, gb = q.row_bytes(fwd_bits), q.row_bytes(bwd_bits)
            if fwd_bits:
                assert ex.qf_send == [B_local * split[rank] * fb] * world and ex.qf_recv == [B_local * split[r] * fb for r in range(world)]
            if bwd_bits:
                assert ex.qb_recv == [B_local * split[rank] * gb] * world
            wf, wb = ex.wire_bytes_per_rank()
            assert wf == (B_local * sum(split) * fb if fwd_bits else ex.bytes_per_rank())
            assert wb == (B_glob * split[rank] * gb if bwd_bits else B_glob * widths[rank] * 4)
            grads_home.clear()
            steps = 4
            for k in range(steps):
                lookup.k = k
                ex.step(*request(rank, k))
                if k == 1:
                    for src in range(world):
                        exp = pooled_of(src, 0)[rank * B_local:(rank + 1) * B_local]
                        exp = rt(exp, fwd_bits) if fwd_bits else exp
                        assert torch.equal(ex.recv_block(0, src), exp), (fwd_bits, src)
            ex.drain()
            assert len(grads_home) == steps and calls["fused"] == (steps if fwd_bits == 4 else 0)
            for k in range(steps):                                   # gradient the owner receives for batch k
                mine = pooled_of(rank, k)
                exp = torch.empty_like(mine)
                for j in range(world):
                    blk = mine[j * B_local:(j + 1) * B_local]
                    blk = rt(blk, fwd_bits) if fwd_bits else blk     # what rank j received ...
                    g = blk * float(j + 2)                            # ... its "dense part" ...
                    exp[j * B_local:(j + 1) * B_local] = rt(g, bwd_bits) if bwd_bits else g     # ... and sent back
                assert torch.equal(grads_home[k], exp), (fwd_bits, bwd_bits, k)
            lookup.k = 9
            ex.step_serial(*request(rank, 9))
            assert len(grads_home) == steps + 1
        with pytest_raises(ValueError, "whole number"):
            ShardedEmbeddingExchange(lookup, backward, world, rank, B_local, [12, 8], torch.device("cpu"), quant=RowQuant(8, 8, 0, quantize, dequantize))
    finally:
        dist.destroy_process_group()
