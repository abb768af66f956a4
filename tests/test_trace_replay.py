"""Trace replay: parser and first-pass statistics against fixtures produced by the reference's own code
(tests/golden/gen_trace_golden.py), and the reference's unit tests for the replay driver
(train/comms/pt/tests/commsTraceReplay_tests.py) restated against this build's classes with a mock backend."""
import json
import os
import time

import pytest
import torch

from param_amd.comms.pt import commsTraceParser
from param_amd.comms.pt.comms_utils import commsArgs, paramToCommName
from param_amd.comms.pt.commsTraceReplay import commsTraceReplayBench, replayParamsHolder


class MockBackend:
    """collectives are no-ops; allocation is real (CPU); barriers take a measurable moment"""

    def __init__(self):
        self.calls = []
        self.mock_collective = lambda ca: None
        names = ["all_to_all", "all_to_allv", "all_to_all_single", "all_reduce", "reduce", "all_gather", "barrier", "wait"]
        self.collectiveFunc = {n: self._make(n) for n in names}
        self.computeFunc = {}

    def _make(self, name):
        def fn(ca, retFlag=False):
            self.calls.append(name)
            self.mock_collective(ca)
            return object() if retFlag else None
        return fn

    def alloc_ones(self, size, dev, dtype, scaleFactor=1.0):
        return torch.ones(size, dtype=dtype) * scaleFactor

    def alloc_random(self, size, dev, dtype, scaleFactor=1.0):
        return torch.ones(size, dtype=dtype)

    def sync_barrier(self, ca, desc=""):
        time.sleep(0.001)

    def complete_accel_ops(self, ca):
        ca.waitObj.clear()

    def get_default_group(self):
        return None

    def get_global_rank(self):
        return 0

    def get_world_size(self):
        return 1


def _params():
    p = replayParamsHolder()
    p.dcheck, p.device = 1, "cpu"
    return p


@pytest.fixture
def golden(golden_dir):
    return (json.load(open(os.path.join(golden_dir, "basic_trace.json"))),
            json.load(open(os.path.join(golden_dir, "basic_trace_parsed.json"))))


# ---------------------------------------------------------------- parser / statistics vs the reference's output
def test_param_to_comm_name_matches_reference(golden):
    for name, want in golden[1]["paramToCommName"].items():
        assert paramToCommName(name) == want, name


def test_basic_trace_collectives_parse_like_the_reference(golden):
    trace, gold = golden
    parsed = commsTraceParser.parseTrace(trace, "basic", 0, 2)
    assert [c.id for c in parsed] == list(range(len(trace)))
    comms = [c for c in parsed if c.comms is not None]
    assert len(comms) == len(gold["ops"])
    for c, g in zip(comms, gold["ops"]):
        for key in ("comms", "req", "inMsgSize", "outMsgSize", "dtype", "inSplit", "outSplit", "worldSize", "root", "pgId",
                    "startTimeNs"):
            assert getattr(c, key) == g[key], (g["id"], key)
        # the reference's base parser leaves markerStack unset where the full parser defaults it to [name]
        assert c.markerStack == (g["markerStack"] if g["markerStack"] is not None else [c.comms])
        assert c.toDict() == g["toDict"]


def test_basic_trace_compute_entries():
    """parse of ``compute`` entries: the reference's ``_parseBasicTraceCompute`` is not importable in the build image
    (needs pydot) -- restated from commsTraceParser.py:114-151, parity unpinned"""
    e = {"compute": "emb_lookup", "direction": "backward", "emb_dim": 128, "num_embs": 1000, "batch_size": 64,
         "num_emb_tables": 8, "bag_size": 20, "count": 3}
    (c,) = commsTraceParser.parseTrace([e], "basic")
    assert (c.compute, c.comms, c.count, c.markerStack) == ("emb_lookup", None, 3, ["emb_lookup"])
    assert c.toEmbLookupTuple() == ("backward", 128, 1000, 64, 8, 20) and c.num_emb_tables_batched == -1
    (d,) = commsTraceParser.parseTrace([{"compute": "emb_lookup", "emb_dim": 8, "num_embs": 10, "batch_size": 2,
                                         "num_emb_tables": 1, "bag_size": 1}], "basic")
    assert d.count == 1 and d.direction == "forward"
    with pytest.raises(ValueError):
        commsTraceParser.parseTrace([{"compute": "conv"}], "basic")
    with pytest.raises(ValueError):
        commsTraceParser.parseTrace([{"neither": 1}], "basic")
    with pytest.raises(ValueError):
        commsTraceParser.parseTrace([], "et")


def test_init_trace_stat_dry_run_matches_reference(golden):
    trace, gold = golden
    g = gold["initTraceStat_dry_run"]
    b = commsTraceReplayBench()
    b.comms_trace = [c for c in commsTraceParser.parseTrace(trace, "basic") if c.comms is not None]
    # the fixture comes from the reference's base parser, which does not default markerStack
    for c, ref in zip(b.comms_trace, gold["ops"]):
        c.markerStack = ref["markerStack"]
    b.is_dry_run = True
    b.initTraceStat()
    assert (b.num_msg, b.max_msg_cnt) == (g["num_msg"], g["max_msg_cnt"])
    assert b.collInMsgBytes == g["collInMsgBytes"] and b.collOutMsgBytes == g["collOutMsgBytes"]
    assert list(b.collLat) == g["collLat_keys"]
    assert dict(b.comms_blocks) == g["comms_blocks"]


# ---------------------------------------------------------------- the reference's replay unit tests, restated
def test_prep_comms_no_tensor():
    b = commsTraceReplayBench()
    b.backendFuncs = MockBackend()
    for name in ("wait", "barrier"):
        ip, op = b.prepComms(commsArgs(comms=name), None)
        assert len(ip) == 0 and len(op) == 0


def test_prep_comms_no_shrink():
    b = commsTraceReplayBench()
    b.backendFuncs = MockBackend()
    b.shrink = False
    b.collectiveArgs.world_size = 1
    ip, op = b.prepComms(commsArgs(comms="all_reduce", dtype="int", inMsgSize=1, outMsgSize=1), _params())
    assert len(ip) == 1 and len(op) == 1 and ip[0] == 1 and op[0] == 1


def test_prep_comms_shrink_alltoallv():
    b = commsTraceReplayBench()
    b.backendFuncs = MockBackend()
    b.shrink = True
    b.collectiveArgs.world_size = 1
    cur = commsArgs(comms="all_to_allv", dtype="int", inMsgSize=4, outMsgSize=4, inSplit=[1, 1, 1, 1],
                    outSplit=[1, 1, 1, 1], worldSize=4)
    ip, op = b.prepComms(cur, _params())
    assert len(ip) == 1 and len(op) == 1 and ip[0] == 1 and op[0] == 1
    assert b.collectiveArgs.ipTensor_split == [1] and b.collectiveArgs.opTensor_split == [1] and cur.worldSize == 1


def test_prep_comms_shrink_allgather():
    b = commsTraceReplayBench()
    b.backendFuncs = MockBackend()
    b.shrink = True
    b.collectiveArgs.world_size = 1
    ip, op = b.prepComms(commsArgs(comms="all_gather", dtype="int", inMsgSize=4, outMsgSize=4, worldSize=4), _params())
    assert len(ip) == 1 and len(op) == 1


def _three_op_trace():
    return [commsArgs(comms="test", inMsgSize=1, outMsgSize=1, dtype="int", markerStack=["test_stack"]),
            commsArgs(comms="all_gather", inMsgSize=2, outMsgSize=2, dtype="int"),
            commsArgs(comms="wait", markerStack=["test_stack"])]


@pytest.mark.parametrize("warmup", [True, False])
def test_replay_runs_and_skips_unknown_collectives(warmup):
    b = commsTraceReplayBench()
    b.backendFuncs = MockBackend()
    b.comms_trace = _three_op_trace()
    b.collectiveArgs.world_size = 1
    b.replayTrace(_params(), warmup)
    assert b.backendFuncs.calls == []          # max_msg_cnt 0 before initTraceStat(): nothing is replayed (reference :995)
    b.initTraceStat()                          # the run's order (runBench): statistics first -- "0 = no limit" becomes the length
    b.replayTrace(_params(), warmup)
    assert b.backendFuncs.calls == ["all_gather", "wait"]          # "test" is not a collective: warned and skipped
    assert len(b.traceWithPerf) == (0 if warmup else 3)


@pytest.mark.parametrize("blocking", [True, False])
def test_run_comms_latencies(blocking):
    b = commsTraceReplayBench()
    b.is_blocking = blocking
    b.backendFuncs = MockBackend()
    lat, glob = b.runComms("all_gather", commsArgs(req=0), "test_stack")
    assert lat is not None and glob is not None
    assert (lat != glob) if blocking else (lat == glob)            # blocking adds the trailing barrier
    assert ("0" in b.collectiveArgs.waitObjIds) == (not blocking)   # request id recorded for a later wait


def test_init_trace_stat():
    for dry in (True, False):
        b = commsTraceReplayBench()
        b.comms_trace = _three_op_trace()
        b.is_dry_run = dry
        b.initTraceStat()
        assert len(b.collInMsgBytes) == 2 and len(b.collOutMsgBytes) == 2
        assert sum(b.collInMsgBytes["all_gather"]) == 8 and sum(b.collOutMsgBytes["all_gather"]) == 8
        blocks = b.comms_blocks["test_stack"]
        if dry:
            assert blocks == [{"comms": "test", "in_msg_size": 1, "out_msg_size": 1}, {"comms": "wait"}]
        else:
            assert blocks == []


def test_init_bench_sets_replay_parameters():
    import argparse

    b = commsTraceReplayBench()
    args = b.readArgs(argparse.ArgumentParser(), ["--use-timestamp", "--max-msg-cnt", "1000", "--num-replays", "3",
                                                  "--output-ranks", "0", "--reuse-tensors", "--z", "1"])
    b.initBench(_params(), args)
    assert (b.use_timestamp, b.max_msg_cnt, b.shrink, b.do_warm_up, b.num_replays, b.reuse_tensors, b.outputRanks,
            b.is_blocking) == (True, 1000, False, False, 3, True, [0], True)


def test_rebalance_split_equal_policy():
    b = commsTraceReplayBench()
    b.collectiveArgs.device = "cpu"
    b.collectiveArgs.world_size = 2
    b.rebalance_policy = "equal"
    b.backendFuncs = MockBackend()
    # a second rank with inMsgSize 11: the mocked all_reduce yields 16
    b.backendFuncs.mock_collective = lambda ca: setattr(ca, "ipTensor", torch.tensor([16], dtype=torch.int))
    cur = commsArgs(comms="all_to_allv", inMsgSize=5, outMsgSize=3, inSplit=[3, 2], outSplit=[1, 2])
    b.rebalanceSplit(cur)
    assert (cur.inMsgSize, cur.outMsgSize, cur.inSplit, cur.outSplit) == (8, 8, [4, 4], [4, 4])


def test_rebalance_split_unsupported_policy():
    b = commsTraceReplayBench()
    b.rebalance_policy = "unsupported"
    cur = commsArgs(comms="all_to_allv", inMsgSize=5, outMsgSize=3, worldSize=2, inSplit=[3, 2], outSplit=[1, 2])
    b.rebalanceSplit(cur)
    assert (cur.inMsgSize, cur.outMsgSize, cur.inSplit, cur.outSplit) == (5, 3, [3, 2], [1, 2])


def test_compute_entry_fails_loudly_without_a_gpu(tmp_path):
    """an ``emb_lookup`` entry is served by the HIP kernels only: on a CPU device the replay raises, it does not fall back"""
    from param_amd.comms.pt import comms_utils

    class NoGpuBackend(MockBackend):
        def alloc_batched_embedding_tables(self, rows, dim, dev, dtype, layout="bd"):
            from param_amd import BatchedEmbeddingBagMI355
            return BatchedEmbeddingBagMI355(rows, dim, dtype=dtype, device=dev)

    b = commsTraceReplayBench()
    b.backendFuncs = NoGpuBackend()
    b.collectiveArgs.device = torch.device("cpu")
    (cur,) = commsTraceParser.parseTrace([{"compute": "emb_lookup", "emb_dim": 8, "num_embs": 10, "batch_size": 2,
                                           "num_emb_tables": 1, "bag_size": 1}], "basic")
    with pytest.raises(Exception):
        comms_utils.init_emb_lookup(b.collectiveArgs, cur, b.backendFuncs)


def test_param_profile_ranges_and_timer():
    """param_profile.paramProfile (reference param_profile.py:18-40): a named profiler range that advances a paramTimer; the
    replay labels every collective with the reference's range names, so a torch.profiler trace shows them"""
    import torch

    from param_amd.comms.pt.param_profile import paramProfile, paramTimer

    t = paramTimer()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        with paramProfile(timer=t, description="# PARAM unit range") as p:
            torch.ones(8).sum()
        with paramProfile(description="# untimed"):
            pass
        b = commsTraceReplayBench()
        b.is_blocking = True
        b.replayIter = 3
        b.backendFuncs = MockBackend()
        b.runComms("all_gather", commsArgs(req=0), "blockA")
    assert t.getTimeNS() == p.intervalNS > 0
    names = {e.name for e in prof.events()}
    assert {"# PARAM unit range", "# untimed", "# PARAM replay 3 pre-comm barrier # blockA", "# PARAM replay 3:blockA",
            "# PARAM replay 3 post-comm barrier # blockA"} <= names


def test_replay_report_text_equals_reference(capsys):
    """reportBenchTime: for the same statistics (message sizes per collective, replay latencies, batch latencies) the text equals
    what the REFERENCE's commsTraceReplayBench.reportBenchTime prints, in dry-run and in replay mode
    (tests/golden/replay_report.json, gen_replay_report.py ran the reference's method on a bare instance)"""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay_report.json")))
    st = gold["stats"]
    for dry, key in ((True, "dry_run"), (False, "replay")):
        b = commsTraceReplayBench()
        b.comms_trace = [None] * st["n_msgs"]
        b.trace_file, b.is_dry_run = st["trace_file"], dry
        b.collInMsgBytes = {k: list(v) for k, v in st["collInMsgBytes"].items()}
        b.collOutMsgBytes = {k: list(v) for k, v in st["collOutMsgBytes"].items()}
        b.collLat = {k: list(v) for k, v in st["collLat"].items()}
        b.compLat = {}
        b.totalTraceLatency, b.totalCommsLatency, b.totalCompsLatency = st["totalTraceLatency"], st["totalCommsLatency"], 0.0
        b.batchLat, b.colls_per_batch = list(st["batchLat"]), st["colls_per_batch"]
        capsys.readouterr()
        b.reportBenchTime()
        assert capsys.readouterr().out == gold[key], key


def test_example_trace_replays_on_the_host_with_stub_lookup(tmp_path, monkeypatch):
    """examples/trace_replay/0.json (the trace the GPU test replays: a2a's, all_reduces, emb_lookup forward / backward as compute
    entries) through commsTraceReplay.main on one gloo rank with the lookup kernel stubbed out -- the driver's bookkeeping around
    the compute entries (profiler ranges, reuse cache, per-kernel latency table) without the device"""
    import contextlib
    import io
    import json
    import os

    from param_amd.comms.pt import comms_utils, commsTraceReplay
    from param_amd.comms.pt.mi355_backend import MI355XBackend
    from tests.dist_workers import free_port

    def fake_init(ca, cur, bf):
        ca.direction, ca.emb_dim, ca.batch_size = cur.direction, cur.emb_dim, cur.batch_size
        ca.num_emb_ops, ca.num_emb_tables_batched, ca.embRequests, ca.emb, ca.LookupOut, ca.grad_output = 1, -1, [None], [None], None, None

    monkeypatch.setattr(comms_utils, "init_emb_lookup", fake_init)
    monkeypatch.setattr(MI355XBackend, "emb_lookup", lambda self, ca: None)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_SIZE"):
        monkeypatch.delenv(k, raising=False)
    tdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "trace_replay")
    for blocking in ("1", "0"):
        out = tmp_path / f"z{blocking}"
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            b = commsTraceReplay.main(["--trace-path", tdir, "--device", "cpu", "--backend", "gloo", "--master-ip", "127.0.0.1",
                                       "--master-port", str(free_port()), "--num-replays", "3", "--do-warm-up", "--reuse-tensors",
                                       "--z", blocking, "--output-path", str(out)])
        assert len(b.compLat["emb_lookup"]) == 6 and min(b.compLat["emb_lookup"]) > 0
        assert len(b.collLat["all_to_allv"]) == 9 and len(b.collLat["all_reduce"]) == 3
        assert len(b.embLookupReuse) == 2
        rec = json.load(open(out / "replayedCommsPerf.rank0.json"))
        fwd = [r for r in rec if r.get("compute") == "emb_lookup" and r["direction"] == "forward"]
        assert len(fwd) == 3 and fwd[0]["num_emb_tables"] == 8 and fwd[0]["latency_us"] > 0
        assert "Replayed 6 emb_lookup (compute)" in buf.getvalue()
