#!/usr/bin/env python3
"""Generates tests/golden/replay_report.json: the text the REFERENCE's ``commsTraceReplayBench.reportBenchTime``
(train/comms/pt/commsTraceReplay.py:311-445) prints for fixed statistics -- message-size tables per collective, the latency
tables of a replay, the batch-latency table -- in dry-run and in replay mode.  Needs /root/reference (build container only)."""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.makedirs("/tmp/pb", exist_ok=True)
if not os.path.exists("/tmp/pb/param_bench"):
    os.symlink("/root/reference", "/tmp/pb/param_bench")
sys.path.insert(0, "/tmp/pb")
from param_bench.train.comms.pt.commsTraceReplay import commsTraceReplayBench  # noqa: E402

STATS = {
    "trace_file": "/data/traces/rank0.json", "n_msgs": 7,
    "collInMsgBytes": {"all_to_allv": [1024, 4096, 4096], "all_reduce": [576, 32]},
    "collOutMsgBytes": {"all_to_allv": [2048, 4096, 8192], "all_reduce": [576, 32]},
    "collLat": {"all_to_allv": [120.5, 98.25, 101.0], "all_reduce": [33.0, 35.5], "barrier": []},
    "totalTraceLatency": 512.75, "totalCommsLatency": 388.25, "batchLat": [1.5, 2.25], "colls_per_batch": 2,
}


def run(dry):
    b = commsTraceReplayBench()
    b.comms_trace = [None] * STATS["n_msgs"]
    b.trace_file = STATS["trace_file"]
    b.is_dry_run = dry
    b.collInMsgBytes = {k: list(v) for k, v in STATS["collInMsgBytes"].items()}
    b.collOutMsgBytes = {k: list(v) for k, v in STATS["collOutMsgBytes"].items()}
    b.collInUniMsgBytes = {k: set(v) for k, v in STATS["collInMsgBytes"].items()}
    b.collOutUniMsgBytes = {k: set(v) for k, v in STATS["collOutMsgBytes"].items()}
    b.collLat = {k: list(v) for k, v in STATS["collLat"].items()}
    b.totalTraceLatency, b.totalCommsLatency = STATS["totalTraceLatency"], STATS["totalCommsLatency"]
    b.batchLat, b.colls_per_batch = list(STATS["batchLat"]), STATS["colls_per_batch"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        b.reportBenchTime()
    return buf.getvalue()


out = {"stats": STATS, "dry_run": run(True), "replay": run(False)}
json.dump(out, open(os.path.join(HERE, "replay_report.json"), "w"), indent=1)
print(out["replay"])
