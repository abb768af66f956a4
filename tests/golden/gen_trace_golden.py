#!/usr/bin/env python3
"""Generates tests/golden/basic_trace_parsed.json from tests/golden/basic_trace.json (a hand-written trace in the
reference's basic format) with the REFERENCE's code.  Needs /root/reference (build container only); outputs are data.

The reference's ``commsTraceParser`` module cannot be imported in this image (its ``et_replay`` import needs ``pydot``,
absent), and the reference's replay then falls back to its base parser ``extractCommsInfo``
(commsTraceReplay.py:1519-1527,1531-1573) -- that is what runs here: collectives only, dtype taken verbatim.  So the
fixture covers the COLLECTIVE entries of the trace (dtype lower-cased first, as ``_parseBasicTraceComms`` would) through
``extractCommsInfo`` and ``initTraceStat`` (dry run); the parse of ``compute`` entries has no runnable reference
here (parity unpinned, stated in the test)."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
work = tempfile.mkdtemp()
os.makedirs(os.path.join(work, "pb"))
os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
sys.path.insert(0, os.path.join(work, "pb"))
from param_bench.train.comms.pt.commsTraceReplay import commsTraceReplayBench, extractCommsInfo  # noqa: E402
from param_bench.train.comms.pt.comms_utils import paramToCommName  # noqa: E402

trace = [dict(e) for e in json.load(open(os.path.join(HERE, "basic_trace.json"))) if "comms" in e]
for e in trace:
    if "dtype" in e:
        e["dtype"] = e["dtype"].lower()
parsed = extractCommsInfo(trace)
ops = [{"id": c.id, "comms": c.comms, "markerStack": c.markerStack, "req": c.req, "inMsgSize": c.inMsgSize,
        "outMsgSize": c.outMsgSize, "dtype": c.dtype, "inSplit": c.inSplit, "outSplit": c.outSplit,
        "worldSize": c.worldSize, "root": c.root, "pgId": c.pgId, "startTimeNs": c.startTimeNs, "toDict": c.toDict()}
       for c in parsed]
b = commsTraceReplayBench()
b.comms_trace = parsed
b.is_dry_run = True
b.initTraceStat()
stat = {"num_msg": b.num_msg, "max_msg_cnt": b.max_msg_cnt, "collInMsgBytes": b.collInMsgBytes,
        "collOutMsgBytes": b.collOutMsgBytes, "collLat_keys": list(b.collLat), "comms_blocks": dict(b.comms_blocks)}
names = ["all_to_all", "AllToAllV", "alltoallbase", "ALL_TO_ALL_SINGLE", "allreduce", "All-Gather", "reduce_scatter_base",
         "wait", "barrier", "recvAnySource", "all_gather_base"]
json.dump({"ops": ops, "initTraceStat_dry_run": stat, "paramToCommName": {n: paramToCommName(n) for n in names}},
          open(os.path.join(HERE, "basic_trace_parsed.json"), "w"), indent=1)
print("ok", len(ops))
