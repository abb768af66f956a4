"""Comms-trace parser -- the "basic" trace format of reference ``train/comms/pt/commsTraceParser.py``
(``parseTrace`` ``:27-63``, ``_parseBasicTrace`` ``:66-151``): a JSON list whose entries are either a collective

    {"comms": "all_to_allv", "req": 0, "in_msg_size": 1024, "out_msg_size": 1024, "dtype": "float32",
     "in_split": [...], "out_split": [...], "world_size": 8, "pg_id": 0, "markers": ["## a2a ##"], "startTime_ns": 0}

or a compute kernel; of the two the reference replays, the one on this build's path is the embedding lookup

    {"compute": "emb_lookup", "direction": "forward", "emb_dim": 128, "num_embs": 10000000, "batch_size": 8192,
     "num_emb_tables": 8, "bag_size": 20, "count": 1}

Sizes are ELEMENT counts.  Collective names are normalised with ``paramToCommName``; entries that are neither are a
``ValueError`` like in the reference.  The ``et`` (PyTorch execution trace) and ``kineto`` formats need the
reference's ``et_replay`` package and are outside this build: asking for them raises.
"""
from __future__ import annotations

from . import comms_utils
from .comms_utils import commsArgs

VALID_TRACE_TYPES = ["basic"]
_P2P = ("send", "recv", "isend", "irecv")


def parseTrace(in_trace: list, trace_type: str, target_rank: int = 0, total_ranks: int = 1) -> list:
    if trace_type == "basic":
        return _parseBasicTrace(in_trace)
    if trace_type in ("et", "kineto"):
        raise ValueError(f"trace type {trace_type!r} needs the reference's et_replay tooling and is not part of this build; "
                         "convert the trace to the basic format")
    raise ValueError("Unrecognized trace format.")


def _parseBasicTrace(in_trace: list) -> list:
    out = []
    for cnt, cur in enumerate(in_trace):
        new = commsArgs()
        new.id = cnt
        new.markerStack = cur.get("markers")
        if "comms" in cur:
            _parseBasicTraceComms(cur, new)
        elif "compute" in cur:
            _parseBasicTraceCompute(cur, new)
        if new.comms is None and new.compute is None:
            raise ValueError("Trace file contains an element that is not a supported in PARAM! "
                             "Please format all elements as comms or compute for replay.")
        out.append(new)
    return out


def _parseBasicTraceComms(cur: dict, new: commsArgs) -> None:
    new.comms = comms_utils.paramToCommName(cur["comms"].lower())
    if new.markerStack is None:
        new.markerStack = [new.comms]
    new.req = cur.get("req")
    new.startTimeNs = cur.get("startTime_ns")
    new.worldSize = cur.get("world_size")
    new.root = cur.get("root")
    new.pgId = cur.get("pg_id")
    new.groupRanks = cur.get("global_ranks")
    if new.comms not in ("wait", "barrier", "init", "batch_isend_irecv"):
        new.inMsgSize = cur["in_msg_size"]
        new.outMsgSize = cur["out_msg_size"]
        new.dtype = cur["dtype"].lower()
    if new.comms == "all_to_allv":
        new.inSplit = cur["in_split"]
        new.outSplit = cur["out_split"]
    if new.comms in _P2P:
        new.src_rank = cur["src_rank"]
        new.dst_rank = cur["dst_rank"]


def _parseBasicTraceCompute(cur: dict, new: commsArgs) -> None:
    new.compute = cur["compute"].lower()
    if new.markerStack is None:
        new.markerStack = [new.compute]
    new.count = cur.get("count", 1)          # number of times the kernel is launched
    if new.compute == "emb_lookup":
        new.direction = cur.get("direction", "forward")
        new.emb_dim = cur.get("emb_dim")
        new.num_embs = cur.get("num_embs")
        new.batch_size = cur.get("batch_size")
        new.num_emb_tables_per_device = cur.get("num_emb_tables")
        new.num_emb_tables_batched = -1
        new.bag_size = cur.get("bag_size")
    elif new.compute == "gemm":
        raise ValueError("Trace file contains a gemm compute element: the GEMM kernel is outside this build "
                         "(SURVEY 2.2 X1); replay it with the reference")
    else:
        raise ValueError(f"Trace file contains {new.compute} compute element that is not supported in PARAM!")
