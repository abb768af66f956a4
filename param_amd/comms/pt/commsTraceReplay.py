"""commsTraceReplay.py -- replay a recorded sequence of collectives and ``emb_lookup`` compute kernels.

Own restatement of the replay driver of reference ``train/comms/pt/commsTraceReplay.py`` for the operations on this
build's path: the embedding lookup as the compute kernel (HIP) and the all-to-all family / all_reduce /
reduce / barrier / wait as collectives (RCCL through :class:`MI355XBackend`).

    torchrun --nproc-per-node 8 -m param_amd.comms.pt.commsTraceReplay --trace-path traces/ --trace-type basic \
        --backend rccl_xgmi --device rocm --num-replays 5 --do-warm-up --reuse-tensors --output-path out/

Kept from the reference (file:line of ``commsTraceReplay.py``): flags ``:153-276``; per-rank trace file
``<dir>/<rank>.json`` or one file for all ranks ``:1435-1483``; first pass over the trace for message-size statistics
``initTraceStat :448-507``; tensor preparation from the recorded ELEMENT counts with optional ``--auto-shrink`` to the
current world size ``prepComms :604-696``; ``--rebalance-policy equal`` for all_to_allv ``:509-542``; blocking replay =
barrier / collective / wait-all / barrier with ``global_latency = latency + trailing barrier`` and non-blocking replay
= post only, ``wait`` entries resolved through the recorded request ids ``runComms :756-833``; compute entries
launched ``count`` times on a separate HIP stream ``runCompute :723-754`` with ``--reuse-tensors`` caching the tables
and requests per entry shape ``prepComputeReplay :853-932``; per-operation records + latency tables
``recordCommReplay :934-972``, ``reportBenchTime :311-446``; ``replayedCommsPerf.rank<r>.json`` ``:43-86``.
Left out: remote (http / internal) trace stores, the kineto profiler hooks, process-group
creation from ``init`` entries (entries naming a ``pg_id`` other than the default group are skipped with a warning),
point-to-point ops, and the ``et`` / ``kineto`` trace formats (commsTraceParser.py).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import time

import numpy as np
import torch

from . import comms_utils, commsTraceParser
from .comms_utils import commsArgs, paramStreamGuard, paramTimer, paramToCommName
from .param_profile import paramProfile
from .mi355_backend import BACKEND_NAME, MI355XBackend, register
from .pytorch_backend_utils import collectiveArgsHolder, customized_backend

logger = logging.getLogger(__name__)

LOOP_TIMER_S = 0.02
VALID_TRACE_TYPES = commsTraceParser.VALID_TRACE_TYPES

dtypeMap = {
    "float": torch.float32, "float32": torch.float32, "float16": torch.half, "float64": torch.double,
    "double": torch.double, "int32": torch.int32, "int": torch.int32, "long": torch.long, "bfloat16": torch.bfloat16,
    "bool": torch.bool, "half": torch.half, "byte": torch.uint8, "uint8": torch.uint8, "int8": torch.int8,
    "short": torch.short, "char": torch.int8,
}


def writeCommDetails(commsTracePerf: list, rank: int, folder: str = "./") -> None:
    """one JSON list per rank with every replayed operation and its latency; ``folder == ""`` skips the output"""
    if len(folder) == 0:
        return
    os.makedirs(folder, exist_ok=True)
    comms_file = os.path.join(folder, f"replayedCommsPerf.rank{rank}.json")
    logger.info(f"[Rank {rank:3}] Writing comms details to {comms_file}")
    with open(comms_file, "w") as f:
        json.dump(commsTracePerf, f, indent=2)


class replayParamsHolder:
    """run parameters the replay hands to tensor preparation (subset of commsParamsHolderBase)"""

    def __init__(self, args=None):
        self.device = "cpu"
        self.backend = "gloo"
        self.dtype = torch.float32
        self.dcheck = 0
        self.blockingFlag = 1
        self.bitwidth = 32
        self.quant_a2a_embedding_dim = 32
        self.quant_threshold = 33554432
        self.size_from_trace = True
        if args is not None:
            self.device = "cuda" if args.device == "rocm" else args.device
            self.backend = args.backend
            self.dcheck = args.c
            self.blockingFlag = args.z
            self.bitwidth = getattr(args, "bitwidth", 32)
            self.quant_a2a_embedding_dim = getattr(args, "quant_a2a_embedding_dim", 32)
            self.quant_threshold = getattr(args, "quant_threshold", 33554432)


class _StatDict(dict):
    """``comms_blocks``: a missing block reads as an empty list"""

    def __missing__(self, key):
        self[key] = []
        return self[key]


class commsTraceReplayBench:
    def __init__(self):
        self.collectiveArgs = collectiveArgsHolder()
        self.backendFuncs = None
        self.dtypeMap = dtypeMap
        self.initVal = 1
        self.comms_trace = []
        self.trace_file = ""
        self.trace_type = "basic"
        self.use_one_trace = False
        self.is_dry_run = False
        self.shrink = False
        self.max_msg_cnt = 0          # 0 means no limit
        self.num_msg = 0
        self.is_blocking = False
        self.do_warm_up = False
        self.reuse_tensors = False
        self.allowList = ""
        self.out_path = ""
        self.outputRanks = None
        self.colls_per_batch = -1
        self.use_timestamp = False
        self.num_replays = 1
        self.rebalance_policy = ""
        self.replayIter = 0
        self.report = False
        self.world_size = 1

        self.collInMsgBytes = {}
        self.collInUniMsgBytes = {}
        self.collOutMsgBytes = {}
        self.collOutUniMsgBytes = {}
        self.collLat = {}
        self.compLat = {}
        self.comms_blocks = _StatDict()
        self.traceWithPerf = []
        self.batchLat = []
        self.totalCommsLatency = 0.0
        self.totalCompsLatency = 0.0
        self.totalTraceLatency = 0.0
        self.embLookupReuse = {}
        self.tensorReuse = {}

    # ------------------------------------------------------------------ args
    def readArgs(self, parser: argparse.ArgumentParser, argv=None):
        parser.add_argument("--master-ip", type=str, default="127.0.0.1")
        parser.add_argument("--master-port", type=str, default="29500")
        parser.add_argument("--backend", type=str, default=BACKEND_NAME, help="rccl_xgmi | nccl | gloo")
        parser.add_argument("--nw-stack", type=str, default="pytorch-dist")
        parser.add_argument("--device", type=str, default="rocm", choices=["cuda", "rocm", "cpu"])
        parser.add_argument("--z", "--blocking", type=int, default=0, dest="z", help="1: barrier + wait around every collective")
        parser.add_argument("--c", "--check", type=int, default=0, dest="c")
        parser.add_argument("--log", type=str, default="ERROR")
        parser.add_argument("--bitwidth", type=int, default=32, choices=[2, 4, 8, 16, 32], help="Quantization bitwidth")
        parser.add_argument("--quant-a2a-embedding-dim", type=int, default=32, choices=[32, 64, 128, 256])
        parser.add_argument("--quant-threshold", type=int, default=33554432,
                            help="quantise collectives of at least this many elements")
        parser.add_argument("--trace-path", type=str, default="./",
                            help="trace file, or a directory holding <rank>.json per rank")
        parser.add_argument("--trace-type", type=str, default="basic", help=f"supported: {VALID_TRACE_TYPES}")
        parser.add_argument("--use-one-trace", action="store_true", default=False, help="all ranks replay the same file")
        parser.add_argument("--dry-run", action="store_true", default=False, help="analyse the trace, replay nothing")
        parser.add_argument("--auto-shrink", action="store_true", default=False,
                            help="shrink message sizes recorded at a larger scale to the current world size")
        parser.add_argument("--max-msg-cnt", type=int, default=0, help="only replay the first N operations (0: all)")
        parser.add_argument("--do-warm-up", action="store_true", default=False, help="one untimed replay first")
        parser.add_argument("--reuse-tensors", action="store_true", default=False,
                            help="cache and reuse the tensors / embedding tables of each operation shape")
        parser.add_argument("--allow-ops", "--allow-list", type=str, default="all", dest="allow_ops")
        parser.add_argument("--output-path", type=str, default="", nargs="?", const="")
        parser.add_argument("--output-ranks", type=str, default="all")
        parser.add_argument("--colls-per-batch", type=int, default=-1)
        parser.add_argument("--use-timestamp", action="store_true", default=False)
        parser.add_argument("--rebalance-policy", type=str, default="")
        parser.add_argument("--num-replays", type=int, default=1)
        parser.add_argument("--disable-parallel-read", action="store_true", default=False,
                            help="with --use-one-trace: rank 0 reads the trace, the other ranks take it from the rendezvous store")
        parser.add_argument("--enable-profiler", action="store_true", default=False, help="torch.profiler over the replays")
        parser.add_argument("--profiler-num-replays-start", type=int, default=0,
                            help="replay iteration (after the warm-up) at which the profiler starts")
        parser.add_argument("--profiler-num-replays", type=int, default=10, help="replay iterations the profiler records")
        args, _ = parser.parse_known_args(argv)
        return args

    def checkArgs(self, args) -> None:
        if not os.path.isfile(args.trace_path) and not os.path.isdir(args.trace_path):
            raise ValueError(f"The specified trace path '{args.trace_path}' is neither a file nor a directory. "
                             "Please provide a valid path.")
        if getattr(args, "disable_parallel_read", False) and not args.use_one_trace:      # commsTraceReplay.py:300-304
            raise ValueError("--disable-parallel-read is valid only when --use-one-trace is used.")
        if args.trace_type not in VALID_TRACE_TYPES:
            raise ValueError(f"Trace type {args.trace_type} is not valid! Please specify one supported trace type from "
                             f"{VALID_TRACE_TYPES} by using --trace-type.")
        if args.device == "cpu" and args.backend in ("nccl", BACKEND_NAME):
            raise ValueError(f"backend {args.backend} does not support device cpu")

    def setTraceFile(self, args, comms_env_params=None) -> None:
        self.trace_file = args.trace_path
        self.trace_type = args.trace_type

    def initBench(self, commsParams, args) -> None:
        self.is_dry_run = args.dry_run
        self.shrink = args.auto_shrink
        self.max_msg_cnt = args.max_msg_cnt
        self.is_blocking = getattr(args, "z", 0) == 1
        self.do_warm_up = args.do_warm_up
        self.reuse_tensors = args.reuse_tensors
        self.allowList = getattr(args, "allow_ops", "all")
        if args.output_ranks == "all":
            n = self.backendFuncs.get_world_size() if self.backendFuncs is not None else 1
            self.outputRanks = list(range(n))
        else:
            self.outputRanks = comms_utils.parseRankList(args.output_ranks)
        self.out_path = getattr(args, "output_path", "")
        self.colls_per_batch = getattr(args, "colls_per_batch", -1)
        self.use_timestamp = args.use_timestamp
        self.rebalance_policy = getattr(args, "rebalance_policy", "").lower()
        self.num_replays = args.num_replays
        self.use_one_trace = args.use_one_trace
        self.disable_parallel_read = getattr(args, "disable_parallel_read", False)
        self.enable_profiler = getattr(args, "enable_profiler", False)
        self.profiler_num_replays_start = getattr(args, "profiler_num_replays_start", 0)
        self.profiler_num_replays = getattr(args, "profiler_num_replays", 10)

    # ------------------------------------------------------------------ statistics
    def initTraceStat(self) -> None:
        """first pass: message counts and sizes per collective, operations per marker block"""
        self.num_msg = len(self.comms_trace)
        self.max_msg_cnt = self.num_msg if self.max_msg_cnt == 0 else self.max_msg_cnt
        for cur in self.comms_trace[: self.max_msg_cnt]:
            if cur.compute is not None:
                self.compLat.setdefault(cur.compute, [])
                continue
            name = paramToCommName(cur.comms)
            if name not in self.collLat:
                self.collLat[name] = []
                if cur.inMsgSize is not None:
                    self.collInMsgBytes[name], self.collInUniMsgBytes[name] = [], set()
                    self.collOutMsgBytes[name], self.collOutUniMsgBytes[name] = [], set()
            if cur.inMsgSize is not None:
                es = torch.tensor([], dtype=self.dtypeMap[cur.dtype]).element_size()
                self.collInMsgBytes[name].append(cur.inMsgSize * es)
                self.collInUniMsgBytes[name].add(cur.inMsgSize * es)
                self.collOutMsgBytes[name].append(cur.outMsgSize * es)
                self.collOutUniMsgBytes[name].add(cur.outMsgSize * es)
            for block in (cur.markerStack or []):
                entries = self.comms_blocks[block]
                if self.is_dry_run:   # a replay fills the blocks later, with latencies
                    if name not in ("wait", "barrier"):
                        entries.append({"comms": name, "in_msg_size": cur.inMsgSize, "out_msg_size": cur.outMsgSize})
                    else:
                        entries.append({"comms": name})

    def reportBenchTime(self) -> None:
        size_hdr = f" {'Total (MB)':>10} {'Max.':>15} {'Min.':>10} {'Average':>13} {'p50':>13} {'p95':>13}"
        size_row = "{:>10.2f} {:15.2f} {:10.2f} {:15.2f} {:15.2f} {:15.2f}"
        lat_hdr = f" {'Total':>10} {'Max.':>10} {'Min.':>10} {'Average':>10} {'p50':>10} {'p95':>10}"
        lat_row = " {:10.2f} {:10.2f} {:10.2f} {:10.2f} {:10.2f} {:10.2f}"

        def six(a):
            return a.sum(), a.max(), a.min(), np.average(a), np.percentile(a, 50), np.percentile(a, 95)

        print(f"\n+++++ {len(self.comms_trace)} msgs recorded in {self.trace_file} +++++\n")
        for name, msgs in self.collInMsgBytes.items():
            print("-" * 50 + f"\n+ {len(msgs)} {name}\n" + "-" * 50)
            for title, arr in (("Input", np.array(msgs)), ("Output", np.array(self.collOutMsgBytes[name]))):
                s = six(arr)
                print(f"Size of {title} tensors (bytes)\n{size_hdr}")
                print(size_row.format(s[0] / 1024 / 1024, *s[1:]))
        if self.is_dry_run:
            return
        print("\n{} Performance of replayed comms {}".format("=" * 20, "=" * 20))
        print("{}\n Total latency (us) of comms in trace {}: \n{}".format("-" * 50, self.totalTraceLatency, "-" * 50))
        for table, total, kind in ((self.collLat, self.totalCommsLatency, ""), (self.compLat, self.totalCompsLatency, " (compute)")):
            for name, lats in table.items():
                if not lats:
                    continue
                lat = np.array(lats)
                print("{}\n Replayed {} {}{} ({:.2f}%): \n{}".format("-" * 50, len(lats), name, kind,
                                                                     lat.sum() / max(total, 1e-12) * 100, "-" * 50))
                print(f"Latency (us)\n{lat_hdr}")
                print(lat_row.format(*six(lat)))
        if self.colls_per_batch > 0 and self.batchLat:
            print("\n{} Batch Latency Performance {}".format("=" * 20, "=" * 20))
            print(f"Batch Latency (ms)\n{lat_hdr}")
            print(lat_row.format(*six(np.array(self.batchLat))))

    # ------------------------------------------------------------------ tensors
    def _alloc(self, n, commsParams, ones=False):
        dev, dtype = commsParams.device, commsParams.dtype
        if ones:
            return self.backendFuncs.alloc_ones([n], dev, dtype, self.initVal)
        return self.backendFuncs.alloc_random([n], dev, dtype, max(self.collectiveArgs.world_size, 1))

    def prepComms(self, curComm: commsArgs, commsParams, regenerateTensors: bool = True):
        """input / output tensors of one recorded collective: (ipTensor, opTensor)"""
        commOp = paramToCommName(curComm.comms)
        if commOp in ("wait", "barrier", "batch_isend_irecv"):
            return ([], [])
        ca = self.collectiveArgs
        if self.backendFuncs is not None and not self.shrink and curComm.pgId is None:
            ca.group = self.backendFuncs.get_default_group()
        if self.shrink:
            cur_ws = ca.world_size
            real_ws = cur_ws
            if curComm.worldSize is not None:
                real_ws = curComm.worldSize
            elif commOp == "all_to_allv":    # infer the recorded scale from the split lists
                if curComm.inSplit:
                    real_ws = len(curComm.inSplit)
                elif curComm.outSplit:
                    real_ws = len(curComm.outSplit)
            n_in = (curComm.inMsgSize // real_ws) * cur_ws
            n_out = (curComm.outMsgSize // real_ws) * cur_ws
            if commOp == "all_to_allv":
                curComm.outSplit = curComm.outSplit[:cur_ws] if curComm.outSplit is not None else []
                curComm.inSplit = curComm.inSplit[:cur_ws] if curComm.inSplit is not None else []
                if curComm.inSplit:
                    n_in = sum(curComm.inSplit)
                if curComm.outSplit:
                    n_out = sum(curComm.outSplit)
            elif commOp == "all_gather":
                n_out = n_in * cur_ws
            curComm.inMsgSize, curComm.outMsgSize, curComm.worldSize = n_in, n_out, cur_ws
        commsParams.size_from_trace = True
        if curComm.dtype is None:               # an entry without a data type (hand-built traces): the parser's default
            curComm.dtype = "float32"
        commsParams.dtype = self.dtypeMap[curComm.dtype]
        key = (commOp, curComm.inMsgSize, curComm.outMsgSize, curComm.dtype, str(curComm.inSplit), str(curComm.outSplit))
        if not regenerateTensors and key in self.tensorReuse:
            ip, op = self.tensorReuse[key]
            self._set_splits(commOp, curComm)
            if commsParams.dcheck == 1:      # in-place reductions overwrote the cached input: back to ones
                for t in (ip if isinstance(ip, (list, tuple)) else [ip]):
                    t.fill_(self.initVal)
            return ip, op
        ip, op = self._prep(commOp, curComm, commsParams)
        if not regenerateTensors:
            self.tensorReuse[key] = (ip, op)
        return ip, op

    def _set_splits(self, commOp, curComm):
        ca = self.collectiveArgs
        world = curComm.worldSize if curComm.worldSize else max(ca.world_size, 1)
        if commOp in ("all_to_allv", "all_to_all_single"):
            ca.opTensor_split = curComm.outSplit if curComm.outSplit else [curComm.outMsgSize // world] * world
            ca.ipTensor_split = curComm.inSplit if curComm.inSplit else [curComm.inMsgSize // world] * world
        else:
            ca.opTensor_split, ca.ipTensor_split = [], []
        return world

    def _prep(self, commOp, curComm, commsParams):
        ones = commsParams.dcheck == 1
        world = self._set_splits(commOp, curComm)
        n_in, n_out = curComm.inMsgSize, curComm.outMsgSize
        if commOp == "all_to_all":           # list form: one tensor per peer
            ip = [self._alloc(n_in // world, commsParams, ones) for _ in range(world)]
            op = [self._alloc(n_out // world, commsParams) for _ in range(world)]
            return ip, op
        ip = self._alloc(n_in, commsParams, ones)
        if commOp in ("all_reduce", "reduce", "broadcast"):
            return ip, ip                    # in place
        return ip, self._alloc(n_out, commsParams)

    def rebalanceSplit(self, curComm: commsArgs) -> None:
        """``--rebalance-policy equal``: every rank sends the same amount to every peer; the total is the
        all-reduced sum of the recorded input sizes rounded to a multiple of world_size**2"""
        if self.rebalance_policy == "equal":
            ca = self.collectiveArgs
            ca.ipTensor = torch.tensor([curComm.inMsgSize], dtype=torch.int, device=ca.device)
            self.backendFuncs.collectiveFunc["all_reduce"](ca)
            self.backendFuncs.complete_accel_ops(ca)
            w2 = ca.world_size * ca.world_size
            total = w2 * round(ca.ipTensor[0].item() / w2)
            curComm.inMsgSize = total // ca.world_size
            curComm.outMsgSize = curComm.inMsgSize
            curComm.inSplit = [curComm.inMsgSize // ca.world_size] * ca.world_size
            curComm.outSplit = curComm.inSplit
        else:
            logger.error("Unsupported balancing policy. Ignoring.")

    def commRebalance(self, curComm: commsArgs) -> None:
        if curComm.comms == "all_to_allv" and self.rebalance_policy:
            self.rebalanceSplit(curComm)

    def resetComms(self) -> None:
        """drop outstanding handles between replays"""
        self.collectiveArgs.waitObj.clear()
        self.collectiveArgs.waitObjIds.clear()

    # ------------------------------------------------------------------ one operation
    def runCompute(self, func, curBlockStack: str):
        ca, bf = self.collectiveArgs, self.backendFuncs
        timer = paramTimer()
        with paramProfile(timer=timer, description=f"# PARAM replay {getattr(self, 'replayIter', 0)}: " + curBlockStack):
            with paramStreamGuard(stream=ca.compute_stream, curDevice=ca.device, backendFuncs=bf, is_blocking=False):
                for _ in range(ca.computeCount):
                    func(ca)
            if self.is_blocking:      # blocking replay times the kernels, non-blocking replay their launch
                bf.sync_stream(ca.compute_stream, ca.device)
        lat = timer.getTimeUS()
        return lat, lat

    def runComms(self, collName: str, curComm: commsArgs, curBlockStack: str):
        ca, bf = self.collectiveArgs, self.backendFuncs
        ca.quant_time.reset()          # per replayed collective (reference commsTraceReplay.py:769-770)
        ca.dequant_time.reset()
        timer = paramTimer()
        it = getattr(self, "replayIter", 0)
        # the ranges carry the reference's labels (commsTraceReplay.py:771-790, 824-830): a profiler trace of a replay shows every
        # collective under its marker stack, the fences around it apart
        if self.is_blocking:
            with paramProfile(description=f"# PARAM replay {it} pre-comm barrier # " + curBlockStack):
                bf.sync_barrier(ca)
        with paramProfile(timer=timer, description=f"# PARAM replay {it}:" + curBlockStack):
            if collName in bf.collectiveFunc:
                if curComm.req is not None:
                    ca.collectiveId = str(curComm.req)
                retObj = bf.collectiveFunc[collName](ca, retFlag=True)
            else:
                retObj = None
                logger.warning(f"Unsupported collective name: {collName}. Skipping replaying the collective")
            if self.is_blocking:
                bf.complete_accel_ops(ca)
            if curComm.req is not None and not self.is_blocking and collName != "wait":
                ca.waitObjIds[str(curComm.req)] = retObj      # a later "wait" entry names this request
        latency = global_latency = timer.getTimeUS()
        if self.is_blocking:
            post = paramTimer()
            with paramProfile(timer=post, description=f"# PARAM replay {it} post-comm barrier # " + curBlockStack):
                bf.sync_barrier(ca)
            global_latency = latency + post.getTimeUS()
        return latency, global_latency

    def waitForTimestamp(self, curComm: commsArgs, startTime: float) -> None:
        if curComm.startTimeNs is not None:
            while time.monotonic_ns() - startTime <= curComm.startTimeNs:
                if (curComm.startTimeNs - (time.monotonic_ns() - startTime)) / 1e9 >= LOOP_TIMER_S:
                    time.sleep(LOOP_TIMER_S)

    def prepComputeReplay(self, commsParams, curComm):
        ca, bf = self.collectiveArgs, self.backendFuncs
        ca.computeCount = curComm.count
        ca.reuseTensors = self.reuse_tensors
        if curComm.compute != "emb_lookup":
            raise ValueError(f"compute kernel {curComm.compute} is not replayable in this build")
        key = curComm.toEmbLookupTuple()
        if self.reuse_tensors and key in self.embLookupReuse:
            (ca.direction, ca.emb_dim, ca.batch_size, ca.num_emb_ops, ca.num_emb_tables_batched, ca.embRequests, ca.emb,
             ca.LookupOut, ca.grad_output) = self.embLookupReuse[key]
        else:
            curComm.device = commsParams.device
            comms_utils.init_emb_lookup(ca, curComm, bf)
            if self.reuse_tensors:
                self.embLookupReuse[key] = (ca.direction, ca.emb_dim, ca.batch_size, ca.num_emb_ops, ca.num_emb_tables_batched,
                                            ca.embRequests, ca.emb, ca.LookupOut, ca.grad_output)
        if ca.compute_stream is None:
            ca.compute_stream = bf.get_new_stream()
        return bf.computeFunc["emb_lookup"]

    def recordCommReplay(self, commsParams, curComm, collName, latency, curBlockStack, global_latency, curBlocks):
        rec = curComm.toDict()
        rec["dtype_size"] = torch.tensor([], dtype=self.dtypeMap[curComm.dtype]).element_size() if curComm.dtype in self.dtypeMap else 0
        rec["marker_stack"] = curBlockStack
        rec["quant_us"] = self.collectiveArgs.quant_time.getTimeUS()        # (reference :952-953)
        rec["dequant_us"] = self.collectiveArgs.dequant_time.getTimeUS()
        rec["latency_us"] = latency
        rec["global_latency_us"] = global_latency
        if curComm.compute is not None:
            self.compLat.setdefault(collName, []).append(latency)
            self.totalCompsLatency += latency
        else:
            self.collLat.setdefault(collName, []).append(latency)
            self.totalCommsLatency += latency
            for block in curBlocks:
                self.comms_blocks[block].append(rec)
        self.traceWithPerf.append(rec)

    def getCommGroupInfo(self, curComm: commsArgs, commsParams):
        """(rank in the group, description); -1 when this process is not a member.  Only the default group exists here:
        an entry recorded on another ``pg_id`` is skipped unless ``--auto-shrink`` folds it onto the default group."""
        if curComm.pgId is not None and not self.shrink and curComm.pgId not in (0, "0"):
            return -1, f"PG: id={curComm.pgId} (not created in this build)"
        return max(self.collectiveArgs.global_rank, 0), "default group"

    # ------------------------------------------------------------------ the replay
    def replayTrace(self, commsParams, warmup: bool = False) -> None:
        coll_in_batch_num = 0
        batch_begin = 0.0
        startTime = time.monotonic_ns()
        limit = self.max_msg_cnt          # initTraceStat() turns "0 = no limit" into the trace's length (reference :995)
        allow = self.allowList
        for cnt, curComm in enumerate(self.comms_trace[:limit]):
            curBlocks = curComm.markerStack if curComm.markerStack is not None else []
            curBlockStack = " ".join(curBlocks) if len(curBlocks) > 0 else "Unamed/Unknown"
            if curComm.compute is not None:
                func = self.prepComputeReplay(commsParams, curComm)
                latency, global_latency = self.runCompute(func, curBlockStack)
                recordName = curComm.compute
            else:
                if warmup:
                    self.commRebalance(curComm)
                collName = paramToCommName(curComm.comms)
                groupRank, _ = self.getCommGroupInfo(curComm, commsParams)
                if (allow and collName not in allow) or groupRank == -1 or (collName == "wait" and self.is_blocking):
                    continue
                self.collectiveArgs.ipTensor, self.collectiveArgs.opTensor = self.prepComms(
                    curComm, commsParams, not self.reuse_tensors)
                if not warmup and self.colls_per_batch > 0 and coll_in_batch_num == 0:
                    batch_begin = time.monotonic()
                if not warmup and self.use_timestamp:
                    self.waitForTimestamp(curComm, startTime)
                latency, global_latency = self.runComms(collName, curComm, curBlockStack)
                if self.is_blocking and commsParams.dcheck == 1 and collName not in ("wait", "barrier"):
                    self.dcheck(collName, curComm)
                if not warmup and collName == "wait" and self.colls_per_batch > 0:
                    coll_in_batch_num += 1
                    if coll_in_batch_num == self.colls_per_batch:
                        self.batchLat.append((time.monotonic() - batch_begin) * 1e3)
                        coll_in_batch_num = 0
                recordName = collName
            if not warmup:
                self.recordCommReplay(commsParams, curComm, recordName, latency, curBlockStack, global_latency, curBlocks)

    def dcheck(self, collName, curComm) -> None:
        """``--c 1`` (blocking only): inputs are ones, so all_to_all* outputs are ones and all_reduce gives world_size"""
        ca = self.collectiveArgs
        expect = self.initVal * (ca.world_size if collName == "all_reduce" else 1)
        tensors = ca.opTensor if isinstance(ca.opTensor, (list, tuple)) else [ca.opTensor]
        for t in tensors:
            if t.numel() and not bool((t == expect).all()):
                raise ValueError(f"[{ca.global_rank}] replayed {collName} (id {curComm.id}): "
                                 f"{int((t != expect).sum())} elements differ from {expect}")

    def setBench(self, commsParams) -> None:
        bf, ca = self.backendFuncs, self.collectiveArgs
        ca.group = bf.get_default_group()
        ca.groups = bf.get_groups()
        ca.num_pgs = bf.get_num_pgs()
        ca.device = bf.get_device()
        ca.world_size = bf.get_world_size()
        ca.global_rank = bf.get_global_rank()
        ca.backendFuncs = bf
        ca.srcOrDst = 0
        ca.op = bf.get_reduce_op("sum")
        ca.asyncOp = not self.is_blocking
        ca.ipTensor = ca.opTensor = None
        ca.quant_threshold = getattr(commsParams, "quant_threshold", 0)      # (reference :1382)
        if getattr(commsParams, "bitwidth", 32) < 32:                        # (reference :1424-1425)
            comms_utils.initQuantCommCtx(ca, commsParams)
        if self.allowList in ("all", "default", "*"):
            self.allowList = list(bf.collectiveFunc.keys())
        elif isinstance(self.allowList, str):
            self.allowList = [paramToCommName(op) for op in self.allowList.split(",")]

    def benchTime(self, commsParams) -> None:
        bf, ca = self.backendFuncs, self.collectiveArgs
        if self.do_warm_up:
            self.replayIter = -1
            self.replayTrace(commsParams=commsParams, warmup=True)
        self.resetComms()
        bf.sync_barrier(ca)
        prof = None
        if getattr(self, "enable_profiler", False):
            # the reference skips warm-up + --profiler-num-replays-start replays and records at most --num-replays of them
            # (commsTraceReplay.py:1165-1180; its profiler is unpublished, here torch.profiler writes a chrome trace per rank)
            import torch

            first = min(self.profiler_num_replays_start, max(self.num_replays - 1, 0))
            active = max(1, min(self.profiler_num_replays, self.num_replays - first))
            acts = [torch.profiler.ProfilerActivity.CPU] + ([torch.profiler.ProfilerActivity.CUDA] if ca.device.type == "cuda" else [])
            out_dir = self.out_path or "."
            os.makedirs(out_dir, exist_ok=True)
            trace_file = os.path.join(out_dir, f"replay_profile_rank{bf.get_global_rank()}.json")
            prof = torch.profiler.profile(activities=acts, schedule=torch.profiler.schedule(wait=first, warmup=0, active=active),
                                          on_trace_ready=lambda p: p.export_chrome_trace(trace_file))
            prof.start()
        t0 = time.monotonic_ns()
        for i in range(self.num_replays):
            self.replayIter = i
            self.replayTrace(commsParams=commsParams, warmup=False)
            if prof is not None:
                prof.step()
            bf.complete_accel_ops(ca)     # every posted operation has finished before the handles are dropped
            self.resetComms()
            bf.sync_barrier(ca)
        self.totalTraceLatency = (time.monotonic_ns() - t0) / 1e3
        if prof is not None:
            prof.stop()
        bf.clear_memory(ca)

    def readRawTrace(self, rank: int) -> None:
        path = self.trace_file
        if os.path.isdir(path):
            path = os.path.join(path, f"{0 if self.use_one_trace else rank}.json")
        if getattr(self, "disable_parallel_read", False) and not self.is_dry_run and self.backendFuncs is not None:
            # one reader (commsTraceReplay.py:1496-1510): rank 0 loads the file and leaves it in the rendezvous store under the
            # path; every rank, rank 0 included, takes it from there (``store_get`` blocks until the key exists)
            if rank == 0:
                with open(path) as f:
                    self.backendFuncs.store_set(path, f.read())
            self.comms_trace = json.loads(self.backendFuncs.store_get(path))
            return
        with open(path) as f:
            self.comms_trace = json.load(f)

    def readTrace(self, rank: int, world_size: int = 1) -> None:
        self.readRawTrace(rank)
        self.comms_trace = commsTraceParser.parseTrace(self.comms_trace, self.trace_type, rank, world_size)

    def runBench(self, commsParams) -> None:
        rank = self.backendFuncs.get_global_rank()
        self.report = rank == 0
        self.readTrace(rank, self.backendFuncs.get_world_size())
        self.initTraceStat()
        if not self.is_dry_run:
            self.setBench(commsParams)
            self.benchTime(commsParams)
        if self.report:
            self.reportBenchTime()
        if not self.is_dry_run:
            if rank in self.outputRanks:
                writeCommDetails(self.traceWithPerf, folder=self.out_path, rank=rank)
            self.backendFuncs.sync_barrier(self.collectiveArgs)

    def initBackend(self, bootstrap_info, commsParams, args):
        register()
        if args.backend in customized_backend:
            backend_cls, c10d = customized_backend[args.backend], ("gloo" if commsParams.device == "cpu" else "nccl")
        else:
            backend_cls, c10d = MI355XBackend, args.backend
        self.backendFuncs = backend_cls(bootstrap_info, commsParams)
        self.backendFuncs.initialize_backend(bootstrap_info.master_ip, bootstrap_info.master_port, backend=c10d)
        return self.backendFuncs


def main(argv=None):
    bench = commsTraceReplayBench()
    parser = argparse.ArgumentParser(description="PARAM-Comms trace replay (MI355X / RCCL over xGMI build)")
    args = bench.readArgs(parser, argv)
    comms_utils.init_logging(args.log)
    bench.checkArgs(args)
    bench.setTraceFile(args)
    env = comms_utils.read_comms_env_vars()
    if env["world_size"] < 1:
        env = {"world_size": 1, "local_size": 1, "global_rank": 0, "local_rank": 0}
    if env["local_size"] < 1:
        env["local_size"] = env["world_size"]
    if env["local_rank"] < 0:
        env["local_rank"] = env["global_rank"] % max(1, env["local_size"])
    info = comms_utils.bootstrap_info_holder(args.master_ip, args.master_port, 0, env)
    commsParams = replayParamsHolder(args)
    bf = bench.initBackend(info, commsParams, args)
    bench.initBench(commsParams, args)
    try:
        bench.runBench(commsParams)
        return bench
    finally:
        bf.shutdown()


if __name__ == "__main__":
    main()  # pragma: no cover
