"""comms.py -- nccl-tests style sweep over the collectives of the backend table and the point-to-point patterns,
reporting p50/p75/p95 latency, AlgBW and BusBW per message size.

Own restatement of reference ``train/comms/pt/comms.py`` (CLI ``:50-206`` + comms_utils.py ``:1713-1879``, argument checks
``:208-374``, run_coll_non_graph ``:452-545``, run_coll_cuda_graph ``:375-450``, the four pt2pt measurements ``:554-759``,
rank checks ``:761-824``, initCollectiveArgs ``:826-925``, reports ``:1112-1283``, benchComm ``:1285-1429``, multi-comm
groups ``:1431-1469``) and of the tensor preparation it inherits (comms_utils.py ``:1093-1696``):

  mpirun/torchrun -np N python -m param_amd.comms.pt.comms --master-ip 127.0.0.1 --b 8 --e 256M \
        --n 100 --f 2 --z 1 --collective all_to_all --backend rccl_xgmi --device rocm

Kept: the reference's flag names and defaults, rank discovery from the launcher's environment, per-size tensor shapes of
every collective (one table below instead of a method per collective), blocking (``--z 1``: barrier + wait + device sync
per iteration) vs non-blocking timing, ``--c 1`` validation, ``--i / --o`` splits, ``--ss``, ``--num-coll``, ``--root``,
``--multi-comms`` (rank r works in group r % k; the report still spans all ranks), ``--pt2pt one2one | pairwise`` with
``--src-ranks / --dst-ranks / --window`` (ping, ping-pong, uni- and bi-directional bandwidth), ``--tag``, the preamble lines,
the report recomputing AlgBW from the p50 of per-rank mean latencies, busBW per collective, and the row formats -- all held
to what the reference prints on gloo ranks by tests/golden/comms_surface.json.  Not kept: ``all_gather_v`` /
``reduce_scatter_v`` (the reference's own backend has no entry for them either), the TPU / NVSHMEM / torchcomms stacks.
``--use-perf-logger`` hands every reported row to the loggers registered in logger_utils.py (the reference's plug-in point);
``--size-start-profiler`` drives torch.profiler (the reference's hook is an unpublished profiler).
"""
from __future__ import annotations

import argparse
import logging
import os
import time

import numpy as np
import torch

from . import comms_utils, logger_utils
from .comms_utils import paramDeviceTimer, paramStreamGuard
from .mi355_backend import BACKEND_NAME, MI355XBackend, register
from .pytorch_backend_utils import collectiveArgsHolder, customized_backend, pt2ptPatterns, supportedCollectives

logger = logging.getLogger(__name__)

_DTYPES = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16, "int32": torch.int32,
           "long": torch.long, "float64": torch.float64, "int8": torch.int8, "uint8": torch.uint8, "int": torch.int32,
           "float": torch.float32, "double": torch.float64, "half": torch.float16, "bool": torch.bool}

HEADER_FMT = "{:>40}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}{:>18}"
QUANT_HEADER_FMT = "-QUANT\t{:>40}{:>18}{:>25}{:>15}{:>15}{:>15}"
QUANT_ROW_FMT = "\tCOMMS-RES-QUANT-{}-{}{}\t{:>15}{:>18}{:>25}{:>15}{:>15}{:>15}"
ROW_FMT = "\tCOMMS-RES-{}-{}{}{:>18}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}{:>20}"
PT2PT_HEADER_FMT = "{:>40}{:>20}{:>10}{:>10}{:>25}{:>10}{:>10}{:>15}{:>15}{:>18}{:>18}"
PT2PT_ROW_FMT = "\tCOMMS-RES-{}-{}{}{:>15}{:>20}{:>10}{:>10}{:>25}{:>10}{:>10}{:>15}{:>15}{:>18}{:>18}"

# Tensor shapes of one sweep point with N elements on a group of W ranks (reference: one ``_prep_*`` method per collective,
# comms_utils.py:1093-1515, and the in-place default ``:1688-1690``).  "N": one tensor of N elements, "N/W": one of N // W,
# "WxN/W": a list of W such tensors, "SxN": one tensor of N per source rank of an incast, "=": output is the input (in place).
# The all_to_all family is shaped in ``_prep_all_to_all_family``.
_SHAPES = {
    "all_reduce": ("N", "="), "reduce": ("N", "="), "broadcast": ("N", "="), "multicast": ("N", "="),
    "broadcast_object_list": ("N", "="),
    "all_gather": ("N/W", "WxN/W"), "gather": ("N/W", "WxN/W"), "all_gather_object": ("N/W", "WxN/W"),
    "all_gather_base": ("N/W", "N"),
    "reduce_scatter": ("WxN/W", "N/W"), "scatter": ("WxN/W", "N/W"),
    "reduce_scatter_base": ("N", "N/W"),
    "incast": ("N", "SxN"),
    "pt2pt": ("N", "N"),
}
_HOST_SIDE = ("all_gather_object", "broadcast_object_list")          # pickled objects: tensors stay on the host (:1322, :1636)
# rows report elements PER RANK for these (reportBenchTime, comms.py:1062-1075)
_PER_RANK_ROWS = ("reduce_scatter", "reduce_scatter_v", "reduce_scatter_base", "all_gather", "all_gather_v", "all_gather_base")


class commsParamsHolder:
    """Run parameters (the reference's commsParamsHolderBase/commsParamsHolder, comms_utils.py:801-910)."""

    def __init__(self, args, element_size: int, dtype, collective: str, groupRanks=None):
        self.nw_stack = args.nw_stack
        self.dtype = dtype
        self.backend = args.backend
        self.device = "cuda" if args.device == "rocm" else args.device
        self.blockingFlag = args.z
        self.num_pgs = 1
        self.dcheck = args.c
        self.element_size = element_size
        self.beginSize = args.b
        self.endSize = args.e
        self.maxSize = args.e
        self.stepFactor = args.f
        self.stepBytes = args.sb
        self.collective = collective
        self.numWarmupIters = args.w
        self.numIters = args.n
        self.bitwidth = args.bitwidth
        self.quant_a2a_embedding_dim = args.quant_a2a_embedding_dim
        self.quant_threshold = max(args.e, args.quant_threshold)     # as the reference (comms_utils.py:885-887)
        self.init_only = getattr(args, "init_only", False)
        self.use_device_time = args.use_device_time
        self.include_0B = args.include_0B
        self.graph_launches = getattr(args, "graph_launches", 0)
        self.init_method = getattr(args, "init_method", None)
        self.eager_init = getattr(args, "eager_init", False)
        self.use_perf_logger = getattr(args, "use_perf_logger", None)
        self.use_ext_dist = False
        # round 4: the rest of the reference's holder (comms_utils.py:873-910)
        self.sizes = getattr(args, "ss", None)
        self.inSplit = getattr(args, "i", None)
        self.outSplit = getattr(args, "o", None)
        self.srcOrDst = getattr(args, "root", 0)
        self.num_coll = getattr(args, "num_coll", 1)
        self.multi_comms = getattr(args, "multi_comms", 1)
        self.pt2pt = getattr(args, "pt2pt", None)
        self.window = getattr(args, "window", 100)
        self.src_ranks = getattr(args, "src_ranks", None)
        self.dst_ranks = getattr(args, "dst_ranks", None)
        self.size_start_profiler = getattr(args, "size_start_profiler", None)
        self.profiler_active_iters = getattr(args, "profiler_active_iters", None)
        self.enable_local_report = getattr(args, "enable_local_report", False)
        self.groupRanks = groupRanks if groupRanks is not None else {}


def format_header() -> str:
    """the reference's preamble line (``printPreamble``, comms.py:956-1001): plain ``COMMS-RES`` + the column titles"""
    return "\n\tCOMMS-RES" + HEADER_FMT.format(
        "total-size (B)", "nElementsPerRank", "Time(us):p50", "p75", "p95", "Min", "Max", "AlgBW(GB/s)",
        "BusBW(GB/s)", "TotalTime(us):p50")


def format_quant_header() -> str:
    """the ``--bitwidth < 32`` preamble (comms.py:976-986); ``str.format`` drops the reference's surplus seventh title"""
    return "\n\tCOMMS-RES" + QUANT_HEADER_FMT.format("size (B)", "nElementsPerRank", "P95 Latency(us): Quant", "Comms",
                                                      "De-Quant", "Overall", "TotalLatency(us):p50")


def format_pt2pt_header() -> str:
    """the ``--pt2pt`` preamble (comms.py:960-975); the twelfth title has no field in the reference's format and is dropped"""
    return "\n\tCOMMS-RES" + PT2PT_HEADER_FMT.format(
        "size (B)", "pingLatency(us):p50", "p75", "p95", "pingPongLatency(us):p50", "p75", "p95", "avgUniBW(GB/s)",
        "avgBiBW(GB/s)", "totalUniBW(GB/s)", "totalBiBW(GB/s)", "TotalLatency(us):p50")


def format_quant_row(collective, data_type, tag, memSize, numElements, quant_p95, dequant_p95, p95):
    """``reportBenchTimeCollWithQuant`` (comms.py:1005-1040): comms = overall p95 - quant p95 - de-quant p95"""
    return QUANT_ROW_FMT.format(collective, data_type, tag, memSize, "%d" % numElements, "%.1f" % quant_p95,
                                "%.1f" % (p95 - quant_p95 - dequant_p95), "%.1f" % dequant_p95, "%.1f" % p95)


def format_row(collective, data_type, tag, memSize, numElements, p50, p75, p95, mn, mx, algBW, busBW, total_p50=0.0):
    return ROW_FMT.format(collective, data_type, tag, memSize, "%d" % numElements, "%.1f" % p50, "%.1f" % p75,
                          "%.1f" % p95, "%.1f" % mn, "%.1f" % mx, "%.3f" % algBW, "%.3f" % busBW, "%.1f" % total_p50)


def format_pt2pt_row(collective, data_type, tag, memSize, ping, pingpong, avgUniBW, avgBiBW, totalUniBW, totalBiBW):
    """``reportBenchTimePt2Pt`` (comms.py:1243-1262); ``ping`` / ``pingpong`` = (p50, p75, p95) in us"""
    return PT2PT_ROW_FMT.format(collective, data_type, tag, memSize, *("%.1f" % v for v in ping),
                                *("%.1f" % v for v in pingpong), "%.3f" % avgUniBW, "%.3f" % avgBiBW,
                                "%.3f" % totalUniBW, "%.3f" % totalBiBW)


def _int_list(s: str):
    return [int(item) for item in s.split(",") if item]


class commsCollBench:
    def __init__(self):
        self.collectiveArgs = collectiveArgsHolder()
        self.backendFuncs = None
        self.tag = ""
        self.initVal = 1
        self.results = []
        self.report = True
        self.comm_size = 1
        self.groupRanks = {0: [0]}

    # ------------------------------------------------------------------ args
    def readArgs(self, parser: argparse.ArgumentParser):
        gpu = torch.cuda.is_available()
        parser.add_argument("--master-ip", type=str, default=os.environ.get("MASTER_ADDR", "127.0.0.1"))
        parser.add_argument("--master-port", type=str, default=os.environ.get("MASTER_PORT", "29500"))
        parser.add_argument("--backend", type=str, default=BACKEND_NAME, help="rccl_xgmi | nccl | gloo")
        parser.add_argument("--nw-stack", type=str, default="pytorch-dist")
        parser.add_argument("--device", type=str, default="rocm" if gpu else "cpu", choices=["cuda", "rocm", "cpu"])
        parser.add_argument("--w", "--warmup-iters", type=int, default=5, dest="w", help="number of warmup iterations")
        parser.add_argument("--n", "--num_iters", "--num-iters", type=int, default=5, dest="n", help="number of iterations")
        parser.add_argument("--num-coll", "--num-coll-per-iteration", type=int, default=1, dest="num_coll",
                            help="number of collective operations to execute for every iteration")
        parser.add_argument("--b", "--begin-size", type=str, default="8", dest="b", help="minimum size, in bytes, to start with")
        parser.add_argument("--e", "--end-size", type=str, default="64", dest="e", help="maximum size, in bytes, to end at")
        parser.add_argument("--f", "--step-factor", type=int, default=2, dest="f", help="multiplication factor between sizes")
        parser.add_argument("--sb", "--step-bytes", type=int, default=0, dest="sb",
                            help="step bytes between sizes, 0 disables the additive step and uses --f")
        parser.add_argument("--i", "--in-split", type=_int_list, default=None, dest="i",
                            help="comma-separated split of number of elements in input tensor")
        parser.add_argument("--o", "--out-split", type=_int_list, default=None, dest="o",
                            help="comma-separated split of number of elements in output tensor")
        parser.add_argument("--ss", "--sizes", type=_int_list, default=None, dest="ss",
                            help="benchmark only specified sizes, comma-separated")
        parser.add_argument("--z", "--blocking", type=int, default=0, dest="z", choices=[0, 1],
                            help="use blocking/non-blocking mode for collectives")
        parser.add_argument("--c", "--check", type=int, default=0, dest="c", choices=[0, 1], help="enable data validation check")
        parser.add_argument("--bitwidth", type=int, default=32, choices=[2, 4, 8, 16, 32], help="Quantization bitwidth")
        parser.add_argument("--quant-a2a-embedding-dim", type=int, default=32, choices=[32, 64, 128, 256],
                            help="Embedding dimension used by quantization alltoall if enabled")
        parser.add_argument("--quant-threshold", type=int, default=33554432,
                            help="threshold of message sizes to perform quantization if enabled")
        parser.add_argument("--collective", "--collectives", type=str, default="all_reduce", dest="collective",
                            help=f"collective operation(s), comma-separated; supported: {supportedCollectives}")
        parser.add_argument("--data-types", "--data-type", "--dtype", type=str, default="float32", dest="data_types")
        parser.add_argument("--root", type=int, default=0, help="root process for reduce / broadcast / gather / scatter")
        parser.add_argument("--src-ranks", type=str, nargs="?",
                            help="src ranks of an incast or of pt2pt: list separated by comma or start:end (pt2pt default: 0)")
        parser.add_argument("--dst-ranks", type=str, nargs="?",
                            help="dst ranks of a multicast or of pt2pt: list separated by comma or start:end (pt2pt default: 1)")
        parser.add_argument("--multi-comms", type=int, default=1, help="number of communicator groups (rank r joins group r %% k)")
        parser.add_argument("--pt2pt", type=str, default=None, choices=pt2ptPatterns, help="point to point pattern")
        parser.add_argument("--window", type=int, default=100, help="window size for pt2pt throughput test")
        parser.add_argument("--size-start-profiler", type=str, default=None, help="run torch.profiler at the specified size")
        parser.add_argument("--profiler-active-iters", "--pa", type=int, required=False, dest="profiler_active_iters",
                            help="iterations the profiler records (default: --n)")
        parser.add_argument("--tag", type=str, default=None, help="keyword added to the final output lines")
        parser.add_argument("--use-device-time", action="store_true", default=False)
        parser.add_argument("--include-0B", action="store_true", default=False)
        parser.add_argument("--graph-launches", type=int, default=0, help="Number of graph launches for each data-size")
        parser.add_argument("--enable-local-report", action="store_true", default=False,
                            help="every node's local rank 0 reports too")
        parser.add_argument("--init-only", action="store_true", default=False, help="initialise the backend and stop")
        parser.add_argument("--eager-init", action="store_true", default=False,
                            help="pass device_id to init_process_group: the RCCL communicator is created at once")
        parser.add_argument("--init-method", "--pg-init-method", type=str, default=None, dest="init_method",
                            help="URL for init_process_group (env://, tcp://host:port, file://...) instead of the TCP store")
        parser.add_argument("--enable-torch-nccl-timing", action="store_true", default=False,
                            help="TORCH_NCCL_ENABLE_TIMING=1: c10d records start events for every collective")
        parser.add_argument("--use-perf-logger", "--use-custom-perf-logger", nargs="+", type=str, default=None, dest="use_perf_logger",
                            help="names of registered performance loggers (logger_utils.register_perf_logger); built in: jsonl")
        parser.add_argument("--log", "--log-level", type=str, default="ERROR", dest="log")
        return parser.parse_args()

    def checkArgs(self, args):
        """argument checks that need no backend (reference checkBasicArgs + checkArgsdataType, comms.py:277-374)"""
        if getattr(args, "nw_stack", "pytorch-dist") != "pytorch-dist":           # comms_utils.py:1884-1888; the one stack of this build
            logger.error(f"Specified backend: {args.nw_stack} is not one of the supported backends: ['pytorch-dist']. "
                         "Make sure the input is using the correct case.")
            comms_utils.gracefulExit()
        args.b = comms_utils.parsesize(args.b)
        args.e = comms_utils.parsesize(args.e)
        if getattr(args, "pt2pt", None) is not None:                  # _checkPt2Pt (comms.py:208-216)
            args.collective = "pt2pt"
        args.collectives = [c.strip() for c in args.collective.split(",")]
        for c in args.collectives:
            if c not in supportedCollectives and c != "pt2pt":
                logger.error(f"Specified collective: {c} is not one of the supported collectives: {supportedCollectives}")
                comms_utils.gracefulExit()
        args.dtypes = [d.strip().lower() for d in args.data_types.split(",") if d.strip()]
        for d in args.dtypes:
            if d not in _DTYPES:
                logger.error(f"Specified dtype: {d} is not one of the supported commstyle: {list(_DTYPES)}")
                comms_utils.gracefulExit()
            if d == "bfloat16" and args.backend == "gloo":
                logger.error(f"Specified dtype: {d} does not work with gloo backend")
                comms_utils.gracefulExit()
        self.tag = f"-{args.tag}" if getattr(args, "tag", None) is not None else ""
        if getattr(args, "size_start_profiler", None):
            args.size_start_profiler = comms_utils.parsesize(args.size_start_profiler)
        i_split, o_split = getattr(args, "i", None), getattr(args, "o", None)
        if i_split is not None or o_split is not None:                # _check_for_in_out_split (comms.py:218-243)
            if "all_to_allv" not in args.collectives:
                logger.error("Collective does not support input-split argument (--i) or output-split argument (--o)")
                comms_utils.gracefulExit()
            args.split_elements = sum(i_split if i_split is not None else o_split)
        if args.b < 1:
            logger.warning(f"Starting size (--b {args.b}) should be greater than 1 byte...fix and continue")
            args.b = 1
        if args.e < args.b:
            logger.warning(f"the begin-size (--b {args.b}) is larger than the end-size (--e {args.e})")
        if args.device == "cpu" and args.backend == "nccl":
            raise ValueError(f"backend {args.backend} does not support device cpu")
        # --backend rccl_xgmi --device cpu: the plug-in moves host tensors over gloo (MI355XBackend._pg_backend), which is
        # how it runs under the reference's own comms.py on a box without GPUs (tests/golden/gen_ref_plugin_rows.py)
        if args.bitwidth < 32:                                       # _check_bitwidth (comms.py:251-265)
            if args.device == "cpu":
                logger.error(f"collective quantization may not be fully supported for {args.device}")
            for c in args.collectives:
                for d in args.dtypes:
                    comms_utils.checkQuantArgs(c, _DTYPES[d], args.b, args.quant_a2a_embedding_dim, args.z)
        if args.graph_launches > 0 and args.device not in ("cuda", "rocm"):          # comms.py:332-334
            logger.error("cuda graph is only supported for cuda or rocm device")
            comms_utils.gracefulExit()
        if args.graph_launches > 0 and args.bitwidth < 32:
            logger.error("--graph-launches replays the plain collectives: not with --bitwidth < 32 (host-side timers in the quantised path)")
            comms_utils.gracefulExit()
        if args.graph_launches > 0 and "pt2pt" in args.collectives:
            logger.error("--graph-launches replays collectives: not with --pt2pt")
            comms_utils.gracefulExit()
        reduce_ops = ("all_reduce", "reduce", "reduce_scatter", "reduce_scatter_base")
        if args.c == 1 and args.z == 0 and any(c in reduce_ops for c in args.collectives):        # comms.py:296-305
            logger.warning(f"Data validation is not supported for {list(reduce_ops)} in non-blocking mode, disabled and continue")
            args.c = 0

    def checkArgsWithBackend(self, args):
        """the checks that need the world size (reference checkArgs, comms.py:294-334)"""
        world = self.backendFuncs.get_world_size()
        for name, split in (("input", getattr(args, "i", None)), ("output", getattr(args, "o", None))):
            if split is not None and len(split) * max(1, getattr(args, "multi_comms", 1)) != world:
                logger.error(f"An {name} split must be provided for all participating ranks")
                comms_utils.gracefulExit()
        for name in ("src_ranks", "dst_ranks"):
            val = getattr(args, name, None)
            if val and isinstance(val, str):
                ranks = comms_utils.parseRankList(val)
                if len(ranks) == 0 or any(r < 0 or r >= world for r in ranks):
                    logger.error(f"wrong {name} ({ranks})")
                    comms_utils.gracefulExit()
                setattr(args, name, ranks)
        k = getattr(args, "multi_comms", 1)
        if k < 1 or world % k != 0:
            logger.error(f"--multi-comms {k}: the {world} ranks do not divide into {k} equal groups")
            comms_utils.gracefulExit()
        if not 0 <= getattr(args, "root", 0) < world // k:
            logger.error(f"--root {args.root} is not a rank of a group of {world // k}")
            comms_utils.gracefulExit()

    def genMultiCommGroups(self, multi_comms: int, backend: str):
        """``--multi-comms k``: rank r joins group r % k (comms.py:1431-1469); one group of all ranks otherwise"""
        bf, ca = self.backendFuncs, self.collectiveArgs
        rank, world = bf.get_global_rank(), bf.get_world_size()
        ca.pgId = 0
        if multi_comms > 1:
            ca.pgId = rank % multi_comms
            groupRanks = {pg: [r for r in range(world) if r % multi_comms == pg] for pg in range(multi_comms)}
            for pg, ranks in groupRanks.items():
                logger.info(f"PARAM COMMS Rank {rank} created group {pg} with ranks {ranks}")
            bf.groupRanks = groupRanks
            bf.initialize_groups(groupRanks, backend=backend)
        else:
            groupRanks = {0: list(range(world))}
        self.groupRanks = groupRanks
        return groupRanks

    # ------------------------------------------------------------------ tensors of one sweep point
    def _alloc_in(self, shape, dev, commsParams, scale):
        """an input tensor: ones x initVal under ``--c 1`` (predictable sums), random otherwise"""
        if commsParams.dcheck == 1:
            return self.backendFuncs.alloc_ones(shape, dev, commsParams.dtype, self.initVal)
        return self.backendFuncs.alloc_random(shape, dev, commsParams.dtype, scale)

    def _prep_all_to_all_family(self, commsParams, numElements, world, dev, scale):
        ca = self.collectiveArgs
        in_split, out_split = commsParams.inSplit, commsParams.outSplit
        if commsParams.collective == "all_to_allv" and (in_split is not None or out_split is not None):
            # explicit splits (comms_utils.py:1115-1124): N in, N out, the lists as given -- consistency is the caller's
            ca.ipTensor = self._alloc_in([numElements], dev, commsParams, scale)
            ca.opTensor = self.backendFuncs.alloc_random([numElements], dev, commsParams.dtype, scale)
            ca.ipTensor_split = list(in_split) if in_split is not None else [numElements // world] * world
            ca.opTensor_split = list(out_split) if out_split is not None else [numElements // world] * world
            return numElements
        used, _ = comms_utils.equal_splits(numElements, world)
        used = max(used, world)
        per = used // world
        ip = self._alloc_in([used], dev, commsParams, scale)
        op = self.backendFuncs.alloc_random([used], dev, commsParams.dtype, scale)
        if commsParams.collective == "all_to_all":  # list form: one tensor per peer
            ca.ipTensor = list(ip.split(per))
            ca.opTensor = list(op.split(per))
            ca.ipTensor_split, ca.opTensor_split = [], []
        else:
            ca.ipTensor, ca.opTensor = ip, op
            if commsParams.include_0B and world > 1 and commsParams.collective == "all_to_all_single":     # the reference's scope (:1180)
                mates = (self.groupRanks or {}).get(getattr(ca, "pgId", 0), [])
                me = mates.index(ca.global_rank) if ca.global_rank in mates else ca.global_rank      # rank inside the group
                ins = {i: [0] * world for i in range(world)}
                for i in range(world):
                    for j in range(world):
                        if j != (i + 1) % world:
                            ins[i][j] = used // (world - 1)
                    ins[i][i] += used % (world - 1)
                ca.ipTensor_split = ins[me]
                ca.opTensor_split = [ins[i][me] for i in range(world)]
                ca.opTensor = self.backendFuncs.alloc_random([sum(ca.opTensor_split)], dev, commsParams.dtype, scale)
            else:
                ca.ipTensor_split = [per] * world
                ca.opTensor_split = [per] * world
        return used

    def prepComm(self, commsParams, size_bytes: int):
        ca = self.collectiveArgs
        world = ca.world_size
        dev = ca.device
        coll = commsParams.collective
        numElements = max(size_bytes // commsParams.element_size, 1)
        scale = world
        if coll in ("all_to_all", "all_to_allv", "all_to_all_single"):
            numElements = self._prep_all_to_all_family(commsParams, numElements, world, dev, scale)
        else:
            in_shape, out_shape = _SHAPES[coll]
            if coll in _HOST_SIDE:
                dev = "cpu"
            per = max(numElements // world, 1)
            dims = {"N": [numElements], "N/W": [per]}
            if in_shape == "WxN/W":
                ca.ipTensor = [self._alloc_in([per], dev, commsParams, scale) for _ in range(world)]
            else:
                ca.ipTensor = self._alloc_in(dims[in_shape], dev, commsParams, scale)
            if coll == "broadcast_object_list":                      # each list element is pickled: one tensor in a list
                ca.ipTensor = [ca.ipTensor]
            if out_shape == "=":
                ca.opTensor = ca.ipTensor
            elif out_shape == "WxN/W":
                ca.opTensor = [self.backendFuncs.alloc_random([per], dev, commsParams.dtype, scale) for _ in range(world)]
            elif out_shape == "SxN":
                ca.opTensor = [self.backendFuncs.alloc_random([numElements], dev, commsParams.dtype, scale)
                               for _ in (ca.src_ranks or [])]
            else:
                ca.opTensor = self.backendFuncs.alloc_random(dims[out_shape], dev, commsParams.dtype, scale)
        ca.dataSize = numElements * commsParams.element_size
        ca.numElements = numElements
        ca.waitObj = []
        return numElements

    def setTensorVal(self, tensor):
        """``--c 1``: outputs are overwritten before every iteration so that a stale result cannot validate; in-place
        collectives get their input value back (reference setTensorVal, comms_utils.py:1057-1090)"""
        ca = self.collectiveArgs
        in_place = _SHAPES.get(ca.collective, ("N", "N"))[1] == "="
        val = self.initVal if in_place else -1
        for t in (tensor if isinstance(tensor, (list, tuple)) else [tensor]):
            if torch.is_tensor(t):
                t.fill_(bool(val) if t.dtype == torch.bool else val)

    # ------------------------------------------------------------------ one collective
    def runColl(self, comm_fn, dcheck=False):
        """Timing protocol of the reference's run_coll_non_graph (comms.py:452-545)."""
        ca, bf = self.collectiveArgs, self.backendFuncs
        bf.sync_barrier(ca, desc="runColl_begin")
        elapsed_ns = 0.0
        is_blocking = not ca.asyncOp
        dev_timer = getattr(ca, "comm_dev_time", None)
        groups = ca.groups if isinstance(ca.groups, dict) and ca.groups else {0: bf.get_default_group()}
        my_group = groups.get(getattr(ca, "pgId", 0), bf.get_default_group())
        prof = getattr(ca, "profiler", None)
        for it in range(ca.numWarmupIters + ca.numIters):
            if prof is not None:
                prof.step()
            if it == ca.numWarmupIters:
                bf.complete_accel_ops(ca)
                elapsed_ns = 0.0
                if dev_timer:
                    dev_timer.reset()
                ca.quant_time.reset()
                ca.dequant_time.reset()
            if dcheck:
                self.setTensorVal(ca.opTensor)            # reset before every iteration (comms.py:474-476)
            if is_blocking:
                bf.sync_barrier(ca)
            start = time.monotonic()
            with paramStreamGuard(stream=bf.get_current_stream(device=ca.device), curDevice=ca.device,
                                  backendFuncs=bf, is_blocking=False, timer=dev_timer):
                ca.group = my_group
                for _ in range(ca.numCollPerIter):
                    comm_fn(ca)
            if is_blocking:
                bf.complete_accel_ops(ca)
                if dev_timer:
                    dev_timer.elapsedTime()
            elapsed_ns += (time.monotonic() - start) * 1e9
        start = time.monotonic()
        bf.complete_accel_ops(ca)
        elapsed_ns += (time.monotonic() - start) * 1e9
        memSize = bf.get_mem_size(ca)
        if dev_timer and ca.use_device_time and is_blocking:
            elapsed_ns = dev_timer.elapsedTimeNS
        avgIterNS, algBW = comms_utils.getAlgBW(elapsed_ns, memSize, ca.numIters * ca.numCollPerIter)
        busBW = bf.getBusBW(ca.collective, algBW, ca)
        ca.group = bf.get_default_group()
        bf.sync_barrier(ca, desc="runColl_end")
        return {"timeUS": avgIterNS / 1e3, "algBW": algBW, "busBW": busBW, "memSize": memSize}

    def runCollGraph(self, comm_fn, dcheck=False):
        """``--graph-launches N``: the reference's run_coll_cuda_graph (comms.py:375-450) on HIP graphs -- warm-up on a side stream,
        ``numIters`` collectives captured into ONE hipGraph (blocking form: an async work handle cannot cross a capture), the
        graph replayed N times, wall time over ``numIters * numCollPerIter * N`` collectives.  The launch-bound small-message end
        of a sweep is where it matters: a replay costs one graph launch instead of ``numIters`` collective launches."""
        ca, bf = self.collectiveArgs, self.backendFuncs
        bf.sync_barrier(ca, desc="run_coll_cuda_graph_begin")
        groups = ca.groups if isinstance(ca.groups, dict) and ca.groups else {0: bf.get_default_group()}
        ca.group = groups.get(getattr(ca, "pgId", 0), bf.get_default_group())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(ca.numWarmupIters):
                comm_fn(ca)
        torch.cuda.current_stream().wait_stream(side)
        bf.complete_accel_ops(ca)            # nothing in flight that c10d's watchdog thread would query while the capture is open
        ca.asyncOp = False
        graph = torch.cuda.CUDAGraph()
        in_place = _SHAPES.get(ca.collective, ("N", "N"))[1] == "="
        # thread-local capture mode: the process group's watchdog thread polls its events with hipEventQuery, which a GLOBAL-mode
        # capture on this thread turns into an error (and an abort) in that thread
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(ca.numIters):
                if dcheck and in_place:
                    self.setTensorVal(ca.opTensor)              # reset inside the graph: every replay validates (comms.py:399-401)
                for _ in range(ca.numCollPerIter):
                    comm_fn(ca)
        start = time.monotonic()
        for _ in range(ca.graph_launches):
            graph.replay()
        bf.complete_accel_ops(ca)
        elapsed_ns = (time.monotonic() - start) * 1e9
        memSize = bf.get_mem_size(ca)
        avgIterNS, algBW = comms_utils.getAlgBW(elapsed_ns, memSize, ca.numIters * ca.numCollPerIter * ca.graph_launches)
        busBW = bf.getBusBW(ca.collective, algBW, ca)
        ca.group = bf.get_default_group()
        bf.sync_barrier(ca, desc="runColl_end")
        return {"timeUS": avgIterNS / 1e3, "algBW": algBW, "busBW": busBW, "memSize": memSize}

    # ------------------------------------------------------------------ point to point (comms.py:554-759)
    def _pt2pt_role(self):
        """(is_src, is_dst, peer): the i-th source talks to the i-th destination"""
        ca = self.collectiveArgs
        if ca.global_rank in ca.src_ranks:
            return True, False, ca.dst_ranks[ca.src_ranks.index(ca.global_rank)]
        if ca.global_rank in ca.dst_ranks:
            return False, True, ca.src_ranks[ca.dst_ranks.index(ca.global_rank)]
        return False, False, -1

    def _timed_rounds(self, body):
        """``body()`` between a barrier and a completion, per round; the rounds after the warm-up, in ns"""
        ca, bf = self.collectiveArgs, self.backendFuncs
        lat = []
        for it in range(ca.numWarmupIters + ca.numIters):
            bf.sync_barrier(ca)
            start = time.monotonic()
            body()
            bf.complete_accel_ops(ca)
            if it >= ca.numWarmupIters:
                lat.append((time.monotonic() - start) * 1e9)
        return lat

    def getPingLatency(self):
        """one-way: the source sends, the destination receives, both blocking"""
        ca, bf = self.collectiveArgs, self.backendFuncs
        is_src, is_dst, peer = self._pt2pt_role()
        ca.asyncOp = False

        def body():
            if is_src:
                ca.dst_rank = peer
                bf.send(ca)
            elif is_dst:
                ca.src_rank = peer
                bf.recv(ca)
        return self._timed_rounds(body)

    def getPingPongLatency(self):
        """round trip: send then receive on the source, the mirror image on the destination"""
        ca, bf = self.collectiveArgs, self.backendFuncs
        is_src, is_dst, peer = self._pt2pt_role()
        ca.asyncOp = False
        ca.src_rank = ca.dst_rank = peer

        def body():
            if is_src:
                bf.send(ca)
                bf.recv(ca)
            elif is_dst:
                bf.recv(ca)
                bf.send(ca)
        return self._timed_rounds(body)

    def _windowed_bw(self, both_ways: bool, memSize: int):
        """``window`` messages batched into one isend / irecv group per round; bytes of one message (two when both ways)
        over the per-message time"""
        ca, bf = self.collectiveArgs, self.backendFuncs
        is_src, is_dst, peer = self._pt2pt_role()
        ca.asyncOp = True
        ca.src_rank = ca.dst_rank = peer
        first, second = ("send", "recv") if is_src else ("recv", "send")

        def body():
            if is_src or is_dst:
                for w in range(ca.window):
                    ca.collective = first
                    bf.P2POp(ca, tag=w)
                    if both_ways:
                        ca.collective = second
                        bf.P2POp(ca, tag=w + ca.window)
            bf.batch_isend_irecv(ca)
        lat = self._timed_rounds(body)
        per_msg_ns = float(np.mean(np.array(lat) / ca.window))
        _, bw = comms_utils.getAlgBW(per_msg_ns, (2 if both_ways else 1) * memSize, ca.numCollPerIter)
        return bw

    def runPt2Pt(self):
        ca, bf = self.collectiveArgs, self.backendFuncs
        bf.sync_barrier(ca)
        memSize = bf.get_mem_size(ca)
        bf.sync_barrier(ca, "runpt2pt_begin")
        ping = self.getPingLatency()
        pingpong = self.getPingPongLatency()
        uni = self._windowed_bw(False, memSize)
        bi = self._windowed_bw(True, memSize)
        bf.sync_barrier(ca, "runpt2pt")
        return {"pingPerIterNS": ping, "pingPongPerIterNS": pingpong, "avgUniBW": uni, "avgBiBW": bi, "memSize": memSize}

    def checkPt2PtRanks(self):
        ca = self.collectiveArgs
        ca.src_ranks = ca.src_ranks or [0]
        ca.dst_ranks = ca.dst_ranks or [1]
        problem = None
        if ca.pt2pt == "one2one" and (len(ca.src_ranks) > 1 or len(ca.dst_ranks) > 1):
            problem = "One2one Pt2Pt requires only a single rank is specified in src_ranks and dst_ranks! "
        elif ca.pt2pt == "pairwise" and len(ca.src_ranks) != len(ca.dst_ranks):
            problem = "Pairwise Pt2Pt requires identical number of members in src_ranks and dst_ranks! "
        elif ca.pt2pt == "pairwise" and set(ca.src_ranks) & set(ca.dst_ranks):
            problem = "Pairwise Pt2Pt requires distinct members in src_ranks and dst_ranks! "
        elif any(r >= self.comm_size for r in ca.src_ranks + ca.dst_ranks):
            problem = f"pt2pt ranks {ca.src_ranks} -> {ca.dst_ranks} do not exist in a world of {self.comm_size}"
        if problem:
            if self.report:
                logger.error(problem)
            comms_utils.gracefulExit()
        if self.report:
            print(f"\t collective={ca.collective}\t{ca.pt2pt}, src_ranks={ca.src_ranks}, dst_ranks={ca.dst_ranks}")

    def checkCollectiveRanks(self):
        """incast / multicast: all ranks but the root by default (comms.py:808-824)"""
        ca = self.collectiveArgs
        if ca.collective == "incast":
            ca.src_ranks = [r for r in (ca.src_ranks or range(self.comm_size)) if r != ca.srcOrDst]
        elif ca.collective == "multicast":
            ca.dst_ranks = [r for r in (ca.dst_ranks or range(self.comm_size)) if r != ca.srcOrDst]
        if self.report:
            print(f"\t collective={ca.collective}, src_ranks={ca.src_ranks}, dst_ranks={ca.dst_ranks}")

    # ------------------------------------------------------------------ validation and reports
    def dcheck(self, commsParams, curSize):
        """``--c 1``: inputs are ones x initVal, so copies give initVal and sums give group size x initVal; only the ranks
        that receive are checked (reference dcheck, comms_utils.py:997-1055)."""
        ca = self.collectiveArgs
        coll, rank = commsParams.collective, ca.global_rank
        expect = self.initVal
        if coll in ("all_reduce", "reduce_scatter", "reduce_scatter_base") or (coll == "reduce" and rank == ca.srcOrDst):
            expect = ca.world_size * self.initVal
        if coll in ("incast", "reduce", "gather") and rank != ca.srcOrDst:
            return
        if coll in ("multicast", "pt2pt") and rank not in (ca.dst_ranks or []):
            return
        tensors = ca.opTensor if isinstance(ca.opTensor, (list, tuple)) else [ca.opTensor]
        for k, t in enumerate(tensors):
            want = True if t.dtype == torch.bool else expect
            if t.numel() and not bool((t == want).all()):
                bad = (t != want).nonzero()
                first = int(bad[0][0])
                raise ValueError(f"[{curSize}-bytes {coll}] Wrong value at [{k}][{first}] = {t.reshape(-1)[first]}, expected {want} "
                                 f"({bad.shape[0]} elements differ on rank {rank})")

    def gatherBenchTime(self, values):
        """every rank's numbers on every rank: an all_gather over ALL ranks (also under --multi-comms) -> [ranks, len(values)]"""
        ca, bf = self.collectiveArgs, self.backendFuncs
        values = [values] if np.isscalar(values) else list(values)
        mine = torch.tensor(values, dtype=torch.float64, device=ca.device)
        all_t = [torch.zeros_like(mine) for _ in range(self.comm_size)]
        torch.distributed.all_gather(all_t, mine, group=bf.get_default_group())
        out = np.array([t.cpu().numpy() for t in all_t])
        return out[:, 0] if out.shape[1] == 1 else out

    def reportBenchTimeColl(self, commsParams, results, lat_across_ranks):
        ca = self.collectiveArgs
        # only the ranks that communicate (comms.py:1116-1123): root + listed ranks of an incast / multicast; every rank
        # otherwise (the reference takes the first group-size ranks there, which under --multi-comms mixes the groups)
        if ca.collective == "multicast":
            lat_across_ranks = np.asarray(lat_across_ranks)[[ca.srcOrDst] + list(ca.dst_ranks)]
        elif ca.collective == "incast":
            lat_across_ranks = np.asarray(lat_across_ranks)[[ca.srcOrDst] + list(ca.src_ranks)]
        p50, p75, p95 = (np.percentile(lat_across_ranks, q) for q in (50, 75, 95))
        mn, mx = np.amin(lat_across_ranks), np.amax(lat_across_ranks)
        _, algBW = comms_utils.getAlgBW(p50 * 1e3, results["memSize"], 1)  # adjusted to the final p50
        busBW = self.backendFuncs.getBusBW(ca.collective, algBW, ca) * (commsParams.bitwidth / 32.0)
        rec = {"collective": ca.collective, "dtype": ca.data_type, "memSize": results["memSize"],
               "numElements": results["numElements"], "p50_us": float(p50), "p75_us": float(p75), "p95_us": float(p95),
               "min_us": float(mn), "max_us": float(mx), "algBW_GBps": float(algBW), "busBW_GBps": float(busBW),
               "world_size": ca.world_size}
        if self.report:
            print(format_row(ca.collective, ca.data_type, self.tag, results["memSize"], results["numElements"],
                             p50, p75, p95, mn, mx, algBW, busBW))
            logger_utils.dispatch(commsParams.use_perf_logger, "comms", logger_utils.commsCollPerfMetrics(
                commsOp=ca.collective, Datatype=ca.data_type, Backend=commsParams.backend, Tags=self.tag, InputSize=results["memSize"],
                OutputSize=results["memSize"], NumElements=results["numElements"], p50_latency_us=float(p50), p75_latency_us=float(p75),
                p95_latency_us=float(p95), min_latency_us=float(mn), max_latency_us=float(mx), AlgoBW_GBs=float(algBW),
                BusBW_GBs=float(busBW)), self.backendFuncs)
        self.results.append(rec)
        return rec

    def reportBenchTimeCollWithQuant(self, commsParams, results, lat, quant_lat, dequant_lat):
        ca = self.collectiveArgs
        p95, quant_p95, dequant_p95 = (float(np.percentile(a, 95)) for a in (lat, quant_lat, dequant_lat))
        rec = {"collective": ca.collective, "dtype": ca.data_type, "memSize": results["memSize"],
               "numElements": results["numElements"], "bitwidth": commsParams.bitwidth, "quant_p95_us": quant_p95,
               "comms_p95_us": p95 - quant_p95 - dequant_p95, "dequant_p95_us": dequant_p95, "p95_us": p95,
               "world_size": ca.world_size}
        if self.report:
            print(format_quant_row(ca.collective, ca.data_type, self.tag, results["memSize"], results["numElements"],
                                   quant_p95, dequant_p95, p95))
            logger_utils.dispatch(commsParams.use_perf_logger, "comms", logger_utils.commsQuantCollPerfMetrics(
                commsOp=ca.collective, Datatype=ca.data_type, Backend=commsParams.backend, Tags=self.tag, InputSize=results["memSize"],
                OutputSize=results["memSize"], NumElements=results["numElements"], p95_latency_us=p95, quant_p95_latency_us=quant_p95,
                dequant_p95_latency_us=dequant_p95, quant_comms_p95_latency_us=p95 - quant_p95 - dequant_p95), self.backendFuncs)
        self.results.append(rec)
        return rec

    def reportBenchTimePt2Pt(self, commsParams, results, across_ranks):
        """percentiles and bandwidth sums over the COMMUNICATING ranks only (comms.py:1188-1283): avg = mean over them,
        total = sum / 2 (every message is counted by its sender and by its receiver)"""
        ca = self.collectiveArgs
        comm = np.asarray(across_ranks)[ca.src_ranks + ca.dst_ranks]
        ping, pingpong = ([float(np.percentile(comm[:, c], q)) for q in (50, 75, 95)] for c in (0, 1))
        avgUni, avgBi = float(np.mean(comm[:, 2])), float(np.mean(comm[:, 3]))
        totUni, totBi = float(np.sum(comm[:, 2]) / 2), float(np.sum(comm[:, 3]) / 2)
        rec = {"collective": "pt2pt", "pattern": ca.pt2pt, "dtype": ca.data_type, "memSize": results["memSize"],
               "ping_p50_us": ping[0], "ping_p75_us": ping[1], "ping_p95_us": ping[2], "pingpong_p50_us": pingpong[0],
               "pingpong_p75_us": pingpong[1], "pingpong_p95_us": pingpong[2], "avgUniBW_GBps": avgUni, "avgBiBW_GBps": avgBi,
               "totalUniBW_GBps": totUni, "totalBiBW_GBps": totBi, "src_ranks": list(ca.src_ranks), "dst_ranks": list(ca.dst_ranks)}
        if self.report:
            # the row is named after the holder's collective field, which the bandwidth tests leave at the LAST point-to-point
            # operation they issued ("recv" on a source rank) -- what the reference prints (comms.py:1245), kept for parsers
            print(format_pt2pt_row(ca.collective, ca.data_type, self.tag, results["memSize"], ping, pingpong, avgUni, avgBi,
                                   totUni, totBi))
            logger_utils.dispatch(commsParams.use_perf_logger, "comms", logger_utils.commsPt2PtPerfMetrics(
                commsOp=ca.collective, Datatype=ca.data_type, Backend=commsParams.backend, Tags=self.tag, InputSize=results["memSize"],
                OutputSize=results["memSize"], NumElements=results["numElements"], p50_latency_us=ping[0], p75_latency_us=ping[1],
                p95_latency_us=ping[2], AvgUniBW_GBs=avgUni, AvgBiBW_GBs=avgBi, TotalUniBW_GBs=totUni, TotalBiBW_GBs=totBi),
                self.backendFuncs)
        self.results.append(rec)
        return rec

    def _start_profiler(self, commsParams):
        """``--size-start-profiler S``: torch.profiler around the run of size S, warm-up iterations skipped, the chrome trace
        written to ``$PARAM_COMMS_PROFILE_DIR`` (default ./comms_profile) per rank"""
        ca = self.collectiveArgs
        active = commsParams.profiler_active_iters or (ca.graph_launches if ca.graph_launches else ca.numIters)
        acts = [torch.profiler.ProfilerActivity.CPU]
        if ca.device.type == "cuda":
            acts.append(torch.profiler.ProfilerActivity.CUDA)
        out_dir = os.environ.get("PARAM_COMMS_PROFILE_DIR", "comms_profile")
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"{ca.collective}_{ca.dataSize}B_rank{ca.global_rank}.json")
        prof = torch.profiler.profile(activities=acts, schedule=torch.profiler.schedule(wait=0, warmup=ca.numWarmupIters, active=active),
                                      on_trace_ready=lambda p: p.export_chrome_trace(path))
        prof.start()
        return prof

    def initCollectiveArgs(self, commsParams):
        """per-collective set-up of the holder (reference initCollectiveArgs, comms.py:826-925): group of this rank, sizes of
        the sweep, root as a GLOBAL rank, the preamble lines"""
        ca, bf = self.collectiveArgs, self.backendFuncs
        groups = bf.get_groups()
        pg = getattr(ca, "pgId", 0)
        self.comm_size = bf.get_world_size()
        ca.groups = groups
        ca.num_pgs = len(groups) if groups else 1
        ca.world_size = bf.get_group_size(groups[pg]) if groups and pg in groups else self.comm_size
        my_ranks = (commsParams.groupRanks or self.groupRanks or {0: list(range(self.comm_size))})[pg]
        local_rank = bf.get_local_rank()
        self.report = ca.global_rank == 0 or (commsParams.enable_local_report and local_rank == 0)
        if commsParams.sizes is not None:
            allSizes = list(commsParams.sizes)
            if self.report:
                logger.info(f"Benchmarking with user-specified message sizes {allSizes}, --b and --e are ignored")
        else:
            comms_utils.fixBeginSize(commsParams, ca.world_size)
            allSizes = comms_utils.getSizes(commsParams.beginSize, commsParams.endSize, commsParams.stepFactor,
                                            commsParams.stepBytes)
        ca.collective = commsParams.collective
        ca.op = bf.get_reduce_op("sum")
        ca.srcOrDst = my_ranks[commsParams.srcOrDst]
        ca.src_ranks = list(commsParams.src_ranks) if commsParams.src_ranks else commsParams.src_ranks
        ca.dst_ranks = list(commsParams.dst_ranks) if commsParams.dst_ranks else commsParams.dst_ranks
        ca.pt2pt, ca.window = commsParams.pt2pt, commsParams.window
        ca.asyncOp = False if commsParams.blockingFlag == 1 else True
        ca.numCollPerIter = commsParams.num_coll
        ca.include_0B = commsParams.include_0B
        ca.graph_launches = commsParams.graph_launches
        ca.numIters, ca.numWarmupIters = commsParams.numIters, commsParams.numWarmupIters
        ca.use_device_time = commsParams.use_device_time
        ca.p2pOps = []
        if commsParams.bitwidth < 32:
            comms_utils.initQuantCommCtx(ca, commsParams)
        ca.group = bf.get_default_group()
        bf.sync_barrier(ca)
        if self.report:
            print(f"[Rank {ca.global_rank:>3}] allSizes: {allSizes} element_size: {commsParams.element_size}"
                  + f" local_rank: {local_rank}, num_pg {ca.num_pgs}, groupSize {ca.world_size}")
        if ca.collective == "pt2pt":
            self.checkPt2PtRanks()
        else:
            self.checkCollectiveRanks()
        ca.comm_dev_time = paramDeviceTimer("comm_timer", bf) if (commsParams.use_device_time and ca.device.type == "cuda") else None
        return allSizes

    def benchComm(self, commsParams):
        ca, bf = self.collectiveArgs, self.backendFuncs
        allSizes = self.initCollectiveArgs(commsParams)
        coll = commsParams.collective
        comm_fn = bf.noop if coll == "pt2pt" else bf.collectiveFunc[coll]
        if self.report:
            print(format_pt2pt_header() if coll == "pt2pt" else
                  format_quant_header() if commsParams.bitwidth < 32 else format_header())
        for curSize in allSizes:
            numElements = self.prepComm(commsParams, curSize)
            ca.group = bf.get_default_group()
            ca.profiler = self._start_profiler(commsParams) if commsParams.size_start_profiler == curSize else None
            if coll == "pt2pt":
                results = self.runPt2Pt()
                mine = [float(np.mean(results["pingPerIterNS"])) / 1e3, float(np.mean(results["pingPongPerIterNS"])) / 1e3,
                        results["avgUniBW"], results["avgBiBW"]]
            elif ca.graph_launches > 0:                                    # comms.py:548-552
                results = self.runCollGraph(comm_fn, dcheck=commsParams.dcheck == 1)
            else:
                results = self.runColl(comm_fn, dcheck=commsParams.dcheck == 1)
            if ca.profiler is not None:
                ca.profiler.stop()
                ca.profiler = None
            results["numElements"] = numElements // ca.world_size if ("all_to_all" in coll or coll in _PER_RANK_ROWS) else numElements
            if commsParams.dcheck == 1:
                self.dcheck(commsParams, curSize)
            if coll == "pt2pt":
                self.reportBenchTimePt2Pt(commsParams, results, self.gatherBenchTime(mine))
            else:
                lat = self.gatherBenchTime(results["timeUS"])
                if commsParams.bitwidth < 32:                # average (de-)quantisation overhead per iteration (comms.py:1387-1396)
                    qlat = self.gatherBenchTime(ca.quant_time.getTimeUS() / ca.numIters)
                    dlat = self.gatherBenchTime(ca.dequant_time.getTimeUS() / ca.numIters)
                    self.reportBenchTimeCollWithQuant(commsParams, results, lat, qlat, dlat)
                else:
                    self.reportBenchTimeColl(commsParams, results, lat)
            bf.clear_memory(ca)
            bf.sync_barrier(ca, desc=f"curSize_{curSize}")
        comms_utils.clearQuantCommCtx(ca)
        bf.sync_barrier(ca, "benchtime")      # rank 0 finishes its report before another collective's preamble

    # ------------------------------------------------------------------ whole run
    def initBackend(self, bootstrap_info, args):
        register()
        cp0 = commsParamsHolder(args, 4, torch.float32, args.collectives[0])
        if args.backend in customized_backend:
            backend_cls, c10d_backend = customized_backend[args.backend], ("gloo" if cp0.device == "cpu" else "nccl")
        else:
            backend_cls, c10d_backend = MI355XBackend, args.backend
        self.backendFuncs = backend_cls(bootstrap_info, cp0)
        os.environ["TORCH_NCCL_ENABLE_TIMING"] = "1" if getattr(args, "enable_torch_nccl_timing", False) else "0"   # comms_utils.py:1944-1947
        self.backendFuncs.initialize_backend(bootstrap_info.master_ip, bootstrap_info.master_port, backend=c10d_backend,
                                             eager_mode=bool(cp0.init_only or cp0.eager_init))
        self.c10d_backend = c10d_backend
        return self.backendFuncs

    def runBench(self, args):
        bf, ca = self.backendFuncs, self.collectiveArgs
        ca.device = bf.get_device()
        ca.world_size = bf.get_world_size()
        ca.global_rank = bf.get_global_rank()
        ca.group = bf.get_default_group()
        ca.backendFuncs = bf
        self.comm_size = ca.world_size
        self.checkArgsWithBackend(args)
        groupRanks = self.genMultiCommGroups(getattr(args, "multi_comms", 1), getattr(self, "c10d_backend", args.backend))
        ca.groups = bf.get_groups()
        if getattr(args, "init_only", False):
            return self.results
        for dname in args.dtypes:
            dtype = _DTYPES[dname]
            ca.data_type = dname
            esz = torch.empty(0, dtype=dtype).element_size()
            if getattr(args, "sb", 0) % esz != 0:                           # comms.py:366-368
                logger.error("Step size bytes must be a multiple of element size")
                comms_utils.gracefulExit()
            b, e = args.b, args.e
            if getattr(args, "split_elements", None) is not None:          # --i / --o fix the one size of the run
                args.b = args.e = args.split_elements * esz
                logger.warning(f"Overwriting begin-size (--b {b}) with {args.b} and end-size (--e {e}) with {args.e} to match "
                               f"requested input-split (--i) {args.i} or output-split (--o) {args.o}")
            for coll in args.collectives:
                cp = commsParamsHolder(args, esz, dtype, coll, groupRanks)
                bf.commsParams = cp
                bf.benchmark_comms(lambda idx, p, b: self.benchComm(p), cp)
            args.b, args.e = b, e
        return self.results


def main(argv=None):
    bench = commsCollBench()
    parser = argparse.ArgumentParser(description="PARAM-Comm collective benchmark (MI355X / RCCL over xGMI build)")
    args = bench.readArgs(parser) if argv is None else None
    if argv is not None:
        import sys

        old = sys.argv
        sys.argv = ["comms.py"] + list(argv)
        try:
            args = bench.readArgs(parser)
        finally:
            sys.argv = old
    comms_utils.init_logging(args.log)
    env = comms_utils.read_comms_env_vars()
    if env["global_rank"] == 0 or (args.enable_local_report and env["local_rank"] == 0):       # comms.py:1559-1575
        print("\t PARAM COMM environment: %s " % (str(env)))
        print("\t backend: %s nw-stack: %s args.data_types: %s args.b: %s args.e: %s args.f: %s args.z: %s args.master_ip: %s "
              % (args.backend, args.nw_stack, [d for d in args.data_types.split(",") if d], args.b, args.e, args.f, args.z,
                 args.master_ip))
    bench.checkArgs(args)
    if env["world_size"] < 1:
        env = {"world_size": 1, "local_size": 1, "global_rank": 0, "local_rank": 0}
    if env["local_size"] < 1:
        env["local_size"] = env["world_size"]
    if env["local_rank"] < 0:
        env["local_rank"] = env["global_rank"] % max(1, env["local_size"])
    info = comms_utils.bootstrap_info_holder(args.master_ip, args.master_port, 0, env)
    bf = bench.initBackend(info, args)
    bf.sayHello()  # the reference's call form (comms.py:1533)
    try:
        return bench.runBench(args)
    finally:
        bf.shutdown()


if __name__ == "__main__":
    main()  # pragma: no cover
