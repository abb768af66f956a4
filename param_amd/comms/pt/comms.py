"""comms.py -- nccl-tests style collective sweep for the all-to-all family (and all_reduce),
reporting p50/p75/p95 latency, AlgBW and BusBW per message size.

Own restatement of the part of reference ``train/comms/pt/comms.py`` that produces the
all-to-all bus-BW metric (benchComm ``:1285-1429``, run_coll_non_graph ``:452-545``,
reportBenchTimeColl ``:1112-1186``, CLI ``:50-206`` + comms_utils.py ``:1713-1879``):

  mpirun/torchrun -np N python -m param_amd.comms.pt.comms --master-ip 127.0.0.1 --b 8 --e 256M \
        --n 100 --f 2 --z 1 --collective all_to_all --backend rccl_xgmi --device rocm

Kept: flag names/defaults used by the reference's README example, rank discovery from the
launcher's environment, per-size tensor preparation with equal splits ``numElements // world``
(comms_utils.py:1115-1124,1212-1217), blocking (``--z 1``: barrier + wait + device sync per
iteration) vs non-blocking timing, ``--c 1`` self-check, the report recomputing AlgBW from the
p50 of per-rank mean latencies, busBW = algBW * (n-1)/n for all_to_all*, and the row format.
"""
from __future__ import annotations

import argparse
import logging
import time

import numpy as np
import torch

from . import comms_utils
from .comms_utils import paramDeviceTimer, paramStreamGuard
from .mi355_backend import BACKEND_NAME, MI355XBackend, register
from .pytorch_backend_utils import collectiveArgsHolder, customized_backend, supportedCollectives

logger = logging.getLogger(__name__)

_DTYPES = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16, "int32": torch.int32,
           "long": torch.long, "float64": torch.float64, "int8": torch.int8}

HEADER_FMT = "{:>40}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}{:>18}"
QUANT_HEADER_FMT = "-QUANT\t{:>40}{:>18}{:>25}{:>15}{:>15}{:>15}"
QUANT_ROW_FMT = "\tCOMMS-RES-QUANT-{}-{}{}\t{:>15}{:>18}{:>25}{:>15}{:>15}{:>15}"
ROW_FMT = "\tCOMMS-RES-{}-{}{}{:>18}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}{:>20}"


class commsParamsHolder:
    """Run parameters (subset of the reference's commsParamsHolderBase/commsParamsHolder, comms_utils.py:801-928)."""

    def __init__(self, args, element_size: int, dtype, collective: str):
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
        self.init_only = False
        self.use_device_time = args.use_device_time
        self.include_0B = args.include_0B
        self.graph_launches = getattr(args, "graph_launches", 0)
        self.init_method = None
        self.use_ext_dist = False


def format_header() -> str:
    """the reference's preamble line (``printPreamble``, comms.py:956-1001): plain ``COMMS-RES`` + the column titles"""
    return "\n\tCOMMS-RES" + HEADER_FMT.format(
        "total-size (B)", "nElementsPerRank", "Time(us):p50", "p75", "p95", "Min", "Max", "AlgBW(GB/s)",
        "BusBW(GB/s)", "TotalTime(us):p50")


def format_quant_header() -> str:
    """the ``--bitwidth < 32`` preamble (comms.py:976-986); ``str.format`` drops the reference's surplus seventh title"""
    return "\n\tCOMMS-RES" + QUANT_HEADER_FMT.format("size (B)", "nElementsPerRank", "P95 Latency(us): Quant", "Comms",
                                                      "De-Quant", "Overall", "TotalLatency(us):p50")


def format_quant_row(collective, data_type, tag, memSize, numElements, quant_p95, dequant_p95, p95):
    """``reportBenchTimeCollWithQuant`` (comms.py:1005-1040): comms = overall p95 - quant p95 - de-quant p95"""
    return QUANT_ROW_FMT.format(collective, data_type, tag, memSize, "%d" % numElements, "%.1f" % quant_p95,
                                "%.1f" % (p95 - quant_p95 - dequant_p95), "%.1f" % dequant_p95, "%.1f" % p95)


def format_row(collective, data_type, tag, memSize, numElements, p50, p75, p95, mn, mx, algBW, busBW, total_p50=0.0):
    return ROW_FMT.format(collective, data_type, tag, memSize, "%d" % numElements, "%.1f" % p50, "%.1f" % p75,
                          "%.1f" % p95, "%.1f" % mn, "%.1f" % mx, "%.3f" % algBW, "%.3f" % busBW, "%.1f" % total_p50)


class commsCollBench:
    def __init__(self):
        self.collectiveArgs = collectiveArgsHolder()
        self.backendFuncs = None
        self.tag = ""
        self.initVal = 1
        self.results = []

    # ------------------------------------------------------------------ args
    def readArgs(self, parser: argparse.ArgumentParser):
        parser.add_argument("--master-ip", type=str, default="127.0.0.1")
        parser.add_argument("--master-port", type=str, default="29500")
        parser.add_argument("--backend", type=str, default=BACKEND_NAME, help="rccl_xgmi | nccl | gloo")
        parser.add_argument("--nw-stack", type=str, default="pytorch-dist")
        parser.add_argument("--device", type=str, default="rocm", choices=["cuda", "rocm", "cpu"])
        parser.add_argument("--w", "--warmup-iters", type=int, default=5, dest="w")
        parser.add_argument("--n", "--num-iters", type=int, default=5, dest="n")
        parser.add_argument("--b", "--begin-size", type=str, default="8", dest="b")
        parser.add_argument("--e", "--end-size", type=str, default="8", dest="e")
        parser.add_argument("--f", "--step-factor", type=int, default=2, dest="f")
        parser.add_argument("--sb", "--step-bytes", type=int, default=0, dest="sb")
        parser.add_argument("--z", "--blocking", type=int, default=1, dest="z")
        parser.add_argument("--c", "--check", type=int, default=0, dest="c")
        parser.add_argument("--bitwidth", type=int, default=32, choices=[2, 4, 8, 16, 32], help="Quantization bitwidth")
        parser.add_argument("--quant-a2a-embedding-dim", type=int, default=32, choices=[32, 64, 128, 256],
                            help="Embedding dimension used by quantization alltoall if enabled")
        parser.add_argument("--quant-threshold", type=int, default=33554432,
                            help="threshold of message sizes to perform quantization if enabled")
        parser.add_argument("--collective", type=str, default="all_to_all")
        parser.add_argument("--data-types", "--dtype", type=str, default="float32", dest="data_types")
        parser.add_argument("--use-device-time", action="store_true", default=False)
        parser.add_argument("--include-0B", action="store_true", default=False)
        parser.add_argument("--graph-launches", type=int, default=0, help="Number of graph launches for each data-size")
        parser.add_argument("--log", type=str, default="ERROR")
        return parser.parse_args()

    def checkArgs(self, args):
        args.b = comms_utils.parsesize(args.b)
        args.e = comms_utils.parsesize(args.e)
        args.collectives = [c.strip() for c in args.collective.split(",")]
        for c in args.collectives:
            if c not in supportedCollectives:
                logger.error(f"Specified collective: {c} is not one of the supported collectives: {supportedCollectives}")
                comms_utils.gracefulExit()
        args.dtypes = [d.strip() for d in args.data_types.split(",")]
        for d in args.dtypes:
            if d not in _DTYPES:
                logger.error(f"Specified dtype: {d} is not one of the supported commstyle: {list(_DTYPES)}")
                comms_utils.gracefulExit()
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
        if args.c == 1 and args.z == 0:
            logger.warning("data validation requires blocking mode: forcing --z 1")
            args.z = 1

    # ------------------------------------------------------------------ one collective
    def prepComm(self, commsParams, size_bytes: int):
        ca = self.collectiveArgs
        world = ca.world_size
        dev = ca.device
        numElements = max(size_bytes // commsParams.element_size, 1)
        scale = world
        if commsParams.collective in ("all_to_all", "all_to_allv", "all_to_all_single"):
            used, split = comms_utils.equal_splits(numElements, world)
            used = max(used, world)
            per = used // world
            if commsParams.dcheck == 1:
                ip = self.backendFuncs.alloc_ones([used], dev, commsParams.dtype, self.initVal)
            else:
                ip = self.backendFuncs.alloc_random([used], dev, commsParams.dtype, scale)
            op = self.backendFuncs.alloc_random([used], dev, commsParams.dtype, scale)
            if commsParams.collective == "all_to_all":  # list form: one tensor per peer
                ca.ipTensor = list(ip.split(per))
                ca.opTensor = list(op.split(per))
                ca.ipTensor_split, ca.opTensor_split = [], []
            else:
                ca.ipTensor, ca.opTensor = ip, op
                if commsParams.include_0B and world > 1:
                    ins = {i: [0] * world for i in range(world)}
                    for i in range(world):
                        for j in range(world):
                            if j != (i + 1) % world:
                                ins[i][j] = used // (world - 1)
                        ins[i][i] += used % (world - 1)
                    ca.ipTensor_split = ins[ca.global_rank]
                    ca.opTensor_split = [ins[i][ca.global_rank] for i in range(world)]
                    ca.opTensor = self.backendFuncs.alloc_random([sum(ca.opTensor_split)], dev, commsParams.dtype, scale)
                else:
                    ca.ipTensor_split = [per] * world
                    ca.opTensor_split = [per] * world
            numElements = used
        else:  # all_reduce / reduce: in place on ipTensor
            if commsParams.dcheck == 1:
                ca.ipTensor = self.backendFuncs.alloc_ones([numElements], dev, commsParams.dtype, self.initVal)
            else:
                ca.ipTensor = self.backendFuncs.alloc_random([numElements], dev, commsParams.dtype, scale)
            ca.opTensor = ca.ipTensor
        ca.dataSize = numElements * commsParams.element_size
        ca.numElements = numElements
        return numElements

    def runColl(self, comm_fn, dcheck=False):
        """Timing protocol of the reference's run_coll_non_graph (comms.py:452-545)."""
        ca, bf = self.collectiveArgs, self.backendFuncs
        bf.sync_barrier(ca, desc="runColl_begin")
        elapsed_ns = 0.0
        is_blocking = not ca.asyncOp
        dev_timer = getattr(ca, "comm_dev_time", None)
        for it in range(ca.numWarmupIters + ca.numIters):
            if it == ca.numWarmupIters:
                bf.complete_accel_ops(ca)
                elapsed_ns = 0.0
                if dev_timer:
                    dev_timer.reset()
                ca.quant_time.reset()
                ca.dequant_time.reset()
            if dcheck and ca.collective in ("all_reduce", "reduce"):
                ca.ipTensor.fill_(self.initVal)  # in-place reductions: reset before every iteration (comms.py:474-476)
            if is_blocking:
                bf.sync_barrier(ca)
            start = time.monotonic()
            with paramStreamGuard(stream=bf.get_current_stream(device=ca.device), curDevice=ca.device,
                                  backendFuncs=bf, is_blocking=False, timer=dev_timer):
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
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(ca.numWarmupIters):
                comm_fn(ca)
        torch.cuda.current_stream().wait_stream(side)
        bf.complete_accel_ops(ca)            # nothing in flight that c10d's watchdog thread would query while the capture is open
        ca.asyncOp = False
        graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: the process group's watchdog thread polls its events with hipEventQuery, which a GLOBAL-mode
        # capture on this thread turns into an error (and an abort) in that thread
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(ca.numIters):
                if dcheck and ca.collective in ("all_reduce", "reduce"):
                    ca.ipTensor.fill_(self.initVal)              # reset inside the graph: every replay validates (comms.py:399-401)
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

    def dcheck(self, commsParams, curSize):
        """``--c 1``: inputs are ones, so all_to_all* outputs are ones and all_reduce gives world_size
        (comms_utils.py:997-1055)."""
        ca = self.collectiveArgs
        expect = self.initVal * (ca.world_size if ca.collective in ("all_reduce",) else 1)
        tensors = ca.opTensor if isinstance(ca.opTensor, (list, tuple)) else [ca.opTensor]
        for t in tensors:
            if t.numel() and not bool((t == expect).all()):
                bad = int((t != expect).sum())
                raise ValueError(f"[{ca.global_rank}] {ca.collective}: {bad} elements differ from {expect} at size {curSize}")

    def gatherBenchTime(self, timeUS: float):
        ca, bf = self.collectiveArgs, self.backendFuncs
        mine = torch.tensor([timeUS], dtype=torch.float64, device=ca.device)
        all_t = [torch.zeros_like(mine) for _ in range(ca.world_size)]
        torch.distributed.all_gather(all_t, mine, group=bf.get_default_group())
        return np.array([float(t.item()) for t in all_t])

    def reportBenchTimeColl(self, commsParams, results, lat_across_ranks):
        ca = self.collectiveArgs
        p50, p75, p95 = (np.percentile(lat_across_ranks, q) for q in (50, 75, 95))
        mn, mx = np.amin(lat_across_ranks), np.amax(lat_across_ranks)
        _, algBW = comms_utils.getAlgBW(p50 * 1e3, results["memSize"], 1)  # adjusted to the final p50
        busBW = self.backendFuncs.getBusBW(ca.collective, algBW, ca) * (commsParams.bitwidth / 32.0)
        rec = {"collective": ca.collective, "dtype": ca.data_type, "memSize": results["memSize"],
               "numElements": results["numElements"], "p50_us": float(p50), "p75_us": float(p75), "p95_us": float(p95),
               "min_us": float(mn), "max_us": float(mx), "algBW_GBps": float(algBW), "busBW_GBps": float(busBW),
               "world_size": ca.world_size}
        if ca.global_rank == 0:
            print(format_row(ca.collective, ca.data_type, self.tag, results["memSize"], results["numElements"],
                             p50, p75, p95, mn, mx, algBW, busBW))
        self.results.append(rec)
        return rec

    def reportBenchTimeCollWithQuant(self, commsParams, results, lat, quant_lat, dequant_lat):
        ca = self.collectiveArgs
        p95, quant_p95, dequant_p95 = (float(np.percentile(a, 95)) for a in (lat, quant_lat, dequant_lat))
        rec = {"collective": ca.collective, "dtype": ca.data_type, "memSize": results["memSize"],
               "numElements": results["numElements"], "bitwidth": commsParams.bitwidth, "quant_p95_us": quant_p95,
               "comms_p95_us": p95 - quant_p95 - dequant_p95, "dequant_p95_us": dequant_p95, "p95_us": p95,
               "world_size": ca.world_size}
        if ca.global_rank == 0:
            print(format_quant_row(ca.collective, ca.data_type, self.tag, results["memSize"], results["numElements"],
                                   quant_p95, dequant_p95, p95))
        self.results.append(rec)
        return rec

    def benchComm(self, commsParams):
        ca, bf = self.collectiveArgs, self.backendFuncs
        ca.collective = commsParams.collective
        ca.asyncOp = False if commsParams.blockingFlag == 1 else True
        ca.numCollPerIter = 1
        ca.graph_launches = commsParams.graph_launches
        ca.numIters, ca.numWarmupIters = commsParams.numIters, commsParams.numWarmupIters
        ca.use_device_time = commsParams.use_device_time
        ca.comm_dev_time = paramDeviceTimer("comm_timer", bf) if (commsParams.use_device_time and ca.device.type == "cuda") else None
        comm_fn = bf.collectiveFunc[commsParams.collective]
        comms_utils.fixBeginSize(commsParams, ca.world_size)
        if commsParams.bitwidth < 32:
            comms_utils.initQuantCommCtx(ca, commsParams)
        if ca.global_rank == 0:
            print(format_quant_header() if commsParams.bitwidth < 32 else format_header())
        for curSize in comms_utils.getSizes(commsParams.beginSize, commsParams.endSize, commsParams.stepFactor,
                                            commsParams.stepBytes):
            numElements = self.prepComm(commsParams, curSize)
            ca.group = bf.get_default_group()
            if ca.graph_launches > 0:                                    # comms.py:548-552
                results = self.runCollGraph(comm_fn, dcheck=commsParams.dcheck == 1)
            else:
                results = self.runColl(comm_fn, dcheck=commsParams.dcheck == 1)
            results["numElements"] = numElements // ca.world_size if "all_to_all" in ca.collective else numElements
            if commsParams.dcheck == 1:
                self.dcheck(commsParams, curSize)
            lat = self.gatherBenchTime(results["timeUS"])
            if commsParams.bitwidth < 32:                    # average (de-)quantisation overhead per iteration (comms.py:1387-1396)
                qlat = self.gatherBenchTime(ca.quant_time.getTimeUS() / ca.numIters)
                dlat = self.gatherBenchTime(ca.dequant_time.getTimeUS() / ca.numIters)
                self.reportBenchTimeCollWithQuant(commsParams, results, lat, qlat, dlat)
            else:
                self.reportBenchTimeColl(commsParams, results, lat)
            bf.clear_memory(ca)
        comms_utils.clearQuantCommCtx(ca)

    # ------------------------------------------------------------------ whole run
    def initBackend(self, bootstrap_info, args):
        register()
        cp0 = commsParamsHolder(args, 4, torch.float32, args.collectives[0])
        if args.backend in customized_backend:
            backend_cls, c10d_backend = customized_backend[args.backend], ("gloo" if cp0.device == "cpu" else "nccl")
        else:
            backend_cls, c10d_backend = MI355XBackend, args.backend
        self.backendFuncs = backend_cls(bootstrap_info, cp0)
        self.backendFuncs.initialize_backend(bootstrap_info.master_ip, bootstrap_info.master_port, backend=c10d_backend)
        return self.backendFuncs

    def runBench(self, args):
        bf, ca = self.backendFuncs, self.collectiveArgs
        ca.device = bf.get_device()
        ca.world_size = bf.get_world_size()
        ca.global_rank = bf.get_global_rank()
        ca.group = bf.get_default_group()
        ca.groups = bf.get_groups()
        ca.backendFuncs = bf
        for dname in args.dtypes:
            dtype = _DTYPES[dname]
            ca.data_type = dname
            for coll in args.collectives:
                cp = commsParamsHolder(args, torch.empty(0, dtype=dtype).element_size(), dtype, coll)
                bf.commsParams = cp
                bf.benchmark_comms(lambda idx, p, b: self.benchComm(p), cp)
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
    logging.basicConfig(level=getattr(logging, args.log.upper(), logging.ERROR))
    bench.checkArgs(args)
    env = comms_utils.read_comms_env_vars()
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
