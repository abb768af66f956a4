"""commsComputeBench.py -- a collective on the current stream OVERLAPPED with the embedding-lookup
compute kernel on a second HIP stream, each timed with its own device events.

Own restatement of reference ``train/comms/pt/commsComputeBench.py`` for the one compute kernel on the
hot path (``--kernel emb_lookup``): flags ``:37-136``, ``runColl`` ``:155-257`` (comm under a stream guard
with ``comm_dev_time``, compute under a guard on ``compute_stream`` with ``compute_dev_time``, barrier per
iteration, host elapsed + per-stream device time), ``initCollectiveArgs`` ``:259-360`` (stream ``:275``,
embedding setup ``:303-312`` -> ``init_emb_lookup`` comms_utils.py:1956-2039).

The reference's emb_lookup mode is broken at HEAD (SURVEY bug R5: ``commsParams.direction`` and the
``emb_dim/num_embs/...`` attributes are never set on this path); here the flags feed the kernel directly:
``--ntables`` tables of ``--num-embs`` x ``--emb-dim`` per device, grouped ``--num-emb-tables-batched`` per
op (default: all in one op), requests in the fbgemm layout (indices ``[T*B*L]``, offsets ``[T*B+1]``,
uniform ids), ``--direction forward|backward``.
"""
from __future__ import annotations

import argparse
import logging
import time

import torch

from . import comms_utils
from .comms import commsCollBench, commsParamsHolder, _DTYPES
from .comms_utils import paramDeviceTimer, paramStreamGuard

logger = logging.getLogger(__name__)


CC_HEADER_FMT = "{:>40}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}"
CC_QUANT_HEADER_FMT = "-QUANT\t{:>40}{:>18}{:>25}{:>15}{:>15}{:>15}"
CC_ROW_FMT = "\tCOMMS-RES-{}-{}{}{:>18}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}"
CC_DEV_TIME_FMT = "{:>20}{:>20}"        # TotalLatency(us):p50, CompLatency(us):p50 -- comms-compute mode only


def format_cc_header(mode: str, bitwidth: int = 32) -> str:
    """the reference's preamble (commsComputeBench.py:361-433): the collective sweep's titles with ``Latency(us):p50`` for the
    collective's DEVICE time and, in comms-compute mode, two more columns: host time per iteration and the compute kernel's device
    time (``str.format`` drops the surplus titles, as in the reference)"""
    dev = CC_DEV_TIME_FMT if mode == "comms-compute" else ""
    if bitwidth < 32:
        return "\n\tCOMMS-RES" + (CC_QUANT_HEADER_FMT + dev).format(
            "size (B)", "nElementsPerRank", "P95 Latency(us): Quant", "Comms", "De-Quant", "Overall", "TotalLatency(us):p50",
            "CompLatency(us):p50", "TFlops")
    return "\n\tCOMMS-RES" + (CC_HEADER_FMT + dev).format(
        "total-size (B)", "nElementsPerRank", "Latency(us):p50", "p75", "p95", "Min", "Max", "AlgBW(GB/s)", "BusBW(GB/s)",
        "TotalLatency(us):p50", "CompLatency(us):p50", "TFlops")


def cc_report(collective, world_size, memSize, lat_us, comm_us, comp_us, getBusBW, bitwidth: int = 32):
    """the numbers of one report row (commsComputeBench.py:499-577): with device times, the percentiles are the collective's
    device time across ranks, ``total_p50`` the host time per iteration and ``compute_p50`` the compute kernel's device time;
    without (compute-only mode), the host time alone.  AlgBW is recomputed from the final p50."""
    import numpy as np

    lat = np.asarray(lat_us, dtype=np.float64)
    total_p50 = compute_p50 = 0.0
    if len(comp_us):
        compute_p50 = float(np.percentile(np.asarray(comp_us, dtype=np.float64), 50))
    if len(comm_us):
        total_p50 = float(np.percentile(lat, 50))
        lat = np.asarray(comm_us, dtype=np.float64)
    p50, p75, p95 = (float(np.percentile(lat, q)) for q in (50, 75, 95))
    _, algBW = comms_utils.getAlgBW(p50 * 1e3, memSize, 1)
    busBW = getBusBW(collective, algBW, world_size) * (bitwidth / 32.0)
    return {"p50": p50, "p75": p75, "p95": p95, "min": float(lat.min()), "max": float(lat.max()), "algBW": algBW, "busBW": busBW,
            "total_p50": total_p50, "compute_p50": compute_p50}


def format_cc_row(collective, data_type, tag, memSize, numElements, rep: dict, mode: str) -> str:
    dev = CC_DEV_TIME_FMT if mode == "comms-compute" else ""
    return (CC_ROW_FMT + dev).format(collective, data_type, tag, memSize, "%d" % numElements, "%.1f" % rep["p50"], "%.1f" % rep["p75"],
                                     "%.1f" % rep["p95"], "%.1f" % rep["min"], "%.1f" % rep["max"], "%.3f" % rep["algBW"],
                                     "%.3f" % rep["busBW"], "%.1f" % rep["total_p50"], "%.1f" % rep["compute_p50"])


class commsComputeBench(commsCollBench):
    def readArgs(self, parser):
        parser.add_argument("--mode", type=str, default="comms-compute", choices=["compute", "comms-compute"])
        parser.add_argument("--kernel", type=str, default="emb_lookup", choices=["emb_lookup"])
        parser.add_argument("--num-compute", "--num-compute-per-iteration", type=int, default=100, dest="num_compute")
        parser.add_argument("--emb-dim", type=int, default=128)
        parser.add_argument("--num-embs", type=int, default=100000)
        parser.add_argument("--batch-size", type=int, default=512)
        parser.add_argument("--num-emb-tables-per-device", "--ntables", "--num-emb-tables", type=int, default=8, dest="ntables")
        parser.add_argument("--num-emb-tables-batched", type=int, default=-1)
        parser.add_argument("--bag-size", type=int, default=20)
        parser.add_argument("--direction", type=str, default="forward", choices=["forward", "backward"])
        return super().readArgs(parser)

    def init_emb_lookup(self, args):
        """tables + requests of the compute kernel (reference init_emb_lookup, comms_utils.py:1956-2039)"""
        args.num_emb_tables_per_device = args.ntables
        comms_utils.init_emb_lookup(self.collectiveArgs, args, self.backendFuncs)

    def runColl(self, comm_fn=None, compute_fn=None, dcheck=False):
        ca, bf = self.collectiveArgs, self.backendFuncs
        bf.sync_barrier(ca, desc="runColl_begin")
        elapsed_ns = 0.0
        enable_comms = comm_fn is not None and comm_fn != bf.noop
        for it in range(ca.numWarmupIters + ca.numIters):
            if it == ca.numWarmupIters:
                bf.complete_accel_ops(ca)
                elapsed_ns = 0.0
                for t in (ca.comm_dev_time, ca.compute_dev_time):
                    if t:
                        t.reset()
                ca.quant_time.reset()
                ca.dequant_time.reset()
            start = time.monotonic()
            with paramStreamGuard(stream=bf.get_current_stream(device=ca.device), curDevice=ca.device,
                                  backendFuncs=bf, timer=ca.comm_dev_time, is_blocking=False):
                if enable_comms:
                    for _ in range(ca.numCollPerIter):
                        comm_fn(ca)
                    bf.complete_accel_ops(ca, devSync=False)   # join async work so the end event covers it
            with paramStreamGuard(stream=ca.compute_stream, curDevice=ca.device, backendFuncs=bf,
                                  timer=ca.compute_dev_time, is_blocking=False):
                for _ in range(ca.numComputePerIter):
                    compute_fn(ca)
            bf.sync_barrier(ca, desc="runColl_sync")
            elapsed_ns += (time.monotonic() - start) * 1e9
            for t in (ca.comm_dev_time, ca.compute_dev_time):
                if t:
                    t.elapsedTime()
        memSize = bf.get_mem_size(ca) if enable_comms else 0
        avgIterNS, algBW = comms_utils.getAlgBW(elapsed_ns, memSize, ca.numIters * max(1, ca.numCollPerIter))
        busBW = bf.getBusBW(ca.collective, algBW, ca) if enable_comms else 0.0
        ca.group = bf.get_default_group()
        bf.sync_barrier(ca, desc="runColl_end")
        res = {"timeUS": avgIterNS / 1e3, "algBW": algBW, "busBW": busBW, "memSize": memSize}
        if ca.comm_dev_time:
            res["comm_dev_us"] = ca.comm_dev_time.elapsedTimeNS / 1e3 / ca.numIters
        if ca.compute_dev_time:
            res["compute_dev_us"] = ca.compute_dev_time.elapsedTimeNS / 1e3 / ca.numIters
        return res

    def benchComm(self, commsParams, args):
        ca, bf = self.collectiveArgs, self.backendFuncs
        ca.collective = commsParams.collective
        ca.asyncOp = True                       # comm must not block the host: the compute launches follow it
        ca.numCollPerIter = args.num_coll if args.mode == "comms-compute" else 0
        ca.numComputePerIter = args.num_compute
        ca.numIters, ca.numWarmupIters = commsParams.numIters, commsParams.numWarmupIters
        gpu = ca.device.type == "cuda"
        ca.compute_stream = bf.get_new_stream()
        ca.comm_dev_time = paramDeviceTimer("comm_timer", bf) if gpu else None
        ca.compute_dev_time = paramDeviceTimer("compute_timer", bf) if gpu else None
        self.init_emb_lookup(args)
        compute_fn = bf.computeFunc[args.kernel]
        comm_fn = bf.collectiveFunc[commsParams.collective] if args.mode == "comms-compute" else None
        comms_utils.fixBeginSize(commsParams, ca.world_size)
        if commsParams.bitwidth < 32 and comm_fn is not None:      # --bitwidth (reference commsComputeBench.py:395,727-735)
            comms_utils.initQuantCommCtx(ca, commsParams)
        lookups = args.ntables * args.batch_size * args.bag_size * args.num_compute
        out = []
        self.comm_size = ca.world_size
        if ca.global_rank == 0:            # the reference's preamble lines (commsComputeBench.py:308-311, 361-433)
            print(f"[Rank {ca.global_rank:>3}] mode: {args.mode}, num_coll: {args.num_coll}, kernel: {args.kernel}, num_compute {args.num_compute}, "
                  f"emb_dim {args.emb_dim}, num_embs {args.num_embs}, batch_size {args.batch_size}")
            print(format_cc_header(args.mode, commsParams.bitwidth if comm_fn is not None else 32))
        for curSize in comms_utils.getSizes(commsParams.beginSize, commsParams.endSize, commsParams.stepFactor,
                                            commsParams.stepBytes):
            self.prepComm(commsParams, curSize)
            ca.group = bf.get_default_group()
            r = self.runColl(comm_fn, compute_fn)
            r.update({"size": curSize, "kernel": args.kernel, "direction": args.direction,
                      "lookups_per_iter": lookups,
                      "overlap_efficiency": (max(r.get("comm_dev_us", 0), r.get("compute_dev_us", 0)) / r["timeUS"]) if r["timeUS"] else 0})
            if r.get("compute_dev_us"):
                r["lookups_per_s_compute_stream"] = lookups / (r["compute_dev_us"] * 1e-6)
            if commsParams.bitwidth < 32 and comm_fn is not None:
                r.update({"bitwidth": commsParams.bitwidth, "quant_us": ca.quant_time.getTimeUS() / ca.numIters,
                          "dequant_us": ca.dequant_time.getTimeUS() / ca.numIters})
            # the reference's row: every rank's host time / collective device time / compute device time gathered, percentiles
            # across ranks (commsComputeBench.py:682-700, 499-577)
            across = self.gatherBenchTime([r["timeUS"], r.get("comm_dev_us", 0.0), r.get("compute_dev_us", 0.0)])
            across = across.reshape(-1, 3)
            with_dev = args.mode == "comms-compute" and "comm_dev_us" in r
            rep = cc_report(ca.collective, ca.world_size, r["memSize"], across[:, 0], across[:, 1] if with_dev else [],
                            across[:, 2] if ("compute_dev_us" in r and args.mode == "comms-compute") else [],
                            lambda c, bw, n: bf.getBusBW(c, bw, ca), commsParams.bitwidth if comm_fn is not None else 32)
            r["report"] = rep
            n_el = ca.numElements // ca.world_size if "all_to_all" in ca.collective else ca.numElements
            if ca.global_rank == 0:
                print(format_cc_row(ca.collective, ca.data_type, self.tag, r["memSize"], n_el, rep, args.mode))
            if ca.global_rank == 0:
                print("\tCOMMS-COMPUTE-RES-{}-{}  size {:>12}  iter {:>10.1f} us  comm(dev) {:>10.1f} us  compute(dev) {:>10.1f} us"
                      "  algBW {:>8.3f}  busBW {:>8.3f} GB/s".format(ca.collective, args.kernel, curSize, r["timeUS"],
                                                                     r.get("comm_dev_us", 0.0), r.get("compute_dev_us", 0.0),
                                                                     r["algBW"], r["busBW"]))
            out.append(r)
            bf.clear_memory(ca)
        comms_utils.clearQuantCommCtx(ca)
        return out

    def runBench(self, args):
        bf, ca = self.backendFuncs, self.collectiveArgs
        ca.device, ca.world_size, ca.global_rank = bf.get_device(), bf.get_world_size(), bf.get_global_rank()
        ca.group, ca.groups, ca.backendFuncs = bf.get_default_group(), bf.get_groups(), bf
        dname = args.dtypes[0]
        ca.data_type = dname
        cp = commsParamsHolder(args, torch.empty(0, dtype=_DTYPES[dname]).element_size(), _DTYPES[dname], args.collectives[0])
        bf.commsParams = cp
        return self.benchComm(cp, args)


def main(argv=None):
    bench = commsComputeBench()
    parser = argparse.ArgumentParser(description="comms + embedding-lookup overlap benchmark (MI355X build)")
    import sys

    old = sys.argv
    if argv is not None:
        sys.argv = ["commsComputeBench.py"] + list(argv)
    try:
        args = bench.readArgs(parser)
    finally:
        sys.argv = old
    comms_utils.init_logging(args.log)
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
    try:
        return bench.runBench(args)
    finally:
        bf.shutdown()


if __name__ == "__main__":
    main()  # pragma: no cover
