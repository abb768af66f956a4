"""Harness arithmetic and bootstrap helpers of the comms benchmarks -- own restatement of the
parts of reference ``train/comms/pt/comms_utils.py`` the all-to-all path uses:

  parsesize ``:99-140``          nccl-tests style "8", "4K", "256M", "1G"
  parseRankList ``:143-165``
  getAlgBW ``:168-186``          bytes / ns == GB/s
  getSizes ``:189-215``          --b .. --e stepped by --f (or --sb)
  fixBeginSize ``:218-252``      at least one element per rank for all_to_all*
  read_comms_env_vars ``:293-351``  rank discovery (OpenMPI / MVAPICH / PMI / torchrun / SLURM)
  bootstrap_info_holder ``:781-798``
  paramStreamGuard / paramDeviceTimer ``:717-778``  per-stream device-event timing
  _prep_all_to_all_single / _prep_all_to_allv ``:1093-1219``  tensors + equal splits per size

Pinned against the reference by tests/golden/comms_pure.json (generated in the build container).
"""
from __future__ import annotations

import logging
import os
import sys
import time
from collections import OrderedDict
from contextlib import ContextDecorator

logger = logging.getLogger(__name__)


def gracefulExit(args: int = 0) -> None:
    """Fatal configuration error: the reference's ``gracefulExit`` is ``sys.exit`` (``:83-96``)."""
    sys.exit(args)


def parsesize(ipValue) -> int:
    if isinstance(ipValue, int) or str(ipValue).isnumeric():
        return int(ipValue)
    s = str(ipValue)
    for suffix, unit in (("G", 1 << 30), ("M", 1 << 20), ("K", 1 << 10)):
        pos = s.find(suffix)
        if pos != -1:
            return int(s[:pos]) * unit
    logger.error(f"Could not parse input size {ipValue}")
    gracefulExit()
    return 0


def parseRankList(ipStr: str) -> list:
    ranks: list = []
    if ipStr:
        if ipStr.isnumeric():
            ranks = [int(ipStr)]
        elif "," in ipStr:
            ranks = list(OrderedDict.fromkeys(int(r.strip()) for r in ipStr.split(",")))
        elif ":" in ipStr:
            lo, hi = (int(r.strip()) for r in ipStr.split(":"))
            ranks = list(range(lo, hi + 1))
    return ranks


def getAlgBW(elapsedTimeNS: float, dataSize: int, numIters: int):
    """(avgIterNS, algBW GB/s): bytes divided by nanoseconds is GB/s."""
    avgIterNS = elapsedTimeNS / numIters if numIters != 0 else 0.0
    algBW = dataSize / avgIterNS if avgIterNS != 0 else 0.0
    return (avgIterNS, algBW)


def getSizes(beginSize: int, endSize: int, stepFactor: int, stepBytes: int) -> list:
    sizes, cur, iters = [], beginSize, 0
    while cur <= endSize:
        sizes.append(cur)
        cur = cur * stepFactor if stepBytes == 0 else cur + stepBytes
        iters += 1
        if iters > 100:
            logger.error(f"For finding allSizes numIters: {iters} is greater than maxIters: 100")
            break
    return sizes


def fixBeginSize(commsParams, world_size: int) -> None:
    """In place: make sure every rank gets at least one element (reference ``:218-252``)."""
    coll = commsParams.collective
    if "all_to_all" in coll or coll in ("all_gather", "all_gather_base", "gather", "reduce_scatter_base"):
        if (commsParams.beginSize / commsParams.element_size) < world_size:
            commsParams.beginSize = world_size * commsParams.element_size
        if (getattr(commsParams, "bitwidth", 32) < 32 and
                (commsParams.beginSize / commsParams.element_size / world_size) < commsParams.quant_a2a_embedding_dim):
            commsParams.beginSize = commsParams.quant_a2a_embedding_dim * world_size * commsParams.element_size
    elif coll in ("all_reduce", "reduce"):
        if commsParams.beginSize < commsParams.element_size:
            commsParams.beginSize = commsParams.element_size


def env2int(env_list, default: int = -1) -> int:
    for e in env_list:
        val = int(os.environ.get(e, -1))
        if val >= 0:
            return val
    return default


def read_comms_env_vars() -> dict:
    return {
        "world_size": env2int(["MV2_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "WORLD_SIZE", "SLURM_NTASKS"]),
        "local_size": env2int(["LOCAL_SIZE", "MPI_LOCALNRANKS", "MV2_COMM_WORLD_LOCAL_SIZE",
                               "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"]),
        "global_rank": env2int(["MV2_COMM_WORLD_RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "RANK", "SLURM_PROCID"]),
        "local_rank": env2int(["LOCAL_RANK", "MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK",
                               "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"]),
    }


class bootstrap_info_holder:
    """Communication-world parameters handed to a backend's constructor."""

    def __init__(self, master_ip: str, master_port: str, num_tpu_cores: int, comms_env_params: dict) -> None:
        self.global_rank = comms_env_params["global_rank"]
        self.local_rank = comms_env_params["local_rank"]
        self.local_size = comms_env_params["local_size"]
        self.world_size = comms_env_params["world_size"]
        self.master_ip = master_ip
        self.master_port = master_port
        self.num_tpu_cores = num_tpu_cores


class paramTimer:
    """Accumulating host timer (reference param_profile.py:18-40)."""

    def __init__(self) -> None:
        self.elapsedTimeNS = 0.0
        self._t0 = 0

    def reset(self, newTime: float = 0.0) -> None:
        self.elapsedTimeNS = newTime

    def start(self) -> None:
        self._t0 = time.monotonic_ns()

    def stop(self) -> None:
        self.elapsedTimeNS += time.monotonic_ns() - self._t0

    def getTimeUS(self) -> float:
        return self.elapsedTimeNS / 1e3

    def getTimeNS(self) -> float:
        return self.elapsedTimeNS


class paramDeviceTimer(paramTimer):
    """start/end device events recorded on a stream by :class:`paramStreamGuard`."""

    def __init__(self, name: str, backendFuncs) -> None:
        super().__init__()
        self.name = name
        self.start_event = backendFuncs.get_new_event(enable_timing=True)
        self.end_event = backendFuncs.get_new_event(enable_timing=True)

    def reset(self, newTime: float = 0.0) -> None:
        self.elapsedTimeNS = newTime

    def start(self, stream=None) -> None:
        self.start_event.record(stream)

    def end(self, stream=None) -> None:
        self.end_event.record(stream)

    def elapsedTime(self) -> None:
        """call after a synchronisation that covers both events; torch elapsed_time is in ms"""
        self.elapsedTimeNS += self.start_event.elapsed_time(self.end_event) * 1e6


class paramStreamGuard(ContextDecorator):
    """run the body on ``stream`` (optionally timed with device events), then switch back"""

    def __init__(self, stream, curDevice, backendFuncs, is_blocking: bool = True, timer=None) -> None:
        self.cur_stream = None
        self.stream = stream
        self.curDevice = curDevice
        self.backendFuncs = backendFuncs
        self.is_blocking = is_blocking
        self.timer = timer

    def __enter__(self):
        self.cur_stream = self.backendFuncs.switch_stream(self.stream, self.curDevice)
        if self.timer:
            self.timer.start(self.stream)
        return self

    def __exit__(self, *exc) -> None:
        if self.timer:
            self.timer.end(self.stream)
        if self.is_blocking:
            self.backendFuncs.sync_stream(self.cur_stream, self.curDevice)
        self.backendFuncs.switch_stream(self.cur_stream, self.curDevice)


def equal_splits(numElements: int, world_size: int):
    """Equal all-to-all splits of the sweep (reference ``:1115-1124,1212-1217``): each peer gets
    ``numElements // world_size`` elements; returns (elements actually used, split list)."""
    per = numElements // world_size
    return per * world_size, [per] * world_size
