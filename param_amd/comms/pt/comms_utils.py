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


def checkQuantArgs(collective: str, dtype, beginSize: int, quant_a2a_embedding_dim: int, blockingFlag) -> None:
    """What the reference refuses for ``--bitwidth < 32`` (comms_utils.py:395-429): only the all_to_all family, reduce and
    all_reduce; float32 payloads; a quantised all_to_all must be blocking; a begin size that is not a whole number of
    rows only warns."""
    import torch

    if "all_to_all" not in collective and collective not in ("reduce", "all_reduce"):
        raise NotImplementedError(f"quantized communication for {collective} is currently unsupported.")
    if "all_to_all" in collective:
        if (beginSize // 4) % quant_a2a_embedding_dim != 0:
            logger.warning(f"begin size {beginSize} must be a multiple of --quant-a2a-embedding-dim "
                           f"{quant_a2a_embedding_dim} for all_to_all operation")
        if blockingFlag != 1:
            raise NotImplementedError("quantized All_to_all must be synchronous.")
    if dtype != torch.float32:
        raise NotImplementedError(f"quantization for {dtype} is not supported. Use float32 instead.")


def initQuantCommCtx(collectiveArgs, commsParams) -> None:
    """Arm the quantised collectives of the backend (reference comms_utils.py:371-391, where the handlers come from an
    unpublished package and the open-source build resets ``bitwidth`` to 32): the all_to_all family exchanges row-wise
    quantised payloads (``param_amd.quant``: fp16 / fused 8-, 4-, 2-bit rows of ``quant_a2a_embedding_dim`` values),
    all_reduce / reduce downcast as the reference's ``_downcast`` does (pytorch_dist_backend.py:48-54)."""
    logger.info(f"communication bitwidth set to {commsParams.bitwidth}")
    collectiveArgs.all2all_qcomm = commsParams.bitwidth
    collectiveArgs.allreduce_qcomm = commsParams.bitwidth
    collectiveArgs.reduce_qcomm = commsParams.bitwidth
    collectiveArgs.quant_a2a_embedding_dim = commsParams.quant_a2a_embedding_dim


def clearQuantCommCtx(collectiveArgs) -> None:
    collectiveArgs.all2all_qcomm = None
    collectiveArgs.allreduce_qcomm = 32
    collectiveArgs.reduce_qcomm = 32


def init_logging(level_name: str) -> None:
    """root logger as the reference's drivers set it up (``:1895-1906``): the level named by ``--log``, every line prefixed with
    time, logger, level and this process's rank from the launcher's environment; an unknown level name is an error"""
    level = getattr(logging, str(level_name).upper(), None)
    if not isinstance(level, int):
        raise ValueError(f"Invalid log level: {level_name}")
    rank = read_comms_env_vars()["global_rank"]
    # no force=True: a host application (or a test runner) that configured logging first keeps its handlers; the level applies
    logging.basicConfig(level=level, format="[%(asctime)s][%(name)s][%(levelname)s][Rank{:3}] - %(message)s".format(rank))
    logging.getLogger().setLevel(level)


def get_rank_details(backendFuncs):
    """(local rank, global rank, world size, default group, device, hardware device) of this process, from the backend
    (reference ``:255-273``)"""
    return (backendFuncs.get_local_rank(), backendFuncs.get_global_rank(), backendFuncs.get_world_size(),
            backendFuncs.get_default_group(), backendFuncs.get_device(), backendFuncs.get_hw_device())


def ensureTensorFlush(tensors):
    """read the last element of the (last) tensor back to the host: the collective that produced it has really finished
    (reference ``:488-508``); returns that value, None for an empty list"""
    last = tensors[-1] if isinstance(tensors, (list, tuple)) and len(tensors) > 0 else tensors
    if last is None or isinstance(last, (list, tuple)) or last.nelement() == 0:
        return None
    return last.reshape(-1)[-1].item()


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

    def incrTimeNS(self, timeNS: float) -> None:
        self.elapsedTimeNS += timeNS

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


# ------------------------------------------------------------------------------------------------
# trace replay: one recorded operation, collective-name normalisation, embedding-lookup setup

_NAME_ALIASES = {
    "alltoall": "all_to_all", "alltoallv": "all_to_allv", "alltoallbase": "all_to_allv",
    "alltoallsingle": "all_to_all_single", "allreduce": "all_reduce", "allgather": "all_gather",
    "allgatherbase": "all_gather_base", "reducescatter": "reduce_scatter",
    "reducescatterbase": "reduce_scatter_base", "recvanysource": "recv",
}


def paramToCommName(name: str, supported_comms=None) -> str:
    """Map the spellings found in traces ("alltoallv", "ALL_TO_ALLV", "AllToAllBase" ...) to the internal
    collective name (reference ``:446-486``): lower-case, letters only, then the alias table; unknown names
    pass through unchanged.  With ``supported_comms`` an unknown result is a fatal configuration error."""
    key = "".join(ch for ch in name.lower() if ch.isalpha())
    new_name = _NAME_ALIASES.get(key, name)
    if supported_comms is not None and new_name not in supported_comms:
        logger.error(f"{name} is not a supported communication in PARAM! Supported comms: {list(supported_comms)}")
        gracefulExit()
    return new_name


class commsArgs:
    """One operation of a comms trace: a collective (``comms``) or a compute kernel (``compute``).
    Same attribute and JSON-key names as the reference's ``commsArgs`` (``:552-713``) for the fields the replay of
    collectives and of ``emb_lookup`` reads; sizes are in ELEMENTS.  The embedding-lookup fields carry the names
    the reference's parser and ``init_emb_lookup`` actually use (``emb_dim, num_embs, batch_size,
    num_emb_tables_per_device, num_emb_tables_batched, bag_size``); the reference's constructor declares
    camel-case twins it never reads (``embDim`` ..., and ``toEmbLookupTuple`` reads an unset ``bagSize``)."""

    _FIELDS = ("comms", "compute", "id", "req", "inMsgSize", "outMsgSize", "dtype", "inSplit", "outSplit", "startTimeNs",
               "pgId", "groupRanks", "worldSize", "markerStack", "root", "src_rank", "dst_rank", "count",
               "emb_dim", "num_embs", "batch_size", "num_emb_tables_per_device", "bag_size")

    def __init__(self, **kwargs) -> None:
        for f in self._FIELDS:
            setattr(self, f, kwargs.get(f))
        self.direction = kwargs.get("direction", "forward")
        self.num_emb_tables_batched = kwargs.get("num_emb_tables_batched", -1)
        self.device = kwargs.get("device")

    def toDict(self) -> dict:
        d = {}
        if self.comms is not None:
            d["comms"] = self.comms
        if self.compute is not None:
            d["compute"] = self.compute
            if self.compute == "emb_lookup":
                for key, val in (("direction", self.direction), ("emb_dim", self.emb_dim), ("num_embs", self.num_embs),
                                 ("batch_size", self.batch_size), ("num_emb_tables", self.num_emb_tables_per_device),
                                 ("bag_size", self.bag_size), ("count", self.count)):
                    if val is not None:
                        d[key] = val
        if self.req is not None:
            d["req"] = self.req
        if self.inMsgSize is not None:
            d["in_msg_size"], d["out_msg_size"], d["dtype"] = self.inMsgSize, self.outMsgSize, self.dtype
        if self.inSplit is not None:
            d["in_split"] = self.inSplit
        if self.outSplit is not None:
            d["out_split"] = self.outSplit
        if self.startTimeNs is not None:
            d["startTime_ns"] = self.startTimeNs
        if self.pgId is not None:
            d["pg_id"] = self.pgId
        if self.worldSize is not None:
            d["world_size"] = self.worldSize
        if self.root is not None:
            d["root"] = self.root
        return d

    def toEmbLookupTuple(self):
        """key under which ``--reuse-tensors`` caches the tables/requests of an ``emb_lookup`` entry"""
        return (self.direction, self.emb_dim, self.num_embs, self.batch_size, self.num_emb_tables_per_device, self.bag_size)

    def __eq__(self, other) -> bool:
        return isinstance(other, commsArgs) and self.__dict__ == other.__dict__

    def __repr__(self) -> str:
        return str(self.__dict__)


def init_emb_lookup(collectiveArgs, commsParams, backendFuncs) -> None:
    """Tables and requests of the ``emb_lookup`` compute kernel (reference ``:1956-2039``).  ``commsParams``
    is anything carrying ``direction, emb_dim, num_embs, batch_size, num_emb_tables_per_device,
    num_emb_tables_batched, bag_size`` -- the CLI namespace of commsComputeBench or a trace entry.
    ``num_emb_tables_per_device // num_emb_tables_batched`` batched ops (-1: one op holding all tables), each
    ``num_emb_tables_batched`` tables of ``num_embs x emb_dim`` fp32 rows served by the HIP kernels; one request
    per op in the fbgemm layout (indices ``[T*B*L]`` table-major, offsets ``[T*B+1]``, uniform ids).  Backward:
    a forward pass gives ``LookupOut``; ``grad_output = rand_like(LookupOut)``.  Where the reference logs an
    error and returns when fbgemm is missing, this raises if the HIP library is missing."""
    import torch

    from ...indices import tbe_request

    ca = collectiveArgs
    ca.direction = commsParams.direction or "forward"
    ca.emb_dim, ca.batch_size = commsParams.emb_dim, commsParams.batch_size
    tables = commsParams.num_emb_tables_per_device
    raw = getattr(commsParams, "num_emb_tables_batched", -1)
    raw = -1 if raw in (0, None) else raw
    batched = tables if raw == -1 else raw
    if tables % batched:
        raise ValueError("the number of embedding tables per device must be a multiple of the batched-table count")
    ca.num_emb_tables_batched = raw      # -1 stays -1 (reference :1983): "not batched with the all-to-all"
    ca.num_emb_ops = tables // batched
    ca.emb = [backendFuncs.alloc_batched_embedding_tables([commsParams.num_embs] * batched, ca.emb_dim, ca.device, torch.float32)
              for _ in range(ca.num_emb_ops)]
    ca.embRequests = [tbe_request([commsParams.num_embs] * batched, ca.batch_size, commsParams.bag_size, device=ca.device,
                                  seed=17 * max(ca.global_rank, 0) + i) + (None,) for i in range(ca.num_emb_ops)]
    if ca.direction == "backward":
        for i, (indices, offsets, weights) in enumerate(ca.embRequests):
            ca.LookupOut = ca.emb[i].forward(indices, offsets, weights)
        ca.grad_output = torch.rand_like(ca.LookupOut)
