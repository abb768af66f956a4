"""The backend plug-in boundary of the comms benchmarks -- own restatement of the interface in
reference ``train/comms/pt/pytorch_backend_utils.py`` (``backendFunctions`` ABC ``:156-411``,
``collectiveArgsHolder`` ``:66-153``, ``getBusBW`` ``:200-247``, ``register_customized_backend``
``:417-427``), restricted to what the DLRM all-to-all path and its drivers touch.

Same names, argument meaning and error behaviour, so :class:`param_amd.comms.pt.mi355_backend.
MI355XBackend` can be registered either here or -- when the reference package ``param_bench`` is
importable -- in the reference's own registry.
"""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod

from .comms_utils import paramTimer as _paramTimer

logger = logging.getLogger(__name__)

supportedDevices = ["cpu", "cuda", "rocm"]
supportedC10dBackends = ["nccl", "gloo"]  # "nccl" IS RCCL on ROCm
# the reference's list (pytorch_backend_utils.py:35-53) without all_gather_v / reduce_scatter_v, which its own c10d backend has
# no entry for either, plus the two names its drivers also dispatch on (all_to_all_single, barrier).  The all-to-all family
# is the DLRM sparse path; the rest is what the sweep driver and the trace replay can be asked for.
supportedCollectives = ["all_to_all", "all_to_allv", "all_to_all_single", "all_reduce", "reduce", "barrier",
                        "all_gather", "all_gather_base", "all_gather_object", "gather", "scatter", "broadcast",
                        "broadcast_object_list", "reduce_scatter", "reduce_scatter_base", "incast", "multicast"]
pt2ptPatterns = ["one2one", "pairwise"]
supportedP2pOps = ["send", "recv", "isend", "irecv"]


class collectiveArgsHolder:
    """Parameters of one collective experiment (fields the a2a / emb_lookup path reads;
    reference CollectiveArgsBase ``:66-149``)."""

    def __init__(self) -> None:
        self.group = None
        self.groups = {}
        self.num_pgs = 0
        self.device = {}
        self.world_size = 0
        self.data_type = ""
        self.numIters = 0
        self.numWarmupIters = 0
        self.global_rank = -1
        self.backendFuncs = {}
        self.collective = ""
        self.collectiveId = 0
        self.computeCount = -1
        self.reuseTensors = False
        self.numComputePerIter = 0
        self.numCollPerIter = 0
        self.batch_size = 0
        # embedding-lookup compute kernel (commsComputeBench --kernel emb_lookup)
        self.emb = None
        self.embRequests = None
        self.direction = None
        self.emb_dim = 0
        self.num_emb_tables_batched = -1
        self.num_emb_ops = 0
        self.LookupOut = {}
        self.grad_output = None
        # collective payload
        self.ipTensor_split = []
        self.opTensor_split = []
        self.ipTensor = []
        self.opTensor = []
        self.srcOrDst = -1
        self.asyncOp = -1
        self.dataSize = 0
        self.numElements = 0
        self.waitObj = []
        self.waitObjIds = {}
        self.op = None
        # quantised collectives (reference pytorch_backend_utils.py:132-139; set by comms_utils.initQuantCommCtx)
        self.all2all_qcomm = None
        self.reducescatter_allgather_qcomm = None
        self.allreduce_qcomm = 32
        self.reduce_qcomm = 32
        self.quant_threshold = 0
        self.quant_a2a_embedding_dim = 32
        self.quant_time = _paramTimer()
        self.dequant_time = _paramTimer()
        self.compute_stream = None
        self.use_ext_dist = False
        self.include_0B = False
        self.graph_launches = 0
        self.use_device_time = False
        self.timers = {}
        # sweep driver: process-group id of this rank (--multi-comms), point-to-point patterns, incast / multicast rank lists
        self.pgId = 0
        self.pt2pt = None
        self.window = 100
        self.src_ranks = None
        self.dst_ranks = None
        self.src_rank = -1
        self.dst_rank = -1
        self.p2pOps = []
        self.comm_dev_time = None
        self.profiler = None


class backendFunctions(ABC):
    """Abstract backend: every collective has signature ``(collectiveArgs, retFlag=False)``, reads
    ``ipTensor/opTensor/ipTensor_split/opTensor_split/asyncOp/group`` from the holder, appends a
    waitable to ``collectiveArgs.waitObj`` when ``asyncOp`` and returns it iff ``retFlag``."""

    def __init__(self) -> None:
        self.tcp_store = None
        self.collectiveFunc = {
            "all_to_all_single": self.all_to_all_single,
            "all_to_all": self.all_to_all,
            "all_to_allv": self.all_to_allv,
            "all_reduce": self.all_reduce,
            "reduce": self.reduce,
            "barrier": self.barrier,
            "noop": self.noop,
        }
        self.computeFunc = {"gemm": self.gemm}

    def set_up(self) -> None:
        return

    def tear_down(self) -> None:
        return

    def getBusBW(self, collective: str, algBW: float, collectiveArgs: collectiveArgsHolder) -> float:
        """busBW from algBW (nccl-tests convention, reference ``:200-247``): all_reduce x 2(n-1)/n;
        all_to_all* / (all_)gather / reduce_scatter / scatter x (n-1)/n; reduce / broadcast x 1."""
        busBW = algBW
        n = collectiveArgs.world_size
        if collective == "all_reduce":
            if n != 0:
                busBW = algBW * (2 * (n - 1) / n)
        elif "all_to_all" in collective or collective in (
                "gather", "all_gather", "reduce_scatter", "reduce_scatter_base", "scatter", "all_gather_base",
                "all_gather_object"):
            if n != 0:
                busBW = algBW * ((n - 1) / n)
        elif collective in ("reduce", "broadcast", "broadcast_object_list", "incast", "multicast"):
            busBW = algBW
        else:
            logger.error(f"collective: {collective} is not supported in computing bus BW! ")
        return busBW

    def noop(self, collectiveArgs=None, retFlag=False, pair=False):
        """no-op for the case we want to skip comms/compute"""
        pass

    def sync_barrier(self, collectiveArgs, desc: str = "world"):
        self.barrier(collectiveArgs, name=desc)

    # -- collectives -------------------------------------------------------------------------
    @abstractmethod
    def sayHello(self, global_rank, local_rank, world_size, master_ip): ...

    @abstractmethod
    def all_reduce(self, collectiveArgs, retFlag=False): ...

    @abstractmethod
    def reduce(self, collectiveArgs, retFlag=False): ...

    @abstractmethod
    def all_to_all(self, collectiveArgs, retFlag=False): ...

    @abstractmethod
    def all_to_allv(self, collectiveArgs, retFlag=False): ...

    @abstractmethod
    def all_to_all_single(self, collectiveArgs, retFlag=False): ...

    @abstractmethod
    def complete_accel_ops(self, collectiveArgs): ...

    @abstractmethod
    def barrier(self, collectiveArgs, name="dummy"): ...

    @abstractmethod
    def get_reduce_op(self, opName: str): ...

    # -- compute -----------------------------------------------------------------------------
    @abstractmethod
    def gemm(self, collectiveArgs) -> None: ...

    # -- memory ------------------------------------------------------------------------------
    @abstractmethod
    def get_mem_size(self, collectiveArgs) -> int: ...

    @abstractmethod
    def alloc_random(self, sizeArr, curRankDevice, dtype, scaleFactor=1.0): ...

    @abstractmethod
    def alloc_embedding_tables(self, n, m, curRankDevice, dtype): ...

    @abstractmethod
    def alloc_empty(self, sizeArr, dtype, curRankDevice): ...

    @abstractmethod
    def clear_memory(self, collectiveArgs): ...

    # -- topology ----------------------------------------------------------------------------
    @abstractmethod
    def get_local_rank(self) -> int: ...

    @abstractmethod
    def get_global_rank(self) -> int: ...

    @abstractmethod
    def get_world_size(self) -> int: ...

    @abstractmethod
    def get_local_size(self) -> int: ...

    @abstractmethod
    def get_device(self): ...

    @abstractmethod
    def get_hw_device(self): ...

    @abstractmethod
    def get_default_group(self): ...

    @abstractmethod
    def get_groups(self): ...

    @abstractmethod
    def get_num_pgs(self) -> int: ...

    # -- init --------------------------------------------------------------------------------
    @abstractmethod
    def initialize_backend(self, master_ip, master_port, backend="gloo", eager_mode=False) -> None: ...

    @abstractmethod
    def benchmark_comms(self, benchTime, commsParams) -> None: ...


customized_backend: dict = {}


def register_customized_backend(name: str, func, device=None) -> None:
    """Registry with the reference's semantics (``:417-427``): name -> backend CLASS; a new
    ``device`` string extends the accepted ``--device`` choices."""
    customized_backend[name] = func
    if device is not None and device not in supportedDevices:
        supportedDevices.append(device)
    logger.info(f"Registered custom backend {name} with function {func.__name__}")
