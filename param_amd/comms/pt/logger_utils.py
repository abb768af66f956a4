"""Performance-record plug-in point of the comms benchmarks: metric records, a registry of loggers selected with
``--use-perf-logger NAME [NAME ...]``, and a JSON-lines logger that ships with the build.

The boundary is the reference's (``train/comms/pt/logger_utils.py``: ``commsPerfMetrics`` family ``:22-88``, ``commsPerfLogger``
``:92-120``, ``register_perf_logger`` ``:123-131``; called from ``comms.py:1097-1110``): a logger is an OBJECT with
``logPerf(benchmarkName, metrics, backendFuncs, **kwargs)`` registered under a name; the driver hands it one record per reported
row, on the reporting rank only.  Field names of the records are the reference's, so a logger written against PARAM reads these
unchanged.  ``register()`` in mi355_backend.py style is not needed here: user code imports this module and registers.
"""
from __future__ import annotations

import dataclasses
import json
import logging
import os
from dataclasses import dataclass
from enum import Enum
from typing import Dict, Optional

logger = logging.getLogger(__name__)


class benchType(Enum):
    Collective = 0
    Pt2Pt = 1
    QuantCollective = 2


@dataclass
class commsPerfMetrics:
    """what every record says: operation, data type, benchmark kind, backend, tag, bytes in / out, elements"""

    commsOp: Optional[str] = None
    Datatype: Optional[str] = None
    BenchCommsType: Optional[benchType] = None
    Backend: Optional[str] = None
    Tags: str = ""
    InputSize: float = 0.0
    OutputSize: float = 0.0
    NumElements: int = 0
    NumElements_pair: int = 0


@dataclass
class commsCollPerfMetrics(commsPerfMetrics):
    p50_latency_us: float = 0.0
    p75_latency_us: float = 0.0
    p95_latency_us: float = 0.0
    min_latency_us: float = 0.0
    max_latency_us: float = 0.0
    AlgoBW_GBs: float = 0.0
    BusBW_GBs: float = 0.0
    TFLOPs: Optional[float] = 0.0

    def __post_init__(self):
        self.BenchCommsType = benchType.Collective


@dataclass
class commsQuantCollPerfMetrics(commsPerfMetrics):
    p95_latency_us: float = 0.0
    quant_p95_latency_us: float = 0.0
    dequant_p95_latency_us: float = 0.0
    quant_comms_p95_latency_us: float = 0.0
    TFLOPs: Optional[float] = 0.0

    def __post_init__(self):
        self.BenchCommsType = benchType.QuantCollective


@dataclass
class commsPt2PtPerfMetrics(commsPerfMetrics):
    p50_latency_us: float = 0.0
    p75_latency_us: float = 0.0
    p95_latency_us: float = 0.0
    AvgUniBW_GBs: float = 0.0
    AvgBiBW_GBs: float = 0.0
    TotalUniBW_GBs: float = 0.0
    TotalBiBW_GBs: float = 0.0

    def __post_init__(self):
        self.BenchCommsType = benchType.Pt2Pt


class commsPerfLogger:
    """base of a logger: subclass, implement ``logPerf``, register an instance"""

    def __init__(self, loggerName: str):
        self.name = loggerName

    def logPerf(self, benchmarkName: str, metrics: commsPerfMetrics, backendFuncs, **kwargs) -> None:
        raise NotImplementedError


customized_perf_loggers: Dict[str, commsPerfLogger] = {}


def register_perf_logger(name: str, func: commsPerfLogger) -> None:
    customized_perf_loggers[name] = func
    logger.info(f"Registered custom perf logger {name}")


def dispatch(names, benchmarkName: str, metrics: commsPerfMetrics, backendFuncs, **kwargs) -> None:
    """hand one record to every selected logger; an unregistered name is skipped with a note (comms.py:1099-1110)"""
    for name in names or ():
        if name in customized_perf_loggers:
            customized_perf_loggers[name].logPerf(benchmarkName, metrics, backendFuncs, **kwargs)
        else:
            logger.info(f"Skipping logger '{name}' because it is not registered or implemented")


class JsonLinesPerfLogger(commsPerfLogger):
    """``--use-perf-logger jsonl``: one JSON object per record appended to ``$PARAM_PERF_LOG`` (default
    ``./param_comms_perf.jsonl``): the record's fields, the benchmark name, world size and rank"""

    def __init__(self, loggerName: str = "jsonl", path: Optional[str] = None):
        super().__init__(loggerName)
        self.path = path

    def logPerf(self, benchmarkName, metrics, backendFuncs, **kwargs) -> None:
        rec = dataclasses.asdict(metrics)
        rec["BenchCommsType"] = metrics.BenchCommsType.name if metrics.BenchCommsType is not None else None
        rec.update({"benchmark": benchmarkName, "world_size": backendFuncs.get_world_size() if backendFuncs is not None else None,
                    "rank": backendFuncs.get_global_rank() if backendFuncs is not None else None})
        rec.update({k: v for k, v in kwargs.items() if isinstance(v, (int, float, str, bool, type(None)))})
        path = self.path or os.environ.get("PARAM_PERF_LOG", "param_comms_perf.jsonl")
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


register_perf_logger("jsonl", JsonLinesPerfLogger())
