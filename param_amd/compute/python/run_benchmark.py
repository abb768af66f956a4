"""JSON-config operator microbenchmark -- the slice of reference ``train/compute/python/pytorch/run_benchmark.py``
(``:24-365``) + ``lib/pytorch/build_executor.py`` / ``op_executor.py`` / ``lib/iterator.py`` needed to run the batched
EmbeddingBag operator from the reference's config files (schema of ``examples/pytorch/configs/
split_table_batched_embeddings_ops.json``: op name -> ``build_iterator``, ``input_iterator``, ``config: [{build: [{args,
kwargs}], input: [{args}]}]``).

    python -m param_amd.compute.python.run_benchmark -c my_config.json -d cuda -b --warmup 5 --iteration 20 \
        [--exec-mode discrete|continuous|continuous_events] [--cuda-l2-cache on|off] [-o prefix]

Per (op, build, input) combination one JSON line ``{"op_name", "id": "<config>|<build id>|<input id>", "metric": {"forward":
{"gpu.time": [...ms], "gpu.memory": [...MB]}, "backward": {...}}, "config": {"build", "input"}}`` (reference
``output_stats``, ``build_executor.py:511-541``).  Kept from the reference:

* ``"build_iterator": "RangeConfigIterator"`` expands ``__range__`` arguments of the build configs (config_iter.py,
  pinned to the reference's iterator); the operator's input iterator expands ``__range__`` / ``__list__`` on
  ``[batch_size, pooling_factor]``;
* execution modes (``op_executor.py:157-477``): ``discrete`` -- every call timed on the host clock closed by a device
  synchronize (``timer.py:19-27``), optionally with the caches flushed before each call; ``continuous`` -- one clock
  around all iterations, backward = (forward + backward loop) - forward; ``continuous_events`` -- one start/stop
  device-event pair per call, no host synchronisation inside the loop;
* ``--cuda-l2-cache off``: the reference flushes by writing a buffer the size of the L2 and has no entry for gfx950
  (``op_executor.py:16-28`` raises KeyError there); here the flush writes 2 x (8 x 4 MiB L2 + 256 MiB MALL).
* ``--cuda-graph``: forward and backward captured once in HIP graphs and replayed (``op_executor.py:82-97``).
* ``-r / --resume-id`` and ``-s / --stop_id``: run ids ``<op name>|<config>|<build id>|<input id>``; everything before the resume
  id is skipped, the run ends in front of the stop id (``build_executor.py:72-102,449-463``); ``-a / --append``; ``-o`` writes
  ``<prefix>.json`` headed by one line of run options + system information (``run_benchmark.py:334-344``); ``-p / --profile``
  wraps the run in torch.profiler (``<prefix>_trace.json``), ``--et`` collects an execution trace (``<prefix>_et.json``);
  ``-l / --log-level``, ``--version``.
Not kept: the NSight / CUPTI launchers and their batch mode (NVIDIA tools; on this platform the same JSON runs under
``rocprofv3 --kernel-trace --stats -- python -m param_amd.compute.python.run_benchmark ...``), ``--pt2-model``.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time
from datetime import datetime

import torch

from . import op_map
from .config_iter import BUILD_ITERATORS, tbe_input_iterator
from .split_table_batched_embeddings_ops import generate_batched_request

_FLUSH_BYTES = 2 * (8 * 4 * 1024 * 1024 + 256 * 1024 * 1024)   # MI355X: 8 XCD L2s + the memory-side cache, twice over
_flush_bufs = {}


def _clear_cache(device) -> None:
    buf = _flush_bufs.get(str(device))
    if buf is None:
        buf = _flush_bufs[str(device)] = torch.empty(_FLUSH_BYTES // 4, dtype=torch.float32, device=device)
    buf.fill_(2.0)


__version__ = "1.0.0-mi355x"       # the reference prints its package version for --version (lib/__init__.py)
logger = logging.getLogger(__name__)


class RunWindow:
    """Which (op, config, build, input) combinations of a config file run: skip until the id given by ``--resume-id`` comes up,
    run from there, stop IN FRONT of the id given by ``--stop_id`` (reference BuildExecutor.get_transition_state,
    build_executor.py:72-102; ids ``<op name>|<config>|<build id>|<input id>``, ``:449``)."""

    SKIP, RUN, STOP = "skip", "run", "stop"

    def __init__(self, resume_id=None, stop_id=None):
        self.resume_id, self.stop_id = resume_id, stop_id
        self.state = self.SKIP if resume_id else self.RUN

    def step(self, run_id: str) -> str:
        if self.state == self.SKIP and run_id == self.resume_id:
            self.state = self.RUN
            logger.info(f"Resume benchmark check matched [{run_id}]")
        if run_id == self.stop_id:
            self.state = self.STOP
            logger.info(f"Stop benchmark check matched [{run_id}]")
        return self.state


def _arg_values(arg_list):
    def val(a):
        v = a["value"]
        if a.get("type") in ("genericlist", "tuple"):
            return [val(x) if isinstance(x, dict) else x for x in v]
        return v
    return [val(a) for a in arg_list]


class OpExecutor:
    """runs one built operator on one request under one execution mode; returns the reference's metric dict"""

    def __init__(self, op, device: str, warmup: int, iteration: int, backward: bool, exec_mode: str, l2_cache: bool,
                 use_graph: bool = False):
        self.op, self.device = op, device
        self.warmup, self.iteration, self.backward = warmup, iteration, backward
        self.exec_mode, self.l2_cache, self.use_graph = exec_mode, l2_cache, use_graph
        self._fwd = self._bwd = None

    def _bind(self, data) -> None:
        """the two callables the timing loops launch: eager calls, or replays of HIP graphs captured once
        (``--cuda-graph``, reference ``op_executor.py:82-97``; the kernels go through the C ABI on the capturing stream,
        the radix sort's launches included; the stochastic-rounding seed of a captured Adagrad step is frozen)"""
        if not self.use_graph:
            self._fwd = lambda: self.op.forward(*data)

            def bwd():
                self.op.create_grad()
                self.op.backward()
            self._bwd = bwd
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # warm up off the default stream, as graph capture requires
            self.op.forward(*data)
            if self.backward:
                self.op.create_grad()
                self.op.backward()
        torch.cuda.current_stream().wait_stream(side)
        fg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(fg):
            self.op.forward(*data)
        self._fwd = fg.replay
        if self.backward:
            self.op.create_grad()
            bg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(bg):
                self.op.backward()
            self._bwd = bg.replay

    def _timed(self, fn):
        if not self.l2_cache:
            _clear_cache(self.device)
        torch.cuda.reset_peak_memory_stats(self.device)
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(self.device)
        return (time.perf_counter() - t0) * 1e3, torch.cuda.max_memory_allocated(self.device) / 1048576

    def _discrete(self, count, data):
        fw_t, fw_m, bw_t, bw_m = [], [], [], []
        for _ in range(count):
            t, m = self._timed(self._fwd)
            fw_t.append(t)
            fw_m.append(m)
            if self.backward:
                t, m = self._timed(self._bwd)
                bw_t.append(t)
                bw_m.append(m)
        return fw_t, fw_m, bw_t, bw_m

    def _continuous(self, count, data):
        dev = self.device
        torch.cuda.reset_peak_memory_stats(dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(count):
            self._fwd()
        torch.cuda.synchronize(dev)
        fw = (time.perf_counter() - t0) * 1e3 / count
        fw_m, bw_t, bw_m = [torch.cuda.max_memory_allocated(dev) / 1048576], [], []
        if self.backward:
            torch.cuda.reset_peak_memory_stats(dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(count):
                self._fwd()
                self._bwd()
            torch.cuda.synchronize(dev)
            bw_t = [(time.perf_counter() - t0) * 1e3 / count - fw]      # forward time subtracted (op_executor.py:405)
            bw_m = [torch.cuda.max_memory_allocated(dev) / 1048576]
        return [fw], fw_m, bw_t, bw_m

    def _continuous_events(self, count, data):
        dev = self.device

        def pairs():
            return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]

        ev = pairs()
        torch.cuda.reset_peak_memory_stats(dev)
        torch.cuda.synchronize(dev)
        for a, b in ev:
            a.record()
            self._fwd()
            b.record()
        torch.cuda.synchronize(dev)
        fw_t, fw_m, bw_t, bw_m = [a.elapsed_time(b) for a, b in ev], [torch.cuda.max_memory_allocated(dev) / 1048576], [], []
        if self.backward:
            ev = pairs()
            torch.cuda.reset_peak_memory_stats(dev)
            torch.cuda.synchronize(dev)
            for a, b in ev:
                self._fwd()
                a.record()
                self._bwd()
                b.record()
            torch.cuda.synchronize(dev)
            bw_t, bw_m = [a.elapsed_time(b) for a, b in ev], [torch.cuda.max_memory_allocated(dev) / 1048576]
        return fw_t, fw_m, bw_t, bw_m

    def run(self, data):
        bench = {"discrete": self._discrete, "continuous": self._continuous, "continuous_events": self._continuous_events}[self.exec_mode]
        self._bind(data)
        if self.warmup:
            bench(self.warmup, data)
        fw_t, fw_m, bw_t, bw_m = bench(self.iteration, data) if self.iteration else ([], [], [], [])
        metrics = {"forward": {"gpu.time": fw_t, "gpu.memory": fw_m}}
        if self.backward:
            metrics["backward"] = {"gpu.time": bw_t, "gpu.memory": bw_m}
        return metrics


def run_op(name: str, op_cfg: dict, device: str, warmup: int, iters: int, backward: bool, out_stream=None, alpha=1.0,
           exec_mode: str = "discrete", l2_cache: bool = True, use_graph: bool = False, window: RunWindow = None):
    if name not in op_map:
        raise KeyError(f"operator {name!r} is not registered (registered: {sorted(op_map)})")
    build_iter = op_cfg.get("build_iterator")
    if build_iter not in BUILD_ITERATORS:
        raise KeyError(f"build_iterator {build_iter!r} is not one of {[k for k in BUILD_ITERATORS if k]}")
    op = op_map[name]
    op.device = device
    dev = "cuda" if device.startswith(("cuda", "rocm")) else device
    out_stream = sys.stdout if out_stream is None else out_stream
    results = []
    for ci, cfg in enumerate(op_cfg["config"]):
        for build_id, build in BUILD_ITERATORS[build_iter](cfg["build"]):
            bargs = _arg_values(build.get("args", []))
            bkw = {k: v["value"] for k, v in (build.get("kwargs") or {}).items()}
            num_tables, rows, _dim, _pool, weighted = bargs[0], bargs[1], bargs[2], bargs[3], bargs[4]
            for input_id, (batch_size, pooling_factor) in tbe_input_iterator(cfg["input"]):
                if window is not None:
                    state = window.step(f"{name}|{ci}|{build_id}|{input_id}")
                    if state == RunWindow.SKIP:
                        continue
                    if state == RunWindow.STOP:
                        op.cleanup()
                        return results
                op.cleanup()
                kw = dict(bkw)
                if len(bargs) < 7:
                    kw.setdefault("optimizer", "exact_row_wise_adagrad")  # the reference's choice (comms_utils.py:2014)
                op.build(*bargs, **kw)
                data = generate_batched_request(num_tables, rows, batch_size, pooling_factor, alpha=alpha,
                                                weighted=weighted, device=dev)
                metrics = OpExecutor(op, dev, warmup, iters, backward, exec_mode, l2_cache, use_graph).run(data)
                # reference run id: "<config>|<build id>|<input id>" (benchmark.py:79,101 + build_executor.py:466; the
                # reference appends every further build id of a config to the previous one -- not reproduced)
                stats = {"op_name": name, "id": f"{ci}|{build_id}|{input_id}", "metric": metrics,
                         "config": {"build": {"args": bargs, "kwargs": bkw}, "input": {"args": [batch_size, pooling_factor]}}}
                for pass_name, metric in metrics.items():          # the reference's log lines per pass and metric (build_executor.py:514-532)
                    logger.info(f"pass: {pass_name}")
                    for metric_name, records in metric.items():
                        total = sum(records) if records else 0
                        avg = total / len(records) if records else 0
                        unit = "ms" if metric_name.endswith(".time") else "MB"
                        logger.info(f"metric: {metric_name}, average: {avg:.3f} {unit}, total: {total:.3f} {unit}")
                        logger.info("[" + ", ".join(f"{x:.3f}" for x in records) + "]")
                out_stream.write(json.dumps(stats) + "\n")
                out_stream.flush()
                results.append(stats)
    op.cleanup()
    return results


def sys_info() -> dict:
    """what the header line of an output file says about the box (reference get_sys_info, lib/pytorch/config_util.py)"""
    info = {"pytorch_version": torch.__version__, "hip_version": getattr(torch.version, "hip", None), "device": None}
    if torch.cuda.is_available():
        props = torch.cuda.get_device_properties(0)
        info.update({"device": props.name, "gcn_arch": getattr(props, "gcnArchName", None), "cu_count": props.multi_processor_count,
                     "memory_GB": round(props.total_memory / 2**30, 1)})
    return info


def main(argv=None):
    ap = argparse.ArgumentParser(description="operator microbenchmark from a JSON config (MI355X build)")
    ap.add_argument("-c", "--config", type=str, help="The benchmark config file.")
    ap.add_argument("-d", "--device", type=str, default="cuda")
    ap.add_argument("-w", "--warmup", type=int, default=1)
    ap.add_argument("-i", "--iteration", type=int, default=1)
    ap.add_argument("-b", "--backward", action="store_true")
    ap.add_argument("-o", "--output-prefix", type=str, default=None, help="write <prefix>.json instead of stdout")
    ap.add_argument("-r", "--resume-id", type=str, default=None,
                    help="resume at this run id (<op name>|<config>|<build id>|<input id>), skip everything before it")
    ap.add_argument("-s", "--stop_id", "--stop-id", type=str, default=None, dest="stop_id",
                    help="stop in front of this run id, skip the rest")
    ap.add_argument("-a", "--append", action="store_true", help="append to the output file rather than overwrite")
    ap.add_argument("--exec-mode", type=str, default="discrete", choices=["discrete", "continuous", "continuous_events"])
    ap.add_argument("--cuda-l2-cache", type=str, default="on", choices=["on", "off"],
                    help="off: flush the L2s and the memory-side cache before every timed call (discrete mode)")
    ap.add_argument("--cuda-graph", action="store_true", help="capture forward / backward in HIP graphs once and replay them")
    ap.add_argument("--alpha", type=float, default=1.0, help="generate_requests distribution switch (reference :93-135)")
    ap.add_argument("-p", "--profile", action="store_true", help="torch.profiler around the run: <prefix>_trace.json")
    ap.add_argument("--et", action="store_true", help="collect an execution trace: <prefix>_et.json")
    ap.add_argument("-l", "--log-level", type=str, default="INFO")
    ap.add_argument("--version", action="store_true", help="print the version and stop")
    a = ap.parse_args(argv)
    logging.basicConfig(level=getattr(logging, a.log_level.upper(), logging.INFO))
    if a.version:
        print(f"PARAM train compute version: {__version__}")
        return []
    if not a.config:
        ap.print_usage()
        return []
    if a.cuda_graph and not a.device.startswith(("cuda", "rocm")):
        logger.warning("Cannot use --cuda-graph when not running on cuda device, cuda-graph is disabled")
        a.cuda_graph = False
    cfg = json.load(open(a.config))
    stream = open(a.output_prefix + ".json", "a" if a.append else "w") if a.output_prefix else None
    prefix = a.output_prefix or f"benchmark_result_{os.getpid()}"
    window = RunWindow(a.resume_id, a.stop_id)
    out = []
    et = prof = None
    try:
        if stream:     # header line of the reference's result files (run_benchmark.py:334-344)
            print(json.dumps({"run_options": {k: v for k, v in vars(a).items()}, "sys_info": sys_info(),
                              "start_time": datetime.now().isoformat(timespec="seconds")}, default=str), file=stream)
        if a.et:
            from torch.profiler import ExecutionTraceObserver

            et = ExecutionTraceObserver()
            et.register_callback(f"{prefix}_et.json")
            et.start()
        if a.profile:
            acts = [torch.profiler.ProfilerActivity.CPU] + ([torch.profiler.ProfilerActivity.CUDA] if torch.cuda.is_available() else [])
            prof = torch.profiler.profile(activities=acts, record_shapes=True)
            prof.start()
        for name, op_cfg in cfg.items():
            out += run_op(name, op_cfg, a.device, a.warmup, a.iteration, a.backward, out_stream=stream, alpha=a.alpha,
                          exec_mode=a.exec_mode, l2_cache=a.cuda_l2_cache == "on", use_graph=a.cuda_graph, window=window)
            if window.state == RunWindow.STOP:
                break
    finally:
        if prof is not None:
            prof.stop()
            prof.export_chrome_trace(f"{prefix}_trace.json")
        if et is not None:
            et.stop()
            et.unregister_callback()
        if stream:
            stream.close()
    return out


if __name__ == "__main__":
    main()  # pragma: no cover
