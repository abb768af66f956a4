#!/usr/bin/env python3
"""Runs the REFERENCE's unmodified drivers with this build's plug-ins and writes tests/golden/ref_plugin_rows.json.
Needs /root/reference (build container only); the fixture is data, nothing of the reference travels.

1. ``examples/reference_plugin/launch_comms.py <reference>/train/comms/pt/comms.py --backend rccl_xgmi --device cpu`` on 2
   gloo ranks: ``MI355XBackend`` registered in the reference's ``customized_backend`` table, then the reference's script
   run unmodified (this is the command documented there).  The
   deterministic columns of every COMMS-RES row (collective, dtype, bytes, elements per rank, column count) are stored,
   after checking that they equal what the reference's OWN backend (``--backend gloo``) prints for the same arguments.
   Blocking and non-blocking mode, the all-to-all family plus the collectives the driver itself needs (all_gather for
   its latency report) and the rest of the ABC's table.
2. ``examples/reference_plugin/launch_dlrm.py``: the reference's ``dlrm.py`` with ``dlrm.PyTorchDistBackend`` bound to
   ``MI355XBackend`` (dlrm.py:1327 hard-codes its backend class: that assignment is the one line a maintainer adds).
   With PARAM_AMD_HOST_TABLES=1 the embedding tables are a torch CPU stand-in -- the product lookup has no CPU path --
   so what is exercised here is every collective of the DLRM iteration through the plug-in: the ``--print-comms``
   records must equal tests/golden/dlrm_np2 (made by the reference with its own backend).
3. ``train/compute/python``: the operator, its input iterator and data generator registered in the reference's
   registries (param_amd/compute/python/reference_plugin.py); the reference's ``BenchmarkConfig`` resolves its example
   config to them and the (id, arguments) stream of build and input configs is stored for the host test to replay
   through this build's own iterator.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# the launchers a user runs: committed files, used here as they are
COMMS_LAUNCH = os.path.join(REPO, "examples", "reference_plugin", "launch_comms.py")
DLRM_LAUNCH = os.path.join(REPO, "examples", "reference_plugin", "launch_dlrm.py")

SWEEPS = [
    {"name": "blocking_a2a_family", "z": "1",
     "collective": "all_to_allv,all_to_all_single,all_reduce"},
    {"name": "nonblocking_rest_of_table", "z": "0",
     "collective": "all_to_allv,all_gather,all_gather_base,reduce_scatter,reduce_scatter_base,broadcast,reduce,gather,scatter"},
    # --multi-comms: the reference's driver leaves the group table on the backend object and calls initialize_groups(backend=...)
    # (comms.py:1455-1456); --pt2pt: send / recv / P2POp / batch_isend_irecv of the plug-in under the reference's four measurements
    {"name": "multi_comms_groups", "z": "1", "collective": "all_reduce,all_to_allv", "extra": ["--multi-comms", "2"]},
    {"name": "pt2pt_one2one", "z": "0", "collective": "all_reduce", "extra": ["--pt2pt", "one2one", "--window", "4"], "pt2pt": True,
     "c": "0"},        # the reference's --c 1 dereferences an unset rank list under --pt2pt (comms_utils.py:1036)
]


def run2(work, script, argv, port, pythonpath, pre=(), extra_env=None):
    env = dict(os.environ, PYTHONPATH=pythonpath, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               LOCAL_SIZE="2", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, script, *pre, "--master-ip", "127.0.0.1", "--master-port", str(port)] + argv, cwd=work,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    return outs[0][0]


def rows_of(text):
    rows = []
    for ln in text.splitlines():
        m = re.match(r"\tCOMMS-RES-(\w+)-(\w+)\s+(\d+)\s+(\d+)\s", ln)
        if m:
            rows.append({"collective": m.group(1), "dtype": m.group(2), "size": int(m.group(3)),
                         "elements_per_rank": int(m.group(4)), "columns": len(ln.split())})
    return rows


def pt2pt_rows_of(text):
    return [{"name": ln.split()[0], "size": int(ln.split()[1]), "columns": len(ln.split())} for ln in text.splitlines()
            if ln.startswith("\tCOMMS-RES-")]


def main():
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pb"))
    os.symlink(REF, os.path.join(work, "pb", "param_bench"))
    pb = os.path.join(work, "pb")
    result = {"comms": [], "dlrm": {}, "compute_python": {}}
    port = 29581

    # 1. comms.py sweeps: plug-in vs the reference's own backend
    for sw in SWEEPS:
        common = ["--b", "64", "--e", "1024", "--f", "4", "--n", "3", "--w", "1", "--z", sw["z"], "--c", sw.get("c", "1"),
                  "--collective", sw["collective"], "--device", "cpu"] + sw.get("extra", [])
        plug = run2(work, COMMS_LAUNCH, common + ["--backend", "rccl_xgmi"], port, pb, pre=(f"{REF}/train/comms/pt/comms.py",))
        own = run2(work, f"{REF}/train/comms/pt/comms.py", common + ["--backend", "gloo"], port + 1, pb)
        port += 2
        assert "Hello from Rank 1" in plug, "sayHello() under the reference driver"
        r_plug, r_own = (rows_of(plug), rows_of(own)) if not sw.get("pt2pt") else (pt2pt_rows_of(plug), pt2pt_rows_of(own))
        assert r_plug and r_plug == r_own, (r_plug, r_own)
        result["comms"].append({"name": sw["name"], "args": common + ["--backend", "rccl_xgmi"], "rows": r_plug,
                                "header": [ln for ln in plug.splitlines() if "COMMS-RES" in ln and ("total-size" in ln or "pingLatency" in ln)][0]})
        print(sw["name"], len(r_plug), "rows == reference backend's")

    # 2. dlrm.py with the plug-in's collectives: --print-comms records == the reference's own (dlrm_np2 fixture)
    flags = ["--backend", "rccl_xgmi", "--device", "cpu", "--mini-batch-size", "8", "--num-batches", "4", "--warmup-batches", "1",
             "--arch-mlp-bot", "16-8", "--arch-mlp-top", "8-1", "--arch-sparse-feature-size", "8",
             "--arch-embedding-size", "100-200-300-400", "--num-indices-per-lookup", "5", "--print-comms"]
    run2(work, DLRM_LAUNCH, flags, port, f"{pb}:{REF}/train/comms/pt", extra_env={"PARAM_AMD_HOST_TABLES": "1"})
    for r in (0, 1):
        got = json.load(open(os.path.join(work, "dlrm_np2", f"rank{r}.json")))
        exp = json.load(open(os.path.join(HERE, "dlrm_np2", f"rank{r}.json")))
        assert got == exp, f"rank {r}: records differ from the reference backend's"
        result["dlrm"][f"rank{r}_records"] = len(got)
    result["dlrm"]["args"] = flags
    result["dlrm"]["equal_to_reference_backend_records"] = True
    print("dlrm.py print-comms records == dlrm_np2 fixture")

    # 3. compute/python registries
    sys.path.insert(0, pb)
    sys.path.insert(0, REPO)
    from param_amd.compute.python.reference_plugin import GENERATOR_NAME, ITERATOR_NAME, OP_NAME, register_in_reference
    reg = register_in_reference()
    from param_bench.train.compute.python.lib.config import BenchmarkConfig
    from param_bench.train.compute.python.lib import pytorch as lib_pytorch
    from param_bench.train.compute.python.lib.init_helper import load_modules
    load_modules(lib_pytorch)          # registers "PyTorch:DefaultDataGenerator" (as run_benchmark.py:226 does)
    cfg_text = json.dumps({OP_NAME: {
        "build_iterator": "RangeConfigIterator", "input_iterator": ITERATOR_NAME,
        "build_data_generator": "PyTorch:DefaultDataGenerator", "input_data_generator": GENERATOR_NAME,
        "config": [{"build": [{"args": [
            {"type": "int", "name": "num_tables", "value": [1, 2], "__range__": ["value"]},
            {"type": "int", "name": "rows", "value": 2000}, {"type": "int", "name": "dim", "value": 64},
            {"type": "int", "name": "pooling", "value": 0}, {"type": "bool", "name": "weighted", "value": False},
            {"type": "str", "name": "weights_precision", "value": "fp32"}],
            "kwargs": {"optimizer": {"type": "str", "value": "exact_row_wise_adagrad"}}}],
            "input": [{"args": [{"type": "int", "name": "batch_size", "value": [4, 8, 4], "__range__": ["value"]},
                                {"type": "int", "name": "pooling_factor", "value": [3, 5], "__list__": ["value"]}]}]}]}})
    bc = BenchmarkConfig({"device": "cpu"})
    bc.load_json(cfg_text)
    assert len(bc.op_configs) == 1 and bc.op_configs[0].op is reg["operator"]
    oc = bc.op_configs[0]
    stream = []
    import torch
    for config in oc.info["config"]:
        for build_id, build_config in oc.build_iterator(config, "build", "cpu"):
            for input_id, input_config in oc.input_iterator({"build": build_config, "input": config["input"]}, "input", "cpu"):
                (idx, off, psw), _ = oc.input_data_generator().get_data(input_config, "cpu", alpha=0.25)
                stream.append({"build_id": build_id, "input_id": input_id,
                               "build_args": [a["value"] for a in build_config["args"]],
                               "input_args": [a["value"] for a in input_config["args"]],
                               "indices_len": int(idx.numel()), "offsets_len": int(off.numel()),
                               "indices_sum": int(idx.sum()), "offsets_last": int(off[-1]), "psw": psw is not None})
    result["compute_python"] = {"config": json.loads(cfg_text), "stream": stream, "alpha": 0.25}
    print("compute/python:", len(stream), "(build, input) configs through the reference's BenchmarkConfig")

    json.dump(result, open(os.path.join(HERE, "ref_plugin_rows.json"), "w"), indent=1)
    print("wrote ref_plugin_rows.json")


if __name__ == "__main__":
    main()
