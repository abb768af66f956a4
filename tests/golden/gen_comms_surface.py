#!/usr/bin/env python3
"""Generates tests/golden/comms_surface.json: what the REFERENCE's ``train/comms/pt/comms.py`` prints (own gloo backend, CPU
tensors, ``--c 1`` where the reference validates) for the part of its command line that round 4 added to this build's driver:
the whole collective table, ``--root``, ``--i / --o`` splits, ``--ss``, ``--sb``, ``--num-coll``, ``--tag``, several data types,
``--pt2pt one2one | pairwise`` with ``--src-ranks / --dst-ranks / --window``, ``--multi-comms``.

Stored per case: the argument list (per rank where ranks differ), and from rank 0's stdout every line that does not depend on
the clock -- the ``[Rank   0] allSizes`` line, the ``collective=... src_ranks=... dst_ranks=...`` line, the header lines -- plus
the deterministic columns of every COMMS-RES row (collective, dtype + tag, bytes, elements per rank where the row has that
column, column count, field widths).  Also writes comms_cli_defaults.json: the defaults of the reference's own argument
parser for the flags this build keeps, and perf_metric_fields.json: field names and defaults of the reference's performance
records.  Needs /root/reference (build container only); the fixture is data, nothing of the reference travels."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

CASES = [
    {"name": "rest_of_table_root1", "world": 2,
     "argv": ["--b", "64", "--e", "1024", "--f", "4", "--z", "1", "--c", "1", "--root", "1", "--collective",
              "all_gather,all_gather_base,reduce_scatter,reduce_scatter_base,broadcast,reduce,gather,scatter"]},
    {"name": "a2av_splits", "world": 2, "argv": ["--z", "1", "--c", "1", "--collective", "all_to_allv"],
     "per_rank": [["--i", "2,6", "--o", "2,6"], ["--i", "6,2", "--o", "6,2"]]},
    {"name": "sizes_numcoll_tag", "world": 2,
     "argv": ["--ss", "256,4096,64", "--num-coll", "3", "--tag", "mytag", "--z", "0", "--collective", "all_reduce,all_to_allv"]},
    {"name": "step_bytes_dtypes", "world": 2,
     "argv": ["--b", "64", "--e", "256", "--sb", "64", "--z", "1", "--c", "1", "--data-types", "float32,int32,float16",
              "--collective", "all_reduce,all_gather"]},
    {"name": "pt2pt_one2one", "world": 2,
     "argv": ["--pt2pt", "one2one", "--b", "64", "--e", "1024", "--f", "4", "--window", "8"]},
    {"name": "pt2pt_pairwise", "world": 4,
     "argv": ["--pt2pt", "pairwise", "--src-ranks", "0,1", "--dst-ranks", "2:3", "--b", "128", "--e", "512", "--f", "2",
              "--window", "4", "--tag", "pw"]},
    {"name": "multi_comms", "world": 4,
     "argv": ["--multi-comms", "2", "--b", "64", "--e", "1024", "--f", "4", "--z", "1", "--c", "1", "--root", "1",
              "--collective", "all_reduce,all_to_allv,all_gather,broadcast"]},
    {"name": "eight_ranks", "world": 8,
     "argv": ["--b", "64", "--e", "4096", "--f", "8", "--z", "1", "--c", "1", "--collective", "all_to_allv,all_gather_base,reduce_scatter_base,all_reduce"]},
    {"name": "zero_byte_splits", "world": 4,
     "argv": ["--b", "96", "--e", "384", "--f", "2", "--z", "1", "--include-0B", "--collective", "all_to_all_single"]},
    # incast / multicast: the reference hangs on gloo here (both, 4 ranks, probed with a 60 s limit) -- no golden for them
]


def run(work, case, port, pb):
    world = case["world"]
    env = dict(os.environ, PYTHONPATH=pb, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               LOCAL_SIZE=str(world))
    base = [sys.executable, f"{REF}/train/comms/pt/comms.py", "--master-ip", "127.0.0.1", "--master-port", str(port),
            "--n", "3", "--w", "1", "--backend", "gloo", "--device", "cpu"] + case["argv"]
    procs = [subprocess.Popen(base + (case["per_rank"][r] if "per_rank" in case else []), cwd=work,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(world)]
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (case["name"], e[-3000:])
    return [o for o, _ in outs]


def digest(text, pt2pt=False):
    """clock-free content of a rank's stdout"""
    fixed, rows = [], []
    for ln in text.splitlines():
        if ln.startswith("[Rank") and "allSizes" in ln:
            fixed.append(ln)
        elif ln.startswith(("\t collective=", "\t PARAM COMM environment", "\t backend: ")):
            fixed.append(ln)
        elif ln.startswith("\tCOMMS-RES") and not ln.startswith("\tCOMMS-RES-"):
            fixed.append(ln)
        elif ln.startswith("\tCOMMS-RES-"):
            f = ln.split()
            row = {"name": f[0], "size": int(f[1]), "columns": len(f)}
            if not pt2pt:
                row["elements_per_rank"] = int(f[2])
            row["widths"] = [len(x) for x in re.findall(r"\s*\S+", ln)]
            rows.append(row)
    return {"fixed": fixed, "rows": rows}


def main():
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pb"))
    os.symlink(REF, os.path.join(work, "pb", "param_bench"))
    pb = os.path.join(work, "pb")
    out = []
    for k, case in enumerate(CASES):
        texts = run(work, case, 29611 + k, pb)
        rec = {"name": case["name"], "world": case["world"], "argv": case["argv"]}
        if "per_rank" in case:
            rec["per_rank"] = case["per_rank"]
        rec["rank0"] = digest(texts[0], pt2pt="--pt2pt" in case["argv"])
        rec["other_ranks_print_rows"] = any("COMMS-RES-" in t for t in texts[1:])
        out.append(rec)
        print(case["name"], len(rec["rank0"]["rows"]), "rows;", rec["rank0"]["fixed"][:2])
    json.dump(out, open(os.path.join(HERE, "comms_surface.json"), "w"), indent=1)
    # defaults of the reference's own parser for the flags this build keeps (run in a child: the import needs the alias path)
    code = (
        "import argparse, json\n"
        "from param_bench.train.comms.pt import comms\n"
        "b = comms.commsCollBench(); p = argparse.ArgumentParser(allow_abbrev=False); b.readArgs(p)\n"
        "a, _ = p.parse_known_args([])\n"
        "keep = ['w','n','num_coll','b','e','f','sb','i','o','ss','z','c','bitwidth','quant_a2a_embedding_dim','quant_threshold',"
        "'collective','root','src_ranks','dst_ranks','multi_comms','pt2pt','window','size_start_profiler','profiler_active_iters',"
        "'tag','include_0B','graph_launches','use_device_time','log','nw_stack','data_types','enable_local_report','init_only']\n"
        "print(json.dumps({k: getattr(a, k) for k in keep}))\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=pb), capture_output=True, text=True, cwd=work)
    assert r.returncode == 0, r.stderr[-2000:]
    defaults = json.loads(r.stdout.strip().splitlines()[-1])
    json.dump(defaults, open(os.path.join(HERE, "comms_cli_defaults.json"), "w"), indent=1)
    print("defaults", defaults)
    # field names and defaults of the reference's performance records (logger_utils.py): the perf-logger plug-in boundary
    code = (
        "import dataclasses, json\n"
        "from param_bench.train.comms.pt import logger_utils as L\n"
        "out = {}\n"
        "for c in (L.commsPerfMetrics, L.commsCollPerfMetrics, L.commsQuantCollPerfMetrics, L.commsPt2PtPerfMetrics):\n"
        "    inst = c()\n"
        "    out[c.__name__] = {f.name: (getattr(inst, f.name).name if f.name == 'BenchCommsType' and getattr(inst, f.name) is not None"
        " else getattr(inst, f.name)) for f in dataclasses.fields(c)}\n"
        "out['benchType'] = {m.name: m.value for m in L.benchType}\n"
        "print(json.dumps(out))\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=pb), capture_output=True, text=True, cwd=work)
    assert r.returncode == 0, r.stderr[-2000:]
    json.dump(json.loads(r.stdout.strip().splitlines()[-1]), open(os.path.join(HERE, "perf_metric_fields.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
