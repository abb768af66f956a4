#!/usr/bin/env python3
"""Generates tests/golden/dlrm_np2/rank{0,1}.json: the per-rank ``--print-comms`` records of the REFERENCE's
``train/comms/pt/dlrm.py`` run here as 2 gloo ranks on CPU (SURVEY Appendix B recipe).  Needs /root/reference;
run in the build container only -- the fixture (data) is committed, nothing of the reference travels.

The harness below only pre-registers ``--use-device-time`` (reference bug R1: dlrm.py reads the attribute but
never adds the flag) and then calls the reference's own entry points.

Also writes dlrm_np2_variants.json: the all_reduce message sizes of rank 0 under the flags that size the top MLP
(``--arch-interaction-op cat``, ``--arch-interaction-itself``, ``--arch-project-size``; dlrm.py:583-601)."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = ["--backend", "gloo", "--device", "cpu", "--mini-batch-size", "8", "--num-batches", "4", "--warmup-batches", "1",
         "--arch-mlp-bot", "16-8", "--arch-mlp-top", "8-1", "--arch-sparse-feature-size", "8",
         "--arch-embedding-size", "100-200-300-400", "--num-indices-per-lookup", "5", "--print-comms"]
HARNESS = '''
import argparse
import dlrm
from param_bench.train.comms.pt import comms_utils
env = comms_utils.read_comms_env_vars()
b = dlrm.commsDLRMBench()
p = argparse.ArgumentParser()
p.add_argument("--use-device-time", action="store_true", default=False)
args = b.readArgs(p); b.checkArgs(args); b.initBench(args, env)
bi = comms_utils.bootstrap_info_holder(args.master_ip, args.master_port, args.num_tpu_cores, env)
b.runBench(bi, comms_utils.commsDlrmParamsHolder(args, env), args)
'''

if __name__ == "__main__":
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pb"))
    os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
    open(os.path.join(work, "harness.py"), "w").write(HARNESS)
    port = "29547"
    env = dict(os.environ, PYTHONPATH=f"{work}/pb:/root/reference/train/comms/pt", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=port, WORLD_SIZE="2", LOCAL_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(work, "harness.py"), "--master-ip", "127.0.0.1",
                               "--master-port", port] + FLAGS, cwd=work, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in (0, 1)]
    texts = [p.communicate(timeout=600)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0]
    out = os.path.join(HERE, "dlrm_np2")
    os.makedirs(out, exist_ok=True)
    for r in (0, 1):
        shutil.copy(os.path.join(work, "dlrm_np2", f"rank{r}.json"), os.path.join(out, f"rank{r}.json"))
    print("wrote", out)
    # the clock-free part of rank 0's report (dlrm.py:1011-1198): header line, per table the (iterations, region, memory p50)
    # columns of every row and the number of tab-separated fields, the total_time rows, the separator lines
    import json as _json

    tables, cur, header, seps = [], None, None, 0
    for ln in texts[0].splitlines():
        f = ln.split("\t")
        if len(f) > 3 and f[1].strip() == "iters" and f[2].strip() == "region":
            header = ln
        elif len(f) >= 6 and f[1].strip().isdigit():
            if f[2].strip() == "intermed_calc_length":
                cur = []
                tables.append(cur)
            if cur is not None:
                cur.append({"iters": int(f[1]), "region": f[2].strip(), "memory": f[3].strip(), "fields": len(f),
                            "widths": [len(x) for x in f]})
        elif ln.strip().startswith("-----"):
            seps += 1
    assert header is not None and len(tables) == 2 and len(tables[0]) == 22, (header, [len(t) for t in tables])
    keep = ("\t mpi-params", "\t rank: 0 args.model", "\tdims_sum_per_rank", "\t ipConfig[", "\t ln_top", "\t ln_bot", "\t n_emb")
    config_lines = [ln for ln in texts[0].splitlines() if ln.startswith(keep)]
    epoch = [ln.replace(os.uname()[1], "<host>") for ln in texts[0].splitlines() if "starting new epoch" in ln]
    _json.dump({"flags": FLAGS, "config_lines": config_lines, "epoch_line": epoch[0],
                "header": header, "tables": tables, "separator_lines": seps,
                "rank1_prints_table": any("intermed_calc_length" in ln for ln in texts[1].splitlines())},
               open(os.path.join(out, "report_rank0.json"), "w"), indent=1)
    print("report rows:", [(r["region"], r["memory"]) for r in tables[0]])
    # top-MLP sizing variants (dlrm.py:583-601): the all_reduce message sizes of rank 0's records under
    # --arch-interaction-op cat / --arch-interaction-itself / --arch-project-size
    import json

    variants = {"cat": ["--arch-interaction-op", "cat"], "dot_itself": ["--arch-interaction-itself"],
                "project3": ["--arch-project-size", "3"], "cat_project2_perf_debug": ["--arch-interaction-op", "cat",
                                                                                      "--arch-project-size", "2", "--perf-debug"]}
    rec = {}
    for k, (name, extra) in enumerate(variants.items()):
        vport = str(29548 + k)
        venv = dict(env, MASTER_PORT=vport)
        flags = [f for f in FLAGS if f not in ("4",)]
        flags = FLAGS[:FLAGS.index("--num-batches") + 1] + ["2"] + FLAGS[FLAGS.index("--num-batches") + 2:]
        procs = [subprocess.Popen([sys.executable, os.path.join(work, "harness.py"), "--master-ip", "127.0.0.1",
                                   "--master-port", vport] + flags + extra, cwd=work, env=dict(venv, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in (0, 1)]
        assert [p.wait(timeout=600) for p in procs] == [0, 0], name
        r0 = json.load(open(os.path.join(work, "dlrm_np2", "rank0.json")))
        rec[name] = {"extra_flags": extra, "num_batches": 2,
                     "rank0": [[c["comms"], c["msg_size"]] for c in r0 if c["comms"] == "all_reduce"]}
        print(name, rec[name]["rank0"][:4])
    json.dump(rec, open(os.path.join(HERE, "dlrm_np2_variants.json"), "w"), indent=1)
