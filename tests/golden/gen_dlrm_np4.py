#!/usr/bin/env python3
"""Generates tests/golden/dlrm_np4/rank{0..3}.json: the ``--print-comms`` records of the REFERENCE's ``train/comms/pt/dlrm.py`` on
4 gloo ranks with SIX tables (and dlrm_np8: 8 gloo ranks, 26 tables -> [4, 4, 3, 3, 3, 3, 3, 3], fixed pooling) -- the uneven partition [2, 2, 1, 1] (dlrm.py:390-398), so the pooled all-to-all's splits differ
per peer -- and ragged bags (no --num-indices-per-lookup-fixed).  Same harness as gen_dlrm_np2.py (it pre-registers the
``--use-device-time`` flag the reference forgets).  Needs /root/reference; build container only."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_dlrm_np2 import HARNESS  # noqa: E402

FLAGS = ["--backend", "gloo", "--device", "cpu", "--mini-batch-size", "4", "--num-batches", "3", "--warmup-batches", "1",
         "--arch-mlp-bot", "8-4", "--arch-mlp-top", "4-1", "--arch-sparse-feature-size", "4",
         "--arch-embedding-size", "100-200-300-400-500-600", "--num-indices-per-lookup", "6", "--print-comms"]
# BASELINE configs[3]'s shape in small: 26 tables over 8 ranks -> [4, 4, 3, 3, 3, 3, 3, 3], fixed pooling (the reference's flag is
# ``type=bool``: it takes a value)
FLAGS8 = ["--backend", "gloo", "--device", "cpu", "--mini-batch-size", "2", "--num-batches", "2", "--warmup-batches", "1",
          "--arch-mlp-bot", "8-4", "--arch-mlp-top", "4-1", "--arch-sparse-feature-size", "4",
          "--arch-embedding-size", "-".join(str(50 + t) for t in range(26)), "--num-indices-per-lookup", "3",
          "--num-indices-per-lookup-fixed", "True", "--print-comms"]

def run(world, flags, port, name):
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pb"))
    os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
    open(os.path.join(work, "harness.py"), "w").write(HARNESS)
    env = dict(os.environ, PYTHONPATH=f"{work}/pb:/root/reference/train/comms/pt", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
               WORLD_SIZE=str(world), LOCAL_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(work, "harness.py"), "--master-ip", "127.0.0.1", "--master-port", port]
                              + flags, cwd=work, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.DEVNULL,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = [p.communicate(timeout=900)[1] for p in procs]
    # a rank may die in gloo's teardown AFTER writing its records ("terminate called without an active exception": the reference
    # never destroys its process group); what counts is that every rank wrote its file
    missing = [r for r in range(world) if not os.path.exists(os.path.join(work, name, f"rank{r}.json"))]
    assert not missing, (missing, [e[-1500:] for p, e in zip(procs, errs) if p.returncode])
    out = os.path.join(HERE, name)
    os.makedirs(out, exist_ok=True)
    for r in range(world):
        rec = json.load(open(os.path.join(work, name, f"rank{r}.json")))
        json.dump(rec, open(os.path.join(out, f"rank{r}.json"), "w"))
        print(name, r, len(rec), rec[2])
    json.dump({"flags": flags}, open(os.path.join(out, "flags.json"), "w"))


if __name__ == "__main__":
    run(8, FLAGS8, "29877", "dlrm_np8")
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pb"))
    os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
    open(os.path.join(work, "harness.py"), "w").write(HARNESS)
    port = "29561"
    env = dict(os.environ, PYTHONPATH=f"{work}/pb:/root/reference/train/comms/pt", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
               WORLD_SIZE="4", LOCAL_SIZE="4")
    procs = [subprocess.Popen([sys.executable, os.path.join(work, "harness.py"), "--master-ip", "127.0.0.1", "--master-port", port]
                              + FLAGS, cwd=work, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL) for r in range(4)]
    assert [p.wait(timeout=600) for p in procs] == [0] * 4
    out = os.path.join(HERE, "dlrm_np4")
    os.makedirs(out, exist_ok=True)
    for r in range(4):
        rec = json.load(open(os.path.join(work, "dlrm_np4", f"rank{r}.json")))
        json.dump(rec, open(os.path.join(out, f"rank{r}.json"), "w"))
        print(r, len(rec), rec[2])
    json.dump({"flags": FLAGS}, open(os.path.join(out, "flags.json"), "w"))
