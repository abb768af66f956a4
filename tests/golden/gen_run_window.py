#!/usr/bin/env python3
"""Generates tests/golden/run_window.json: for a stream of run ids and several ``--resume-id`` / ``--stop_id`` pairs, the
state (skip / run / stop) the REFERENCE's ``BuildExecutor.get_transition_state`` returns id by id
(train/compute/python/lib/pytorch/build_executor.py:72-102).  Needs /root/reference (build container only)."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
work = tempfile.mkdtemp()
os.makedirs(os.path.join(work, "pb"))
os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
sys.path.insert(0, os.path.join(work, "pb"))
from param_bench.train.compute.python.lib.pytorch.build_executor import BuildExecutor  # noqa: E402


class Probe(BuildExecutor):
    def run(self, *a, **k):
        pass


IDS = [f"op|{c}|0_{b}|0_{i}" for c in range(2) for b in range(2) for i in range(2)]
CASES = [(None, None), ("op|0|0_1|0_0", None), (None, "op|1|0_0|0_1"), ("op|0|0_1|0_1", "op|1|0_1|0_0"),
         ("op|1|0_0|0_0", "op|1|0_0|0_0"), ("nothing", None), ("op|1|0_0|0_0", "op|0|0_0|0_1")]
out = {"ids": IDS, "cases": []}
for resume, stop in CASES:
    p = Probe()
    p.set_resume_op_run_id(resume)
    p.set_stop_op_run_id(stop)
    out["cases"].append({"resume": resume, "stop": stop, "states": [p.get_transition_state(i).name.lower() for i in IDS]})
json.dump(out, open(os.path.join(HERE, "run_window.json"), "w"), indent=1)
for c in out["cases"]:
    print(c)
