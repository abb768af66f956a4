#!/usr/bin/env python3
"""Generates tests/golden/range_configs.json: the expansions the REFERENCE's ``RangeConfigIterator``
(train/compute/python/lib/iterator.py) produces for a set of ranged build configs of the batched EmbeddingBag operator.
Needs /root/reference (build container only); the fixture holds inputs and expected outputs (data)."""
import copy
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
work = tempfile.mkdtemp()
os.makedirs(os.path.join(work, "pb"))
os.symlink("/root/reference", os.path.join(work, "pb", "param_bench"))
sys.path.insert(0, os.path.join(work, "pb"))
from param_bench.train.compute.python.lib.iterator import DefaultConfigIterator, RangeConfigIterator  # noqa: E402

A = lambda t, n, v, r=False: dict({"type": t, "name": n, "value": v}, **({"__range__": ["value"]} if r else {}))  # noqa: E731
CASES = {
    "scalars": [{"args": [A("int", "num_tables", [1, 3, 2], True), A("int", "rows", 1000), A("int", "dim", [64, 128, 64], True),
                          A("int", "pooling", 0), A("bool", "weighted", [False, True], True),
                          A("str", "weights_precision", ["fp16", "fp32"], True)],
                 "kwargs": {"optimizer": {"type": "str", "value": ["exact_sgd", "exact_row_wise_adagrad"], "__range__": ["value"]},
                            "lr": {"type": "float", "value": 0.05}}}],
    "two_variants": [{"args": [A("int", "num_tables", 1), A("int", "rows", [100, 300, 100], True), A("int", "dim", 32)]},
                     {"args": [A("int", "num_tables", 2), A("int", "rows", 5), A("int", "dim", [8, 16, 8], True)],
                      "kwargs": {"eps": {"type": "float", "value": 1e-8}}}],
    "genericlist": [{"args": [A("int", "num_tables", 2),
                              {"type": "genericlist", "name": "rows", "__range__": ["value"],
                               "value": [A("int", "r0", [10, 30, 10], True), A("int", "r1", 7)]},
                              {"type": "genericlist", "name": "dims", "value": [A("int", "d0", 8), A("int", "d1", 16)]}]}],
    "no_ranges": [{"args": [A("int", "num_tables", 1), A("int", "rows", 228582), A("int", "dim", 128)],
                   "kwargs": {"optimizer": {"type": "str", "value": "exact_row_wise_adagrad"}}}],
}


def values(cfg):
    def val(a):
        v = a["value"]
        if a.get("type") in ("genericlist", "tuple"):
            return [val(x) for x in v]
        return v
    return {"args": [val(a) for a in cfg.get("args", [])], "kwargs": {k: val(a) for k, a in cfg.get("kwargs", {}).items()}}


out = {}
for name, variants in CASES.items():
    exp = [[i, values(copy.deepcopy(c))] for i, c in RangeConfigIterator({"build": copy.deepcopy(variants)}, "build", "cpu")]
    dflt = [[i, values(c)] for i, c in DefaultConfigIterator({"build": copy.deepcopy(variants)}, "build", "cpu")] \
        if name == "no_ranges" else None
    out[name] = {"variants": variants, "range_iterator": exp, "default_iterator": dflt}
json.dump(out, open(os.path.join(HERE, "range_configs.json"), "w"), indent=1)
print({k: len(v["range_iterator"]) for k, v in out.items()})
