#!/usr/bin/env python3
"""Golden outputs of the reference's OWN code for the ``--bitwidth < 32`` mode of the comms sweep (build container only;
imports /root/reference through the package alias its absolute imports expect).  Output: tests/golden/quant_rows.json.

  * ``checkQuantArgs`` (comms_utils.py:395-429): which flag combinations it refuses, with the exception text;
  * ``fixBeginSize`` with bitwidth < 32 (comms_utils.py:218-252);
  * the header ``printPreamble`` prints (comms.py:956-1001) and the row ``reportBenchTimeCollWithQuant`` prints
    (comms.py:1005-1055) for given per-rank latencies -- captured from stdout of the reference's methods;
  * the downcast the open part of the reference applies (``_downcast`` / ``_dequantize``, pytorch_dist_backend.py:48-76)
    on a seeded tensor.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _alias():
    os.makedirs("/tmp/pb", exist_ok=True)
    if not os.path.exists("/tmp/pb/param_bench"):
        os.symlink("/root/reference", "/tmp/pb/param_bench")
    for p in ("/tmp/pb", "/root/reference/train/comms/pt"):
        if p not in sys.path:
            sys.path.insert(0, p)


def main():
    _alias()
    from param_bench.train.comms.pt import comms as ref_comms
    from param_bench.train.comms.pt import comms_utils as cu
    from param_bench.train.comms.pt import pytorch_dist_backend as pdb

    out = {}
    chk = []
    for coll, dtype, begin, dim, z in [("all_to_all", "float32", 1024, 32, 1), ("all_to_allv", "float32", 1024, 32, 0),
                                       ("all_to_all", "float16", 1024, 32, 1), ("all_gather", "float32", 1024, 32, 1),
                                       ("all_reduce", "float32", 8, 32, 0), ("reduce", "float32", 8, 64, 1),
                                       ("all_to_all", "float32", 100, 32, 1), ("all_reduce", "int32", 8, 32, 1)]:
        try:
            cu.checkQuantArgs(coll, getattr(torch, dtype), begin, dim, z)
            res = None
        except Exception as e:          # noqa: BLE001 -- the exception type and text are the fixture
            res = [type(e).__name__, str(e)]
        chk.append([[coll, dtype, begin, dim, z], res])
    out["checkQuantArgs"] = chk
    fb = []
    for coll, begin, esz, world, bw, dim in [("all_to_all", 8, 4, 8, 8, 32), ("all_to_allv", 64, 4, 2, 16, 128),
                                             ("all_to_all", 1 << 20, 4, 8, 8, 256), ("all_reduce", 1, 4, 8, 16, 32)]:
        p = types.SimpleNamespace(collective=coll, beginSize=begin, element_size=esz, bitwidth=bw, quant_a2a_embedding_dim=dim)
        cu.fixBeginSize(p, world)
        fb.append([[coll, begin, esz, world, bw, dim], p.beginSize])
    out["fixBeginSize"] = fb

    # the report: reference methods on a bare instance with a stand-in for the two backend calls they make
    bench = ref_comms.commsCollBench.__new__(ref_comms.commsCollBench)
    bench.tag = ""
    bench.report = False
    bench.collectiveArgs = types.SimpleNamespace(collective="all_to_allv", data_type="float32", world_size=2)
    bench.backendFuncs = types.SimpleNamespace(tensor_list_to_numpy=lambda lst: np.array([float(t) for t in lst]))
    params = types.SimpleNamespace(bitwidth=8, backend="rccl_xgmi", collective="all_to_allv", element_size=4,
                                   use_perf_logger=None, size_start_profiler=None, data_types=["float32"],
                                   enable_profiler=False)
    rows = []
    for mem, nel, lat, ql, dl in [(1024, 128, [120.0, 100.0], [10.0, 12.0], [8.0, 9.5]),
                                  (268435456, 33554432, [5120.7, 5003.2], [401.3, 399.0], [377.7, 380.1])]:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.reportBenchTimeCollWithQuant(params, {"memSize": mem, "numElements": nel}, lat, ql, dl)
        rows.append({"memSize": mem, "numElements": nel, "lat": lat, "quant": ql, "dequant": dl, "row": buf.getvalue()})
    out["quant_rows"] = rows
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.printPreamble(params) if hasattr(bench, "printPreamble") else None
    out["preamble_stdout"] = buf.getvalue()

    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, generator=g) * 50
    out["downcast"] = {"x": x.tolist(), "16": pdb._downcast(x, 16).to(torch.float32).tolist(),
                       "8": pdb._downcast(x, 8).to(torch.float32).tolist(),
                       "dequantize_16": pdb._dequantize(pdb._downcast(x, 16)).tolist()}
    try:
        pdb._downcast(x, 4)
        out["downcast"]["4"] = None
    except NotImplementedError as e:
        out["downcast"]["4"] = str(e)
    with open(os.path.join(HERE, "quant_rows.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(os.path.getsize(os.path.join(HERE, "quant_rows.json")), "bytes")
    print(out["preamble_stdout"]); print(rows[0]["row"])


if __name__ == "__main__":
    main()
