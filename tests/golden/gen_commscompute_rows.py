#!/usr/bin/env python3
"""Generates tests/golden/commscompute_rows.json: header lines and report rows the REFERENCE's ``commsComputeBench.py`` prints
(``printPreamble`` ``:361-433``, ``reportBenchTimeColl`` ``:499-640``) for fixed latencies, per mode (comms-compute / compute)
and kernel (emb_lookup): reference methods on a bare instance with stand-ins for the two backend calls they make.
Needs /root/reference (build container only)."""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    os.makedirs("/tmp/pb", exist_ok=True)
    if not os.path.exists("/tmp/pb/param_bench"):
        os.symlink("/root/reference", "/tmp/pb/param_bench")
    sys.path.insert(0, "/tmp/pb")
    from param_bench.train.comms.pt import commsComputeBench as ref
    from param_bench.train.comms.pt.pytorch_backend_utils import backendFunctions

    out = {"headers": {}, "rows": []}
    bench = ref.commsComputeBench.__new__(ref.commsComputeBench)
    bench.tag = "-tg"
    bench.collectiveArgs = types.SimpleNamespace(collective="all_to_allv", data_type="float32", world_size=4, numComputePerIter=3)
    bench.backendFuncs = types.SimpleNamespace(
        tensor_list_to_numpy=lambda lst: np.array([float(t) for t in lst]),
        getBusBW=lambda coll, bw, ca: backendFunctions.getBusBW(None, coll, bw, ca))
    for mode in ("comms-compute", "compute"):
        for bitwidth in (32, 8):
            params = types.SimpleNamespace(kernel="emb_lookup", mode=mode, bitwidth=bitwidth, backend="rccl_xgmi",
                                           mm_dim=[0, 0, 0], collective="all_to_allv")
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                bench.printPreamble(params)
            out["headers"][f"{mode}/{bitwidth}"] = buf.getvalue()
    for mode, lat, comm, comp in (("comms-compute", [900.0, 1000.0, 1100.0, 950.0], [400.0, 420.0, 380.0, 410.0], [700.0, 800.0, 750.0, 720.0]),
                                  ("compute", [900.0, 1000.0, 1100.0, 950.0], [], [])):
        params = types.SimpleNamespace(kernel="emb_lookup", mode=mode, bitwidth=32, backend="rccl_xgmi", mm_dim=[0, 0, 0],
                                       collective="all_to_allv")
        res = {"memSize": 4194304, "numElements": 262144, "timeUS": 1000.0}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.reportBenchTimeColl(params, dict(res), lat, comm, comp)
        out["rows"].append({"mode": mode, "results": res, "lat": lat, "comm": comm, "comp": comp, "world_size": 4, "tag": "-tg",
                            "row": buf.getvalue()})
    json.dump(out, open(os.path.join(HERE, "commscompute_rows.json"), "w"), indent=1)
    for k, v in out["headers"].items():
        print(k, repr(v))
    for r in out["rows"]:
        print(repr(r["row"]))


if __name__ == "__main__":
    main()
