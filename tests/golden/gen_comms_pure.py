#!/usr/bin/env python3
"""Golden I/O of the reference's PURE harness functions on the all-to-all / DLRM path
(build container only; imports /root/reference through the package alias the reference's
absolute imports expect).  Output: tests/golden/comms_pure.json (data only)."""
from __future__ import annotations

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


def gen_comms_pure():
    _alias()
    from param_bench.train.comms.pt import comms_utils as cu
    from param_bench.train.comms.pt import pytorch_backend_utils as bu
    import dlrm as ref_dlrm

    out = {}
    out["parsesize"] = [[s, cu.parsesize(s)] for s in ["8", "4K", "256M", "1G", "1024", "16M", 4096]]
    out["parseRankList"] = [[s, cu.parseRankList(s)] for s in ["3", "0,2,2,5", "2:6", ""]]
    out["getAlgBW"] = [[a, list(cu.getAlgBW(*a))] for a in
                       [[10000.0, 90000, 1], [1e6, 1 << 20, 10], [0.0, 100, 5], [5e5, 109051904, 0], [123456.0, 109051904, 3]]]
    out["getSizes"] = [[a, cu.getSizes(*a)] for a in [[8, 1024, 2, 0], [64, 1024, 4, 0], [8, 64, 2, 8], [1, 1 << 40, 2, 0]]]
    fb = []
    for coll, begin, esz, world in [("all_to_all", 8, 4, 8), ("all_to_allv", 64, 4, 8), ("all_to_all_single", 4, 8, 2),
                                    ("all_reduce", 1, 4, 8), ("all_gather", 2, 2, 4), ("reduce", 16, 4, 4)]:
        p = types.SimpleNamespace(collective=coll, beginSize=begin, element_size=esz, bitwidth=32,
                                  quant_a2a_embedding_dim=0)
        cu.fixBeginSize(p, world)
        fb.append([[coll, begin, esz, world], p.beginSize])
    out["fixBeginSize"] = fb
    bb = []
    for coll in ["all_to_all", "all_to_allv", "all_to_all_single", "all_reduce", "reduce", "all_gather", "broadcast"]:
        for n in (1, 2, 4, 8):
            ca = types.SimpleNamespace(world_size=n)
            bb.append([[coll, 100.0, n], bu.backendFunctions.getBusBW(None, coll, 100.0, ca)])
    out["getBusBW"] = bb

    net = ref_dlrm.paramDLRM_Net
    out["get_split_lengths_by_len"] = [[[n, r, w], list(net.get_split_lengths_by_len(None, n, r, w))]
                                       for n, r, w in [(26, 3, 8), (26, 0, 8), (64, 5, 8), (8, 1, 2), (7, 6, 7), (13, 2, 4)]]
    gs = []
    for per_rank, w in [([4, 4, 3, 3, 3, 3, 3, 3], 8), ([2, 1], 2)]:
        for r in range(w):
            s = net.get_slice_sparse(None, r, per_rank, w)
            gs.append([[r, per_rank, w], [s.start, s.stop, s.step]])
    out["get_slice_sparse"] = gs
    lo = []
    for lens in [[3, 0, 2, 5], [1], [0, 0, 4]]:
        lo.append([lens, ref_dlrm.lengthsToOffsets(torch.tensor(lens), "cpu").tolist()])
    out["lengthsToOffsets"] = lo
    # calculateLengths: offsets/indices per feature -> concatenated lengths / indices
    offs = [torch.tensor([0, 2, 2, 5]), torch.tensor([0, 1, 4, 4])]
    idxs = [torch.arange(7), torch.arange(10, 16)]
    ln, ix = ref_dlrm.calculateLengths(2, offs, idxs)
    out["calculateLengths"] = {"offsets": [o.tolist() for o in offs], "n_indices": [len(i) for i in idxs],
                               "lengths": ln.tolist(), "indices": ix.tolist()}
    # splitPerTable: received [rank][table][batch] lengths + concatenated indices -> per-table offsets/indices
    rng = np.random.default_rng(0)
    W, F, B = 3, 2, 4
    lengths = torch.tensor(rng.integers(0, 4, W * F * B))
    indices = torch.arange(int(lengths.sum())) * 7 % 101
    o, i = net.splitPerTable(None, lengths, indices, B, F, W, 0, "cpu")
    out["splitPerTable"] = {"world": W, "features": F, "batch": B, "lengths": lengths.tolist(),
                            "indices": indices.tolist(), "offsets_out": [x.tolist() for x in o],
                            "indices_out": [x.tolist() for x in i]}
    # report formats (comms.py:989-1001,1151-1168)
    out["row_fmt"] = "\tCOMMS-RES-{}-{}{}{:>18}{:>18}{:>18}{:>12}{:>12}{:>12}{:>12}{:>15}{:>12}{:>20}".format(
        "all_to_all", "float32", "", 1024, "32", "%.1f" % 12.34, "%.1f" % 13.0, "%.1f" % 14.5, "%.1f" % 11.0,
        "%.1f" % 15.0, "%.3f" % 0.083, "%.3f" % 0.073, "%.1f" % 0.0)
    with open(os.path.join(HERE, "comms_pure.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    gen_comms_pure()
    print(os.path.getsize(os.path.join(HERE, "comms_pure.json")), "bytes")
