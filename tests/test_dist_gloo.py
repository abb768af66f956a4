"""world_size-2 tests of the N>1 path on CPU (gloo over loopback): the backend plug-in, the
collective sweep driver and the lookup -> all-to-all pipeline layout.  8-GPU runs are the
driver's; these make the distributed path correct by construction."""
import json
import os
import re

import pytest
import torch.multiprocessing as mp

from tests import dist_workers as W

WORLD = 2


def _spawn(fn, *extra):
    mp.spawn(fn, args=(WORLD, W.free_port()) + extra, nprocs=WORLD, join=True)


def test_backend_collectives_over_gloo():
    _spawn(W.backend_collectives)


def test_pipeline_layout_over_gloo():
    _spawn(W.pipeline_layout)


def test_comms_sweep_driver_over_gloo(tmp_path, golden_dir):
    _spawn(W.comms_sweep, str(tmp_path))
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    res = r0["results"]
    # 3 collectives x sizes {64, 256, 1024}
    assert [(r["collective"], r["memSize"]) for r in res] == [
        (c, s) for c in ("all_to_allv", "all_to_all", "all_reduce") for s in (64, 256, 1024)]
    for r in res:
        n = r["world_size"]
        factor = 2 * (n - 1) / n if r["collective"] == "all_reduce" else (n - 1) / n
        assert r["busBW_GBps"] == pytest.approx(r["algBW_GBps"] * factor, rel=1e-9)
        assert r["algBW_GBps"] == pytest.approx(r["memSize"] / (r["p50_us"] * 1e3), rel=1e-9)  # bytes/ns
        assert r["min_us"] <= r["p50_us"] <= r["p95_us"] <= r["max_us"]
    # both ranks computed the same report; only rank 0 printed it
    assert [x["p50_us"] for x in r1["results"]] == [x["p50_us"] for x in res]
    rows0 = [ln for ln in r0["stdout"].splitlines() if ln.startswith("\tCOMMS-RES-") and "total-size" not in ln]
    assert len(rows0) == 9 and "COMMS-RES" not in r1["stdout"].replace("[Rank", "")
    # row format identical to the reference's (fixture generated from the reference's format string)
    gold = json.load(open(os.path.join(golden_dir, "comms_pure.json")))["row_fmt"]
    shape = lambda s: re.sub(r"[0-9.]+", "#", s)
    widths = lambda s: [len(x) for x in re.findall(r"\s*\S+", s)]
    assert widths(rows0[1].replace("all_to_allv", "all_to_all")) == widths(gold)
    assert shape(rows0[1]).startswith("\tCOMMS-RES-all_to_allv-float#")


def test_dlrm_sparse_path_over_gloo():
    _spawn(W.dlrm_sparse_path)


def test_dlrm_driver_over_gloo_print_comms(tmp_path, golden_dir):
    """Same flags as the REFERENCE run that produced tests/golden/dlrm_np2 (gen_dlrm_np2.py: the reference's dlrm.py,
    2 gloo ranks, here): with ``--data-generation random`` (the reference's NumPy draw sequence, seed = rank) every
    one of the 28 per-rank --print-comms records -- collective order, msg_size, in/out splits incl. the data-dependent
    index exchange, dtype, over all 4 batches -- equals the reference's, on both ranks."""
    _spawn(W.dlrm_driver, str(tmp_path))
    for r in (0, 1):
        mine = json.load(open(tmp_path / "dlrm_np2" / f"rank{r}.json"))
        gold = json.load(open(os.path.join(golden_dir, "dlrm_np2", f"rank{r}.json")))
        assert len(gold) == 28 and mine == gold, r
    rec = json.load(open(tmp_path / "dlrm_np2" / "rank0.json"))
    assert [r["comms"] for r in rec[:7]] == ["all_to_all", "all_to_all", "all_to_all", "all_reduce", "all_reduce",
                                             "all_to_all", "all_reduce"]
    assert rec[1] == {"comms": "all_to_all", "msg_size": 664, "in_split": [39, 34], "out_split": [39, 44], "dtype": "torch.int64"}
    rep = json.load(open(tmp_path / "report0.json"))
    assert set(n for n in rep["report"] if not n.endswith("_bw")) == {
        "intermed_calc_length", "mem_push_idx", "intermed_bef_offset_xchg", "offset_xchg", "intermed_btw_offset_idx_xchg",
        "idx_xchg", "intermed_post_idx_xchg_sparse_dist", "intermed_emb_lookup_to_a2a_start", "fwd_a2a",
        "intermed_fwd_a2a_grad_push", "mem_push_gradients", "bwd_top_ar", "intermed_top_ar_end_to_bwd_a2a_start", "bwd_a2a",
        "intermed_bwd_a2a_bot_ar", "bwd_bot_ar", "iter_time", "iter_data_prep", "iter_fwd_a2a", "iter_bwd_top_ar", "iter_bwd_a2a"}
    assert rep["report"]["fwd_a2a"]["p50"] > 0 and "fwd_a2a" in rep["stdout"]
    assert rep["report"]["fwd_a2a_bw"]["busBW_GBps"] == pytest.approx(rep["report"]["fwd_a2a_bw"]["algBW_GBps"] / 2)


@pytest.mark.parametrize("blocking", [False, True])
def test_trace_replay_over_gloo(tmp_path, golden_dir, blocking):
    """the collective entries of the golden basic trace replayed on 2 ranks: non-blocking (wait entries resolve the
    recorded request ids) and blocking with the ones-in / expected-out data check"""
    _spawn(W.trace_replay, str(tmp_path), golden_dir, blocking)
    for r in (0, 1):
        rec = json.load(open(tmp_path / "perf" / f"replayedCommsPerf.rank{r}.json"))
        names = [x["comms"] for x in rec]
        per_replay = ["all_to_all_single", "wait", "all_to_allv", "wait", "all_to_allv", "wait", "all_reduce", "wait",
                      "all_reduce", "barrier"]
        if blocking:                                   # blocking replay skips wait entries (a barrier follows every op)
            per_replay = [n for n in per_replay if n != "wait"]
        assert names == per_replay * 2
        a2av = [x for x in rec if x["comms"] == "all_to_allv"][0]
        assert a2av["in_split"] == ([39, 34] if r == 0 else [44, 40]) and a2av["dtype_size"] == 8
        assert all(x["latency_us"] > 0 and x["global_latency_us"] >= x["latency_us"] for x in rec)
        s = json.load(open(tmp_path / f"summary{r}.json"))
        assert s["collLat"]["all_to_allv"] == 4 and s["collLat"]["all_reduce"] == 4 and s["total_us"] > 0
        assert ("Replayed 4 all_to_allv" in s["stdout"]) == (r == 0)     # rank 0 reports


def test_dlrm_helpers_match_reference_goldens(golden_dir):
    import torch

    from param_amd.comms.pt import dlrm as D_

    g = json.load(open(os.path.join(golden_dir, "comms_pure.json")))
    for (n, r, w), v in g["get_split_lengths_by_len"]:
        my, splits = D_.get_split_lengths_by_len(n, r, w)
        assert [my, splits] == v
    for (r, per, w), v in g["get_slice_sparse"]:
        s = D_.get_slice_sparse(r, per, w)
        assert [s.start, s.stop, s.step] == v
    for lens, v in g["lengthsToOffsets"]:
        assert D_.lengthsToOffsets(torch.tensor(lens)).tolist() == v
    c = g["calculateLengths"]
    offs = [torch.tensor(o) for o in c["offsets"]]
    idxs = [torch.arange(7), torch.arange(10, 16)]
    ln, ix = D_.calculateLengths(2, offs, idxs)
    assert ln.tolist() == c["lengths"] and ix.tolist() == c["indices"]
    s = g["splitPerTable"]
    o, i = D_.splitPerTable(torch.tensor(s["lengths"]), torch.tensor(s["indices"]), s["batch"], s["features"], s["world"])
    assert [x.tolist() for x in o] == s["offsets_out"] and [x.tolist() for x in i] == s["indices_out"]


def test_harness_arithmetic_matches_reference_goldens(golden_dir):
    """pure functions vs tests/golden/comms_pure.json (generated by importing the reference)"""
    import types

    from param_amd.comms.pt import comms_utils as cu
    from param_amd.comms.pt.mi355_backend import MI355XBackend

    g = json.load(open(os.path.join(golden_dir, "comms_pure.json")))
    for s, v in g["parsesize"]:
        assert cu.parsesize(s) == v
    for s, v in g["parseRankList"]:
        assert cu.parseRankList(s) == v
    for a, v in g["getAlgBW"]:
        assert list(cu.getAlgBW(*a)) == v
    for a, v in g["getSizes"]:
        assert cu.getSizes(*a) == v
    for (coll, begin, esz, world), v in g["fixBeginSize"]:
        p = types.SimpleNamespace(collective=coll, beginSize=begin, element_size=esz, bitwidth=32,
                                  quant_a2a_embedding_dim=0)
        cu.fixBeginSize(p, world)
        assert p.beginSize == v
    for (coll, bw, n), v in g["getBusBW"]:
        assert MI355XBackend.getBusBW(None, coll, bw, types.SimpleNamespace(world_size=n)) == v
    from param_amd.comms.pt import comms

    assert comms.format_row("all_to_all", "float32", "", 1024, 32, 12.34, 13.0, 14.5, 11.0, 15.0, 0.083, 0.073) == g["row_fmt"]
