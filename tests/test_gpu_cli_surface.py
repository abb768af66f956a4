"""GPU tests of the command-line surface added at the end of round 4 (written without a GPU, gated until round 5's first visit
showed them green on a 1-rank RCCL group: gpurun visit r5_v1, 4 passed; the gate is gone).  Each mirrors a CPU test that passes on
gloo ranks."""
import contextlib
import io
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _port():
    from tests.dist_workers import free_port
    return free_port()


def _clean_env():
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_SIZE"):
        os.environ.pop(k, None)


def test_whole_collective_table_with_validation_single_gpu(tmp_path):
    """every collective of the backend table on the 1-rank RCCL group, ``--c 1`` validating the outputs, perf logger records"""
    from param_amd.comms.pt import comms

    _clean_env()
    colls = ["all_gather", "all_gather_base", "reduce_scatter", "reduce_scatter_base", "broadcast", "reduce", "gather", "scatter",
             "all_to_all", "all_to_allv", "all_to_all_single", "all_reduce"]
    os.environ["PARAM_PERF_LOG"] = str(tmp_path / "perf.jsonl")
    try:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = comms.main(["--master-ip", "127.0.0.1", "--master-port", str(_port()), "--device", "rocm", "--z", "1", "--c", "1",
                              "--b", "1K", "--e", "1M", "--f", "32", "--n", "5", "--w", "2", "--collective", ",".join(colls),
                              "--use-perf-logger", "jsonl", "--tag", "r5"])
    finally:
        os.environ.pop("PARAM_PERF_LOG")
    assert [(r["collective"], r["memSize"]) for r in res] == [(c, s) for c in colls for s in (1 << 10, 32 << 10, 1 << 20)]
    assert all(r["p50_us"] > 0 for r in res) and buf.getvalue().count("COMMS-RES-") == 36
    recs = [json.loads(ln) for ln in open(tmp_path / "perf.jsonl")]
    assert [r["commsOp"] for r in recs] == [c for c in colls for _ in range(3)] and all(r["Tags"] == "-r5" for r in recs)


def test_sizes_dtypes_and_graph_replay_of_the_new_collectives_single_gpu():
    from param_amd.comms.pt import comms

    _clean_env()
    with contextlib.redirect_stdout(io.StringIO()):
        res = comms.main(["--master-ip", "127.0.0.1", "--master-port", str(_port()), "--device", "rocm", "--z", "1", "--c", "1",
                          "--ss", "4096,65536,256", "--data-types", "float32,bfloat16,int32", "--n", "5", "--w", "2",
                          "--collective", "all_to_all_single,all_reduce"])
        gr = comms.main(["--master-ip", "127.0.0.1", "--master-port", str(_port()), "--device", "rocm", "--z", "1", "--c", "1",
                         "--b", "1K", "--e", "64K", "--f", "8", "--n", "10", "--w", "2", "--graph-launches", "5",
                         "--collective", "all_gather_base,reduce_scatter_base,broadcast,all_to_allv"])
    assert [r["memSize"] for r in res] == [4096, 65536, 256] * 6 and [r["dtype"] for r in res][::6] == ["float32", "bfloat16", "int32"]
    assert len(gr) == 12 and all(r["p50_us"] > 0 for r in gr)


def test_overlap_bench_reference_rows_single_gpu():
    """commsComputeBench.py: the reference-format header and rows beside the build's own line; device times in the two extra columns"""
    from param_amd.comms.pt import commsComputeBench as C

    _clean_env()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = C.main(["--master-ip", "127.0.0.1", "--master-port", str(_port()), "--b", "1M", "--e", "4M", "--f", "4", "--n", "4", "--w", "2",
                      "--collective", "all_to_allv", "--device", "rocm", "--kernel", "emb_lookup", "--num-compute", "3", "--ntables", "8",
                      "--num-embs", "200000", "--emb-dim", "128", "--batch-size", "2048", "--bag-size", "20", "--tag", "ov"])
    out = buf.getvalue()
    assert C.format_cc_header("comms-compute", 32) in out
    rows = [ln.split() for ln in out.splitlines() if ln.startswith("\tCOMMS-RES-all_to_allv-float32-ov")]
    assert [int(r[1]) for r in rows] == [1 << 20, 4 << 20] and all(len(r) == 12 for r in rows)
    for r, rec in zip(rows, res):
        assert float(r[11]) == pytest.approx(rec["compute_dev_us"], abs=0.06) and float(r[10]) == pytest.approx(rec["timeUS"], abs=0.06)
        assert rec["report"]["p50"] == pytest.approx(rec["comm_dev_us"], rel=1e-6)


def test_dlrm_driver_report_and_flags_single_gpu(tmp_path):
    from param_amd.comms.pt import dlrm

    _clean_env()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            rep = dlrm.main(["--master-ip", "127.0.0.1", "--master-port", str(_port()), "--device", "rocm", "--mini-batch-size", "256",
                             "--num-batches", "6", "--warmup-batches", "2", "--arch-mlp-bot", "64-32", "--arch-mlp-top", "32-1",
                             "--arch-sparse-feature-size", "128", "--arch-embedding-size", "20000-30000-40000-50000",
                             "--num-indices-per-lookup", "20", "--num-indices-per-lookup-fixed", "True", "--perf-debug",
                             "--arch-interaction-op", "cat", "--print-comms"])
    finally:
        os.chdir(cwd)
    out = buf.getvalue()
    assert rep["fwd_a2a"]["memory"] == 256 * 4 * 128 * 4 and rep["bwd_top_ar"]["memory"] == 4 * (5 * 32 * 32 + 32)
    assert out.count("\tintermed_calc_length") == 0 and out.count("intermed_calc_length") == 2 and out.count("total_time") == 2
    assert "\t ln_top: [160  32   1] " in out
