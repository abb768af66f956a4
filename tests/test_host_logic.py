"""Host-side mirror of the reference's train/compute/pt surface (CPU only)."""
import contextlib
import io
import json
import os
import types

import numpy as np
import pytest
import torch

from param_amd import indices as I
from param_amd.compute.pt import dataset, driver, pytorch_emb


def test_init_indices_matches_reference_goldens(golden_dir):
    d = np.load(os.path.join(golden_dir, "init_indices.npz"))
    seen = 0
    for k in d.files:
        if k.startswith("uniform_s"):
            seed = int(k.split("_s")[1])
            torch.manual_seed(seed)
            np.random.seed(seed)
            got = I.init_indices(0.0, 1000000, 512, 20)
        elif k.startswith("zipf_"):
            p = k.split("_")
            alpha, f, b, n, seed = float(p[1][1:]), int(p[2][1:]), int(p[3][1:]), int(p[4][1:]), int(p[5][1:])
            torch.manual_seed(seed)
            np.random.seed(seed)
            got = I.init_indices(alpha, f, b, n)
        else:
            continue
        assert got.dtype == torch.int64 and np.array_equal(got.numpy(), d[k]), k
        seen += 1
    assert seen >= 6


def test_init_indices_alpha_string_and_underfill():
    torch.manual_seed(0)
    np.random.seed(0)
    a = I.init_indices("1.05", 5000, 4, 4)  # reference bug R2: string alpha crashes there
    assert a.shape == (16,)
    np.random.seed(0)
    with pytest.raises(ValueError, match="distinct"):  # reference bug R4: broadcast error there
        I.init_indices(3.0, 50, 8, 20)


def test_zipf_indices_vectorised_properties():
    g = torch.Generator().manual_seed(1)
    z = I.zipf_indices(1.05, 200000, 512, 20, generator=g).view(512, 20)
    assert z.dtype == torch.int64 and int(z.min()) >= 0 and int(z.max()) < 200000
    assert all(len(set(r.tolist())) == 20 for r in z)          # per-bag sampling without replacement
    # hot rows are the LOW row ids (unpermuted, like the reference): row 0 is in most bags
    assert (z == 0).any(dim=1).float().mean() > 0.8
    assert (z < 1000).float().mean() > 0.3
    g2 = torch.Generator().manual_seed(1)
    assert torch.equal(I.zipf_indices(1.05, 200000, 512, 20, generator=g2).view(512, 20), z)
    u = I.zipf_indices(0.0, 1000, 64, 5)
    assert u.shape == (320,)
    # tiny table where every bag under-fills at first: redraw loop terminates, still distinct
    z = I.zipf_indices(1.2, 64, 32, 16, generator=torch.Generator().manual_seed(0)).view(32, 16)
    assert all(len(set(r.tolist())) == 16 for r in z)


def test_zipf_indices_follow_the_reference_generators_distribution():
    """The bench-scale generator against the reference-compatible one (``init_indices``: bit-identical to the reference's for
    fixed seeds, tests/test_oracle.py): same sampling scheme => same distribution.  Two-sample comparison of the per-row
    frequencies of 4096 bags x 8 lookups over 5000 rows: every one of the 30 hottest rows within 5 binomial sigmas, the
    Kolmogorov-Smirnov distance of the row distributions below 0.02 (two samples of 32 768 draws from one law: ~0.008)."""
    features, batch, nnz = 5000, 4096, 8
    np.random.seed(123)
    ref = I.init_indices(1.05, features, batch, nnz).numpy()
    mine = I.zipf_indices(1.05, features, batch, nnz, generator=torch.Generator().manual_seed(7)).numpy()
    n = ref.size
    assert mine.size == n
    cr = np.bincount(ref, minlength=features).astype(np.float64)
    cm = np.bincount(mine, minlength=features).astype(np.float64)
    for r in range(30):
        p = (cr[r] + cm[r]) / (2 * n)
        sigma = np.sqrt(2 * n * p * (1 - p))            # of the difference of two binomial counts
        assert abs(cr[r] - cm[r]) <= 5 * sigma + 1, (r, cr[r], cm[r])
    ks = np.abs(np.cumsum(cr) / n - np.cumsum(cm) / n).max()
    assert ks < 0.02, ks
    # and both are per-bag samples without replacement
    assert all(len(set(b.tolist())) == nnz for b in mine.reshape(batch, nnz)[:64])
    assert all(len(set(b.tolist())) == nnz for b in ref.reshape(batch, nnz)[:64])


def test_tbe_request_layout():
    idx, off = I.tbe_request([100, 200, 50], batch=4, pooling=3, alpha=0.0, seed=5)
    assert idx.shape == (36,) and off.tolist() == list(range(0, 37, 3))
    assert int(idx[:12].max()) < 100 and int(idx[12:24].max()) < 200 and int(idx[24:].max()) < 50
    idx32, off32 = I.tbe_request([100], 2, 2, index_dtype=torch.int32)
    assert idx32.dtype == off32.dtype == torch.int32
    assert I.fixed_offsets(4, 7).tolist() == [0, 7, 14, 21]
    assert I.fixed_offsets(2, 3, include_last=True).tolist() == [0, 3, 6]


def test_cpu_driver_row_format_matches_reference(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "emb_rows.json")))
    args = types.SimpleNamespace(device="cpu", randomseed=0, warmups=1, steps=2, alpha=0.0, usexlabag=False)
    buf = io.StringIO()
    torch.set_num_threads(1)
    with contextlib.redirect_stdout(buf):
        pytorch_emb.run(args, [(r["features"], r["embdim"], r["nnz"], r["batch"]) for r in gold["rows"]])
    lines = buf.getvalue().splitlines()
    assert lines[:3] == gold["header"]
    for ln, r in zip(lines[3:], gold["rows"]):
        assert ln.startswith(r["raw_prefix"]) and len(ln) == r["len"]
        assert [c.strip() for c in ln.split(",")][5] == r["data_mb"]


def test_bytes_metric_and_algorithmic_bytes():
    # PARAM metric: batch*nnz*embdim*elem (pytorch_emb.py:180); SURVEY 8d: 546.0 B/lookup at D=128 f32 L=20
    per_lookup = pytorch_emb.algorithmic_bytes(64, 8192, 20, 128, 4) / (64 * 8192 * 20)
    assert abs(per_lookup - 546.0) < 1e-9
    per_lookup = pytorch_emb.algorithmic_bytes(64, 8192, 20, 128, 2) / (64 * 8192 * 20)
    assert abs(per_lookup - 290.0) < 1e-9


def test_cli_surface_kept():
    a = pytorch_emb.build_parser().parse_args(
        "--features 1000000 --embdim 32 --nnz 20 --batch 512 --steps 3 --warmups 1 --randomseed 1 "
        "-t float32 -d cpu --usexlabag --alpha 1.05".split())
    assert (a.features, a.embdim, a.nnz, a.batch, a.alpha, a.device) == (1000000, 32, 20, 512, 1.05, "cpu")
    d = driver.build_parser().parse_args("--warmups 2 --steps 5 --device cpu emb -d B --randomseed 3 --alpha 1.2".split())
    assert d.kernel == "emb" and d.dataset == "B" and d.alpha == 1.2 and d.steps == 5
    assert dataset.emb_A[0] == (14000000, 128, 30, 512) and dataset.emb_A[-1] == (26000000, 128, 30, 65536)
    assert len(dataset.emb_A) == 16 and dataset.emb_B[0] == (4800000, 56, 34, 2048) and len(dataset.emb_B) == 6


def test_gpu_modules_refuse_cpu_tensors():
    """No CPU fallback: the product modules raise on host tensors."""
    from param_amd import BatchedEmbeddingBagMI355, EmbeddingBagMI355, fill_random_

    m = EmbeddingBagMI355(10, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.tensor([1, 2]), torch.tensor([0]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fill_random_(torch.empty(8))
    with pytest.raises(RuntimeError):
        BatchedEmbeddingBagMI355([10], 8, device="cpu", init=None).lookup(torch.tensor([1]), torch.tensor([0, 1]))


def test_operator_registry_contract():
    """register_operator semantics of the reference (lib/operator.py:48-54; test_register.py:22-60 there)"""
    from param_amd.compute.python import OperatorInterface, op_map, register_operator
    from param_amd.compute.python.split_table_batched_embeddings_ops import SplitTableBatchedEmbeddingBagsCodegenOp

    assert isinstance(op_map["SplitTableBatchedEmbeddingBagsCodegen"], SplitTableBatchedEmbeddingBagsCodegenOp)
    with pytest.raises(ValueError, match="Duplicate operator registration name"):
        register_operator("SplitTableBatchedEmbeddingBagsCodegen", SplitTableBatchedEmbeddingBagsCodegenOp())
    with pytest.raises(TypeError):
        OperatorInterface()  # forward is abstract
    op = SplitTableBatchedEmbeddingBagsCodegenOp()
    op.device = "cpu"
    with pytest.raises(ValueError, match="Unknown compute device"):
        op.build(2, 100, 8, 0, False, "fp32", "sgd")
    op.device = "cuda"
    with pytest.raises(ValueError, match="SUM"):
        op.build(2, 100, 8, 1, False, "fp32", "sgd")
    with pytest.raises(ValueError, match="SGD"):
        op.build(2, 100, 8, 0, False, "fp32", "adam")


def test_generate_requests_layout():
    from param_amd.compute.python.split_table_batched_embeddings_ops import generate_batched_request, generate_requests

    i, o, w = generate_requests(4, 3, 10, 0, alpha=0)           # arange % L
    assert i.tolist() == [0, 1, 2] * 4 and o.tolist() == [0, 3, 6, 9, 12] and w is None
    i, o, w = generate_requests(4, 3, 5, 12, alpha=0.5, weighted=True)   # arange % E, continued offsets (no leading 0)
    assert i.tolist() == [x % 5 for x in range(12)] and o.tolist() == [15, 18, 21, 24] and w.shape == (12,)
    torch.manual_seed(0)
    i, _, _ = generate_requests(64, 5, 7, 0, alpha=1.0)
    assert int(i.min()) >= 0 and int(i.max()) < 7
    np.random.seed(0)
    i, _, _ = generate_requests(64, 5, 1000, 0, alpha=1.15)
    assert int(i.max()) < 1000 and (i < 10).float().mean() > 0.3            # zipf % E: mass on small ids
    idx, off, wts = generate_batched_request(3, [10, 20, 30], 4, [2, 3, 1], alpha=1.0, device="cpu")
    assert off.tolist() == [0, 2, 4, 6, 8, 11, 14, 17, 20, 21, 22, 23, 24] and idx.numel() == 24 and wts is None


def test_range_config_iterator_matches_reference(golden_dir):
    """``__range__`` expansion of build configs vs tests/golden/range_configs.json (the reference's RangeConfigIterator,
    generated by tests/golden/gen_range_configs.py); the operator's input iterator (not importable without fbgemm) restated"""
    import json
    import os

    from param_amd.compute.python.config_iter import default_config_iterator, full_range, range_config_iterator, tbe_input_iterator

    def val(a):
        return [val(x) for x in a["value"]] if a.get("type") in ("genericlist", "tuple") else a["value"]

    def values(c):
        return {"args": [val(a) for a in c.get("args", [])], "kwargs": {k: val(a) for k, a in c.get("kwargs", {}).items()}}

    gold = json.load(open(os.path.join(golden_dir, "range_configs.json")))
    assert set(gold) == {"scalars", "two_variants", "genericlist", "no_ranges"}
    for name, case in gold.items():
        mine = [[i, values(c)] for i, c in range_config_iterator(case["variants"])]
        assert mine == case["range_iterator"], name
        for _, c in range_config_iterator(case["variants"]):
            assert not any("__range__" in a for a in c["args"])
    assert [[i, values(c)] for i, c in default_config_iterator(gold["no_ranges"]["variants"])] == gold["no_ranges"]["default_iterator"]
    # the reference's own generator tests (test/test_generator.py:13-31)
    assert list(full_range(-3, 2, 1)) == [-3, -2, -1, 0, 1, 2] and list(full_range(5, 11, 2)) == [5, 7, 9, 11]
    assert list(full_range(3, 11, 3)) == [3, 6, 9]
    inputs = [{"args": [{"type": "int", "value": [512, 1024, 512], "__range__": ["value"]},
                        {"type": "int", "value": [20, 50], "__list__": ["value"]}]},
              {"args": [{"type": "int", "value": 64}, {"type": "int", "value": 7}]}]
    assert list(tbe_input_iterator(inputs)) == [("0_0", [512, 20]), ("0_1", [512, 50]), ("0_2", [1024, 20]),
                                                ("0_3", [1024, 50]), ("0_0", [64, 7])]


def test_compute_python_stream_equals_reference_registry_run(golden_dir):
    """ref_plugin_rows.json["compute_python"]: the (build id, input id, arguments, request shape) stream the REFERENCE's
    BenchmarkConfig produced from a ranged config with this build's operator / iterator / data generator registered in the
    reference's registries (gen_ref_plugin_rows.py).  This build's own runner pieces produce the same stream."""
    from param_amd.compute.python.config_iter import BUILD_ITERATORS, tbe_input_iterator
    from param_amd.compute.python.split_table_batched_embeddings_ops import generate_batched_request

    fx = json.load(open(os.path.join(golden_dir, "ref_plugin_rows.json")))["compute_python"]
    (op_name, info), = fx["config"].items()
    assert op_name == "SplitTableBatchedEmbeddingBagsCodegen"
    mine = []
    for config in info["config"]:
        for build_id, build in BUILD_ITERATORS[info["build_iterator"]](config["build"]):
            b = [a["value"] for a in build["args"]]
            for input_id, (batch, pooling) in tbe_input_iterator(config["input"]):
                idx, off, psw = generate_batched_request(b[0], b[1], batch, pooling, fx["alpha"], b[4], "cpu")
                mine.append({"build_id": build_id, "input_id": input_id, "build_args": b,
                             "input_args": [b[0], b[1], b[2], batch, pooling, b[4], b[5]],
                             "indices_len": int(idx.numel()), "offsets_len": int(off.numel()), "indices_sum": int(idx.sum()),
                             "offsets_last": int(off[-1]), "psw": psw is not None})
    assert len(mine) == 8 and mine == fx["stream"]


def test_quantised_sweep_host_logic_equals_the_reference(golden_dir):
    """``--bitwidth < 32``: flag checks, begin-size fix-up, header / row text and the open downcast, against outputs of the
    reference's own functions (tests/golden/quant_rows.json, made by gen_quant_rows.py running the reference here)"""
    from param_amd.comms.pt import comms, comms_utils
    from param_amd.comms.pt.mi355_backend import MI355XBackend
    from param_amd.comms.pt.pytorch_backend_utils import collectiveArgsHolder

    gold = json.load(open(os.path.join(golden_dir, "quant_rows.json")))
    for (coll, dtype, begin, dim, z), want in gold["checkQuantArgs"]:
        try:
            comms_utils.checkQuantArgs(coll, getattr(torch, dtype), begin, dim, z)
            got = None
        except Exception as e:      # noqa: BLE001
            got = [type(e).__name__, str(e)]
        assert got == want, (coll, dtype, z)
    for (coll, begin, esz, world, bw, dim), want in gold["fixBeginSize"]:
        p = types.SimpleNamespace(collective=coll, beginSize=begin, element_size=esz, bitwidth=bw, quant_a2a_embedding_dim=dim)
        comms_utils.fixBeginSize(p, world)
        assert p.beginSize == want
    assert comms.format_quant_header() + "\n" == gold["preamble_stdout"]
    for r in gold["quant_rows"]:
        q95, d95, p95 = (float(np.percentile(r[k], 95)) for k in ("quant", "dequant", "lat"))
        assert comms.format_quant_row("all_to_allv", "float32", "", r["memSize"], r["numElements"], q95, d95, p95) + "\n" == r["row"]
    # arming / disarming the collectives
    ca = collectiveArgsHolder()
    assert ca.all2all_qcomm is None and ca.allreduce_qcomm == 32 and ca.quant_threshold == 0
    comms_utils.initQuantCommCtx(ca, types.SimpleNamespace(bitwidth=8, quant_a2a_embedding_dim=64))
    assert (ca.all2all_qcomm, ca.allreduce_qcomm, ca.reduce_qcomm, ca.quant_a2a_embedding_dim) == (8, 8, 8, 64)
    comms_utils.clearQuantCommCtx(ca)
    assert ca.all2all_qcomm is None and ca.allreduce_qcomm == 32
    # the reduce-family downcast is the reference's _downcast (fp16 / int8; nothing below 8 bits)
    bf = MI355XBackend(types.SimpleNamespace(master_ip="127.0.0.1"), types.SimpleNamespace(device="cpu"))
    x = torch.tensor(gold["downcast"]["x"])
    ca = collectiveArgsHolder()
    ca.asyncOp = False
    for bits in (16, 8):
        got = bf._downcast_reduce(ca, x, bits, lambda q: None, True)
        assert got.tolist() == gold["downcast"][str(bits)] and got.dtype == torch.float32
    with pytest.raises(NotImplementedError, match=gold["downcast"]["4"].split(".")[0]):
        bf._downcast_reduce(ca, x, 4, lambda q: None, True)
    assert ca.quant_time.getTimeUS() > 0 and ca.dequant_time.getTimeUS() > 0
    # CLI: the reference's three flags with its choices and defaults (comms_utils.py:1788-1806)
    import argparse
    import sys
    old = sys.argv
    sys.argv = ["comms.py", "--bitwidth", "8", "--quant-a2a-embedding-dim", "128"]
    try:
        a = comms.commsCollBench().readArgs(argparse.ArgumentParser())
    finally:
        sys.argv = old
    assert (a.bitwidth, a.quant_a2a_embedding_dim, a.quant_threshold) == (8, 128, 33554432)


def test_graph_launches_flag_is_checked_like_the_reference():
    """``--graph-launches`` (reference comms.py:196-199, 332-334): an integer flag, default 0, refused on a cpu device"""
    import argparse
    import sys

    from param_amd.comms.pt import comms

    def parse(argv):
        bench = comms.commsCollBench()
        old = sys.argv
        sys.argv = ["comms.py"] + argv
        try:
            return bench, bench.readArgs(argparse.ArgumentParser())
        finally:
            sys.argv = old

    bench, a = parse([])
    assert a.graph_launches == 0
    bench, a = parse(["--graph-launches", "7", "--device", "rocm"])
    bench.checkArgs(a)
    assert a.graph_launches == 7 and comms.commsParamsHolder(a, 4, torch.float32, "all_to_all").graph_launches == 7
    bench, a = parse(["--graph-launches", "2", "--device", "cpu", "--backend", "gloo"])
    with pytest.raises(SystemExit):
        bench.checkArgs(a)


def _parse_comms(argv):
    import argparse
    import sys

    from param_amd.comms.pt import comms

    bench = comms.commsCollBench()
    old = sys.argv
    sys.argv = ["comms.py"] + argv
    try:
        return bench, bench.readArgs(argparse.ArgumentParser())
    finally:
        sys.argv = old


def test_comms_cli_defaults_equal_the_reference_parser(golden_dir):
    """tests/golden/comms_cli_defaults.json: defaults of the REFERENCE's comms.py parser for every flag this build keeps
    (gen_comms_surface.py ran the reference's ``readArgs`` here).  Same defaults, so a command line without a flag means the
    same run: non-blocking (``--z 0``), all_reduce, 8 .. 64 bytes, window 100, one communicator group."""
    gold = json.load(open(os.path.join(golden_dir, "comms_cli_defaults.json")))
    bench, a = _parse_comms([])
    for k, v in gold.items():
        mine = getattr(a, k)
        if k == "data_types":
            mine = [d for d in mine.split(",")]
        assert mine == v, (k, mine, v)
    # the reference's spellings of the flags (comms.py:50-206, comms_utils.py:1713-1879) all parse
    bench, a = _parse_comms(["--num_iters", "7", "--num-coll-per-iteration", "2", "--begin-size", "1K", "--end-size", "2K",
                             "--in-split", "1,2", "--out-split", "3,4", "--sizes", "8,16", "--data-type", "int32",
                             "--collectives", "all_to_allv", "--pa", "3", "--blocking", "1", "--log-level", "INFO",
                             "--src-ranks", "0:1", "--dst-ranks", "2,3", "--root", "1", "--multi-comms", "2", "--window", "9"])
    assert (a.n, a.num_coll, a.b, a.e, a.i, a.o, a.ss, a.data_types, a.collective, a.profiler_active_iters, a.z, a.log) == \
           (7, 2, "1K", "2K", [1, 2], [3, 4], [8, 16], "int32", "all_to_allv", 3, 1, "INFO")
    bench.checkArgs(a)
    assert (a.b, a.e, a.collectives, a.dtypes, a.split_elements) == (1024, 2048, ["all_to_allv"], ["int32"], 3)


def test_comms_cli_argument_checks():
    """what the reference refuses (comms.py:208-243, 294-334, 761-806): --i / --o without all_to_allv, an unknown collective,
    bfloat16 on gloo, --graph-launches with --pt2pt; --pt2pt replaces the collective list; --c 1 is dropped for non-blocking
    reductions; pt2pt rank lists are validated per pattern"""
    from param_amd.comms.pt import comms

    bench, a = _parse_comms(["--i", "4,4", "--collective", "all_reduce"])
    with pytest.raises(SystemExit):
        bench.checkArgs(a)
    bench, a = _parse_comms(["--nw-stack", "pytorch-xla-tpu"])
    with pytest.raises(SystemExit):
        bench.checkArgs(a)
    bench, a = _parse_comms(["--collective", "all_to_all,nonsense"])
    with pytest.raises(SystemExit):
        bench.checkArgs(a)
    bench, a = _parse_comms(["--data-types", "bfloat16", "--backend", "gloo", "--device", "cpu"])
    with pytest.raises(SystemExit):
        bench.checkArgs(a)
    bench, a = _parse_comms(["--pt2pt", "one2one", "--graph-launches", "3", "--device", "rocm"])
    with pytest.raises(SystemExit):
        bench.checkArgs(a)
    bench, a = _parse_comms(["--pt2pt", "pairwise", "--collective", "all_reduce", "--tag", "x", "--size-start-profiler", "1K"])
    bench.checkArgs(a)
    assert a.collectives == ["pt2pt"] and bench.tag == "-x" and a.size_start_profiler == 1024
    bench, a = _parse_comms(["--c", "1", "--z", "0", "--collective", "all_to_allv,reduce_scatter"])
    bench.checkArgs(a)
    assert a.c == 0
    bench, a = _parse_comms(["--c", "1", "--z", "0", "--collective", "all_to_allv,all_gather"])
    bench.checkArgs(a)
    assert a.c == 1
    # pt2pt rank lists (checkPt2PtRanks): defaults 0 -> 1; one2one takes one pair; pairwise wants equal, disjoint lists
    for pattern, src, dst, ok in (("one2one", None, None, True), ("one2one", [0, 1], [2], False), ("pairwise", [0, 1], [2, 3], True),
                                  ("pairwise", [0, 1], [2], False), ("pairwise", [0, 1], [1, 2], False), ("one2one", [0], [9], False)):
        bench = comms.commsCollBench()
        bench.report, bench.comm_size = False, 4
        ca = bench.collectiveArgs
        ca.collective, ca.pt2pt, ca.src_ranks, ca.dst_ranks = "pt2pt", pattern, src, dst
        if ok:
            bench.checkPt2PtRanks()
            assert (ca.src_ranks, ca.dst_ranks) == (src or [0], dst or [1])
        else:
            with pytest.raises(SystemExit):
                bench.checkPt2PtRanks()
    # incast / multicast: every rank but the root unless listed (checkCollectiveRanks)
    bench = comms.commsCollBench()
    bench.report, bench.comm_size = False, 4
    ca = bench.collectiveArgs
    ca.collective, ca.srcOrDst, ca.src_ranks = "incast", 2, None
    bench.checkCollectiveRanks()
    assert ca.src_ranks == [0, 1, 3]
    ca.collective, ca.srcOrDst, ca.dst_ranks = "multicast", 0, [0, 1, 2]
    bench.checkCollectiveRanks()
    assert ca.dst_ranks == [1, 2]
    # the pt2pt row / header text: field widths of the reference's format strings (comms.py:961, 1244)
    hdr = comms.format_pt2pt_header()
    row = comms.format_pt2pt_row("recv", "float32", "-t", 1024, (1.0, 2.0, 3.0), (4.0, 5.0, 6.0), 0.5, 0.75, 1.0, 1.5)
    assert hdr.startswith("\n\tCOMMS-RES" + " " * 32 + "size (B)") and hdr.endswith("totalBiBW(GB/s)")
    assert row.split() == ["COMMS-RES-recv-float32-t", "1024", "1.0", "2.0", "3.0", "4.0", "5.0", "6.0", "0.500", "0.750", "1.000", "1.500"]


def test_run_benchmark_resume_stop_window_equals_reference(golden_dir, capsys):
    """``run_benchmark.py -r / -s``: skip / run / stop id by id equals the REFERENCE's BuildExecutor.get_transition_state for the
    same id stream (tests/golden/run_window.json, gen_run_window.py); ``--version`` and a missing ``-c`` end the run"""
    from param_amd.compute.python import run_benchmark as R

    gold = json.load(open(os.path.join(golden_dir, "run_window.json")))
    for case in gold["cases"]:
        w = R.RunWindow(case["resume"], case["stop"])
        assert [w.step(i) for i in gold["ids"]] == case["states"], case
    assert R.main(["--version"]) == [] and "PARAM train compute version" in capsys.readouterr().out
    assert R.main([]) == []


def test_perf_logger_records_match_reference_fields(golden_dir, tmp_path):
    """logger_utils.py: the four record classes have the REFERENCE's field names, order and defaults
    (tests/golden/perf_metric_fields.json, from the reference's dataclasses), so a logger written against PARAM reads them;
    registry semantics: name -> instance, unknown names skipped, the built-in ``jsonl`` logger appends one object per record"""
    import dataclasses

    from param_amd.comms.pt import logger_utils as L

    gold = json.load(open(os.path.join(golden_dir, "perf_metric_fields.json")))
    assert {m.name: m.value for m in L.benchType} == gold["benchType"]
    for cls in (L.commsPerfMetrics, L.commsCollPerfMetrics, L.commsQuantCollPerfMetrics, L.commsPt2PtPerfMetrics):
        inst = cls()
        mine = {f.name: (getattr(inst, f.name).name if f.name == "BenchCommsType" and getattr(inst, f.name) is not None
                         else getattr(inst, f.name)) for f in dataclasses.fields(cls)}
        assert list(mine.items()) == list(gold[cls.__name__].items()), cls.__name__
    seen = []

    class Mine(L.commsPerfLogger):
        def logPerf(self, benchmarkName, metrics, backendFuncs, **kwargs):
            seen.append((benchmarkName, metrics.commsOp, kwargs))

    L.register_perf_logger("mine", Mine("mine"))
    try:
        m = L.commsCollPerfMetrics(commsOp="all_to_allv", p50_latency_us=3.0)
        L.dispatch(["mine", "absent"], "comms", m, None, extra=1)
        assert seen == [("comms", "all_to_allv", {"extra": 1})] and L.customized_perf_loggers["mine"].name == "mine"
        L.JsonLinesPerfLogger(path=str(tmp_path / "x.jsonl")).logPerf("replay", m, None, note="n")
        rec = json.loads(open(tmp_path / "x.jsonl").read())
        assert rec["commsOp"] == "all_to_allv" and rec["BenchCommsType"] == "Collective" and rec["benchmark"] == "replay" and rec["note"] == "n"
        L.dispatch(None, "comms", m, None)
    finally:
        L.customized_perf_loggers.pop("mine")


def test_comms_compute_bench_report_equals_reference_text(golden_dir, monkeypatch, capsys):
    """commsComputeBench.py prints the reference's preamble and COMMS-RES rows (device time of the collective as the latency
    columns, host time per iteration and compute device time as the two extra columns in comms-compute mode):
    tests/golden/commscompute_rows.json holds what the REFERENCE's printPreamble / reportBenchTimeColl print for fixed
    latencies.  Then the driver end to end on one gloo rank with the lookup kernel stubbed out (the product kernel has no CPU
    path): header, row and the build's own COMMS-COMPUTE-RES line per size."""
    from param_amd.comms.pt import comms_utils, commsComputeBench as C
    from param_amd.comms.pt.mi355_backend import MI355XBackend
    from param_amd.comms.pt.pytorch_backend_utils import backendFunctions

    gold = json.load(open(os.path.join(golden_dir, "commscompute_rows.json")))
    for key, text in gold["headers"].items():
        mode, bw = key.split("/")
        assert C.format_cc_header(mode, int(bw)) + "\n" == text, key
    for g in gold["rows"]:
        ca = types.SimpleNamespace(world_size=g["world_size"])
        rep = C.cc_report("all_to_allv", g["world_size"], g["results"]["memSize"], g["lat"], g["comm"], g["comp"],
                          lambda c, bw, n: backendFunctions.getBusBW(None, c, bw, ca))
        assert C.format_cc_row("all_to_allv", "float32", g["tag"], g["results"]["memSize"], g["results"]["numElements"], rep,
                               g["mode"]) + "\n" == g["row"], g["mode"]
    # end to end on the host: stub kernel, one rank
    monkeypatch.setattr(comms_utils, "init_emb_lookup", lambda ca, args, bf: None)
    monkeypatch.setattr(MI355XBackend, "emb_lookup", lambda self, ca: None)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_SIZE"):
        monkeypatch.delenv(k, raising=False)
    from tests.dist_workers import _torch_host_row_codec, free_port

    # (the --bitwidth 8 invocation below quantises HOST tensors: the product backend has no host codec, the test injects torch's)
    monkeypatch.setattr(MI355XBackend, "host_row_codec", staticmethod(_torch_host_row_codec))
    res = C.main(["--master-ip", "127.0.0.1", "--master-port", str(free_port()), "--b", "1K", "--e", "4K", "--f", "4", "--n", "3",
                  "--w", "1", "--collective", "all_to_allv", "--device", "cpu", "--backend", "gloo", "--num-compute", "2",
                  "--ntables", "2", "--batch-size", "8", "--bag-size", "3", "--tag", "x"])
    out = capsys.readouterr().out
    assert [r["size"] for r in res] == [1024, 4096] and all(r["report"]["p50"] > 0 and r["lookups_per_iter"] == 2 * 8 * 3 * 2 for r in res)
    assert C.format_cc_header("comms-compute", 32) in out and "mode: comms-compute, num_coll: 1, kernel: emb_lookup" in out
    rows = [ln for ln in out.splitlines() if ln.startswith("\tCOMMS-RES-all_to_allv-float32-x")]
    assert len(rows) == 2 and [int(ln.split()[1]) for ln in rows] == [1024, 4096] and len(rows[0].split()) == 12
    assert out.count("COMMS-COMPUTE-RES-all_to_allv-emb_lookup") == 2
    # the two other invocations of the GPU test: compute only (no collective: zero bytes, the short header), and --bitwidth 8
    res = C.main(["--master-ip", "127.0.0.1", "--master-port", str(free_port()), "--b", "1K", "--e", "1K", "--n", "2", "--w", "1",
                  "--collective", "all_to_allv", "--device", "cpu", "--backend", "gloo", "--mode", "compute", "--direction", "backward",
                  "--num-compute", "2", "--ntables", "4", "--num-emb-tables-batched", "2", "--batch-size", "8"])
    out = capsys.readouterr().out
    assert res[0]["memSize"] == 0 and res[0]["report"]["algBW"] == 0 and C.format_cc_header("compute", 32) in out
    assert len([ln for ln in out.splitlines() if ln.startswith("\tCOMMS-RES-all_to_allv-float32")][0].split()) == 10
    res = C.main(["--master-ip", "127.0.0.1", "--master-port", str(free_port()), "--b", "4K", "--e", "4K", "--n", "3", "--w", "1",
                  "--collective", "all_to_allv", "--device", "cpu", "--backend", "rccl_xgmi", "--ntables", "4", "--batch-size", "8",
                  "--bitwidth", "8", "--quant-a2a-embedding-dim", "128", "--z", "1"])
    out = capsys.readouterr().out
    assert res[0]["bitwidth"] == 8 and res[0]["quant_us"] > 0 and res[0]["dequant_us"] > 0 and C.format_cc_header("comms-compute", 8) in out


def test_run_benchmark_main_with_stub_operator(tmp_path, monkeypatch, golden_dir):
    """run_benchmark.main end to end on the host with the operator and the executor stubbed out (the product operator has no CPU
    path): id stream of a ranged config, ``-r / -s`` window applied to it, result file = header line + one JSON line per run,
    ``--append`` keeps the earlier lines, the per-metric log lines"""
    from param_amd.compute.python import op_map, run_benchmark as R

    calls = []

    class FakeOp:
        device = None

        def cleanup(self):
            pass

        def build(self, *a, **k):
            calls.append(a[:3])

    class FakeExec:
        def __init__(self, *a, **k):
            pass

        def run(self, data):
            return {"forward": {"gpu.time": [0.5, 0.25], "gpu.memory": [1.0, 1.0]}, "backward": {"gpu.time": [], "gpu.memory": []}}

    monkeypatch.setitem(op_map, "FakeTBE", FakeOp())
    monkeypatch.setattr(R, "OpExecutor", FakeExec)
    fx = json.load(open(os.path.join(golden_dir, "ref_plugin_rows.json")))["compute_python"]
    cfg = {"FakeTBE": dict(list(fx["config"].values())[0])}
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    want = [f"0|{r['build_id']}|{r['input_id']}" for r in fx["stream"]]       # the reference's (build id, input id) stream
    res = R.main(["-c", str(path), "-d", "cpu", "-o", str(tmp_path / "out")])
    assert [r["id"] for r in res] == want and len(calls) == len(want)
    lines = (tmp_path / "out.json").read_text().splitlines()
    assert len(lines) == len(want) + 1 and "run_options" in json.loads(lines[0]) and "sys_info" in json.loads(lines[0])
    assert [json.loads(ln)["id"] for ln in lines[1:]] == want and json.loads(lines[1])["op_name"] == "FakeTBE"
    # resume at the third run, stop in front of the sixth; --append leaves the earlier file content in place
    res = R.main(["-c", str(path), "-d", "cpu", "-o", str(tmp_path / "out"), "-a", "-r", f"FakeTBE|{want[2]}", "-s", f"FakeTBE|{want[5]}"])
    assert [r["id"] for r in res] == want[2:5]
    lines2 = (tmp_path / "out.json").read_text().splitlines()
    assert lines2[:len(lines)] == lines and len(lines2) == len(lines) + 1 + 3


def test_comms_cli_refuses_what_one_rank_cannot_run(monkeypatch, capsys):
    """a single rank (the GPU visits' world): --pt2pt needs a peer, --multi-comms must divide the ranks, --root must exist --
    a configuration error ends the run through gracefulExit with the process group shut down, not with a hang or a traceback"""
    import torch.distributed as dist

    from param_amd.comms.pt import comms
    from tests.dist_workers import free_port

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_SIZE"):
        monkeypatch.delenv(k, raising=False)
    for extra in (["--pt2pt", "one2one"], ["--multi-comms", "2"], ["--root", "1", "--collective", "broadcast"],
                  ["--collective", "all_to_allv", "--i", "4,4"]):
        with pytest.raises(SystemExit):
            comms.main(["--master-ip", "127.0.0.1", "--master-port", str(free_port()), "--device", "cpu", "--backend", "gloo"] + extra)
        assert not dist.is_initialized()
    capsys.readouterr()


def test_batched_module_reports_embedding_specs():
    """``embedding_specs``: (rows, dim, location, compute device) per table, the attribute of fbgemm's TBE module the reference's
    operator test reads back (test_split_table_batched_embeddings_ops.py:29-60); scalar rows / dims are broadcast as there"""
    import param_amd

    m = param_amd.BatchedEmbeddingBagMI355([1000, 2000], [64, 128], device="cpu", init=None)
    assert [(s[0], s[1]) for s in m.embedding_specs] == [(1000, 64), (2000, 128)] and m.embedding_specs[0][2:] == ("host", "cpu")


# ----------------------------------------------------------------------------- bench.py launch contract (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_without_launcher_launches_itself(tmp_path):
    """`python bench.py --gpus 4 ...` with WORLD_SIZE unset must not measure one GPU and call it four (round 4 did): it re-executes
    itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 ...` with its own
    arguments.  The launcher is replaced by a stub that records the command line."""
    import subprocess
    import sys

    stub = tmp_path / "stub.sh"
    rec = tmp_path / "cmd.txt"
    stub.write_text(f"#!/bin/bash\nprintf '%s\\n' \"$@\" > {rec}\nexit 7\n")
    stub.chmod(0o755)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PARAM_AMD_BENCH_LAUNCHER"] = str(stub)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--tables", "26"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, r.stderr[-500:]                      # the launcher's exit code is the bench's
    cmd = rec.read_text().split("\n")
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:i + 9] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--tables", "26"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """a launcher that started 2 ranks for --gpus 4 (or 1 rank for --gpus 2) is an error before anything is measured"""
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
