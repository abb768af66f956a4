"""Row-wise quantisation for the quantised all-to-all (csrc/rowquant.hip): oracle pinned to torch's operators, the HIP
kernels bit-exact against the oracle through the C ABI."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import rowquant as orq
from param_amd import _lib

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "rowquant.npz"))
DIMS = sorted({int(k.split("_")[1]) for k in GOLD.files if k.startswith("x_")})
BITS = (16, 8, 4, 2)


@pytest.mark.parametrize("bits", BITS)
def test_oracle_matches_the_committed_torch_outputs(bits):
    for dim in DIMS:
        x = GOLD[f"x_{dim}"]
        q = orq.quantize_rows(x, bits)
        assert q.shape == (x.shape[0], orq.row_bytes(dim, bits))
        assert np.array_equal(q, GOLD[f"q{bits}_{dim}"]), (bits, dim)
        d = orq.dequantize_rows(GOLD[f"q{bits}_{dim}"], dim, bits)
        assert np.array_equal(d.view(np.uint32), GOLD[f"d{bits}_{dim}"].view(np.uint32)), (bits, dim)


def test_oracle_matches_the_installed_torch_operators_live():
    ops = {8: (torch.ops.quantized.embedding_bag_byte_prepack, torch.ops.quantized.embedding_bag_byte_unpack),
           4: (torch.ops.quantized.embedding_bag_4bit_prepack, torch.ops.quantized.embedding_bag_4bit_unpack),
           2: (torch.ops.quantized.embedding_bag_2bit_prepack, torch.ops.quantized.embedding_bag_2bit_unpack)}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4096, 128, generator=g) * (torch.rand(4096, 1, generator=g) * 20)
    for bits, (pack, unpack) in ops.items():
        q = pack(x)
        assert np.array_equal(orq.quantize_rows(x.numpy(), bits), q.numpy()), bits
        assert np.array_equal(orq.dequantize_rows(q.numpy(), 128, bits), unpack(q).numpy()), bits
    assert np.array_equal(orq.quantize_rows(x.numpy(), 16).view(np.float16), x.to(torch.float16).numpy())


def test_c_abi_argument_checks_without_a_gpu():
    L = _lib.load()
    assert L.pm_rows_quantized_bytes(10, 128, 8) == 10 * 136
    assert L.pm_rows_quantized_bytes(10, 128, 16) == 10 * 256
    assert L.pm_rows_quantized_bytes(10, 128, 4) == 10 * 68
    assert L.pm_rows_quantized_bytes(10, 128, 2) == 10 * 36
    assert L.pm_rows_quantized_bytes(0, 32, 2) == 0
    assert L.pm_rows_quantized_bytes(1, 128, 32) == _lib.PM_ERR_INVALID and b"bitwidth" in L.pm_last_error()
    assert L.pm_rows_quantized_bytes(1, 100, 8) == _lib.PM_ERR_UNSUPPORTED and b"multiple of 8" in L.pm_last_error()
    assert L.pm_rows_quantized_bytes(1, 1024, 8) == _lib.PM_ERR_UNSUPPORTED
    assert L.pm_rows_quantized_bytes(-1, 128, 8) == _lib.PM_ERR_INVALID
    assert L.pm_rows_quantize(None, 4, 128, 8, None, None) == _lib.PM_ERR_INVALID and b"NULL" in L.pm_last_error()
    assert L.pm_rows_quantize(ctypes.c_void_p(8), 4, 128, 8, ctypes.c_void_p(16), None) == _lib.PM_ERR_INVALID
    assert b"aligned" in L.pm_last_error()
    assert L.pm_rows_dequantize(ctypes.c_void_p(16), 4, 128, 3, ctypes.c_void_p(16), None) == _lib.PM_ERR_INVALID
    assert L.pm_rows_quantize(None, 0, 128, 8, None, None) == _lib.PM_OK        # nothing to do
    from param_amd import quant
    with pytest.raises(RuntimeError, match="no CPU path"):
        quant.quantize_rows(torch.zeros(4, 128), 128, 8)
    assert quant.row_bytes(256, 8) == 264


@pytest.mark.gpu
@pytest.mark.parametrize("bits", BITS)
def test_gpu_kernels_bit_exact_on_the_golden_inputs(bits):
    from param_amd import quant
    for dim in DIMS:
        x = torch.from_numpy(GOLD[f"x_{dim}"]).cuda()
        q = quant.quantize_rows(x, dim, bits)
        assert np.array_equal(q.cpu().numpy(), GOLD[f"q{bits}_{dim}"]), (bits, dim)
        d = quant.dequantize_rows(torch.from_numpy(GOLD[f"q{bits}_{dim}"]).cuda(), dim, bits)
        assert np.array_equal(d.cpu().numpy().view(np.uint32), GOLD[f"d{bits}_{dim}"].view(np.uint32)), (bits, dim)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("dim", [32, 64, 128, 256, 8, 24, 96, 200, 512])
def test_gpu_kernels_bit_exact_against_the_oracle_seeded(bits, dim):
    from param_amd import quant
    g = torch.Generator().manual_seed(1000 * bits + dim)
    n = 3001                                                    # not a multiple of the rows per block
    x = torch.randn(n, dim, generator=g) * (torch.rand(n, 1, generator=g) * 30 + 1e-3)
    x[::97] = x[::97, :1]                                       # constant rows
    q = quant.quantize_rows(x.cuda(), dim, bits)
    want = orq.quantize_rows(x.numpy(), bits)
    assert np.array_equal(q.cpu().numpy(), want)
    d = quant.dequantize_rows(q, dim, bits)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), orq.dequantize_rows(want, dim, bits).view(np.uint32))
    # quantisation error bound of the format: half a step of the row's range (plus the fp16 rounding of scale / bias)
    if bits != 16:
        step = (x.max(1).values - x.min(1).values) / ((1 << bits) - 1)
        tol = step * (0.502 if bits == 8 else 0.51) + x.abs().max(1).values * (1e-6 if bits == 8 else 2e-3) + 1e-6
        assert bool(((d.cpu() - x).abs().max(1).values <= tol).all())


@pytest.mark.gpu
def test_gpu_full_size_exchange_payload_round_trip_properties():
    """one rank's forward payload of the 8-GPU DLRM exchange (8 x 8192 bags x 4 tables x 128): quantising the restored rows
    again reproduces the bias and the codes up to one step (idempotence), and an affine map of a row leaves its codes
    in place up to a rounding -- size-independent properties, checked on the GPU alone"""
    from param_amd import quant
    n, dim = 8 * 8192 * 4, 128
    x = torch.randn(n, dim, device="cuda") * 3
    for bits in (8, 16):
        q = quant.quantize_rows(x, dim, bits)
        d = quant.dequantize_rows(q, dim, bits)
        q2 = quant.quantize_rows(d, dim, bits)
        if bits == 16:
            assert torch.equal(q, q2) and torch.equal(d, x.to(torch.float16).float())
        else:
            # the restored row spans exactly [bias, bias + 255 * scale]: same bias, and codes move by at most one step
            assert torch.equal(q[:, dim + 4:], q2[:, dim + 4:])
            assert int((q[:, :dim].int() - q2[:, :dim].int()).abs().max()) <= 1
            codes2 = quant.quantize_rows(x * 4.0 + 1.0, dim, bits)[:, :dim]    # affine per row: codes move by a rounding at most
            assert int((codes2.int() - q[:, :dim].int()).abs().max()) <= 1
            assert float((codes2 != q[:, :dim]).float().mean()) < 0.02
    with pytest.raises(TypeError):
        quant.quantize_rows(x.half(), dim, 8)
    with pytest.raises(ValueError):
        quant.quantize_rows(x.reshape(-1)[:1000], 128, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("layout", ["bd", "tbd"])
def test_gpu_forward_with_quantised_output_equals_two_pass(bits, layout):
    """pm_embbag_fwd_quantized: the lookup kernel's output burst writes the quantised rows itself -- bytes identical to
    quantising the fp32 forward's output (and through it to the oracle), for fixed pooling (staged kernel), weighted and
    unweighted, fp32 and bf16 tables, both output layouts and a batch slice; ragged requests fall back to two passes"""
    from param_amd import BatchedEmbeddingBagMI355, quant
    from param_amd.compute.python.split_table_batched_embeddings_ops import generate_batched_request
    torch.manual_seed(bits)
    for dtype, D, T, B, L, weighted in ((torch.float32, 128, 5, 300, 20, False), (torch.bfloat16, 64, 8, 64, 7, True),
                                        (torch.float32, 256, 2, 1000, 3, False), (torch.float16, 32, 3, 77, 30, False),
                                        (torch.float32, 8, 2, 500, 20, False), (torch.float32, 512, 2, 96, 4, True),
                                        (torch.bfloat16, 128, 1, 4096, 20, False)):
        m = BatchedEmbeddingBagMI355([5000 + 10 * t for t in range(T)], D, dtype=dtype, device="cuda", layout=layout)
        m.reset_parameters(seed=3)
        idx, off, w = generate_batched_request(T, m.rows, B, [L] * T, alpha=1.05, weighted=weighted, device="cuda")
        ref = m.lookup(idx, off, w)
        want = orq.quantize_rows(ref.cpu().numpy().reshape(-1, D), bits)
        got = m.lookup_quantized(idx, off, bits, w)
        assert got.shape[-1] == orq.row_bytes(D, bits) and got.numel() == want.size
        assert np.array_equal(got.cpu().numpy().reshape(want.shape), want), (dtype, D, layout)
        assert torch.equal(quant.quantize_rows(ref, D, bits).view(-1), got.view(-1))
        if layout == "bd":                                     # batch slice: only those bags' rows are written
            buf = torch.full_like(got, 0xAB)
            m.lookup_quantized(idx, off, bits, w, out=buf, bag_begin=10, bag_count=30)
            assert torch.equal(buf[10:40], got[10:40]) and bool((buf[:10] == 0xAB).all()) and bool((buf[40:] == 0xAB).all())
    # ragged request: not staged -> PM_ERR_UNSUPPORTED from the C entry, the module runs forward + quantise
    m = BatchedEmbeddingBagMI355([3000] * 4, 128, device="cuda", layout=layout)
    m.reset_parameters(seed=1)
    idx, off, _ = generate_batched_request(4, m.rows, 200, [1, 40, 7, 13], alpha=0.0, device="cuda")
    op = m._tables().request(idx, off, 200, None, 0, None)
    L_ = _lib.load()
    direct = torch.empty(200 * 4 * orq.row_bytes(128, bits), dtype=torch.uint8, device="cuda")
    rc = L_.pm_embbag_fwd_quantized(ctypes.byref(op), ctypes.c_void_p(direct.data_ptr()), bits,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    got = m.lookup_quantized(idx, off, bits)
    want = orq.quantize_rows(m.lookup(idx, off).cpu().numpy().reshape(-1, 128), bits)
    assert np.array_equal(got.cpu().numpy().reshape(want.shape), want)
    assert rc in (_lib.PM_OK, _lib.PM_ERR_UNSUPPORTED)
    if rc == _lib.PM_OK:
        assert np.array_equal(direct.cpu().numpy().reshape(want.shape), want)
    # the layout contract, checked on the device
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert L_.pm_embbag_check_ex(ctypes.byref(op), 1, ctypes.c_void_p(err.data_ptr()), None) == _lib.PM_OK and int(err.item()) == 0
    mixed = BatchedEmbeddingBagMI355([100, 100], [64, 128], device="cuda")
    idx2, off2, _ = generate_batched_request(2, mixed.rows, 16, [2, 2], alpha=0.0, device="cuda")
    op2 = mixed._tables().request(idx2, off2, 16, None, 0, None)
    assert L_.pm_embbag_check_ex(ctypes.byref(op2), 1, ctypes.c_void_p(err.data_ptr()), None) == _lib.PM_OK and int(err.item()) > 0
    assert L_.pm_embbag_check_ex(ctypes.byref(op2), 0, ctypes.c_void_p(err.data_ptr()), None) == _lib.PM_OK and int(err.item()) == 0
    with pytest.raises(ValueError, match="common embedding dim"):
        mixed.lookup_quantized(idx2, off2, bits)
