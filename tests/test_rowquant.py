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
