"""The batched multi-table EmbeddingBag operator plug-in -- own restatement of reference
``train/compute/python/workloads/pytorch/split_table_batched_embeddings_ops.py`` (operator ``:239-324``,
request generator ``:93-135``, data generator ``:138-229``) served by the MI355X HIP kernels instead of
``fbgemm_gpu.SplitTableBatchedEmbeddingBagsCodegen``.

Kept: the operator name ``"SplitTableBatchedEmbeddingBagsCodegen"``, ``build(num_tables, rows, dims, pooling,
weighted, weights_precision, optimizer, lr, eps, weight_decay, weight_decay_mode)``, ``forward(indices, offsets,
per_sample_weights)`` storing and returning ``fwd_out``, ``create_grad()`` = ``ones_like(fwd_out)``,
``backward(grad=None)``, ``cleanup()``; the request layout (indices concatenated table-major, offsets running
on across tables, ``T*B+1`` entries) and the generator's alpha switch (0: ``arange % L``, <= 0.5: ``arange % E``,
<= 1: uniform, > 1: ``np.random.zipf(alpha) % E``).

Stated differences: pooling must be SUM (PoolingMode 0, the only mode on the reference hot path); the
optimizer fused into ``backward`` is plain SGD (``"sgd"``/``"exact_sgd"``) or exact row-wise Adagrad
(``"exact_row_wise_adagrad"``, the reference's choice at comms_utils.py:2014) -- other optimizer names raise;
weight decay (L2 / decoupled) and stochastic rounding of 16-bit tables apply to the row-wise Adagrad update (plain SGD
updates round to nearest); ``device`` must be a ROCm device.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .operator import OperatorInterface, register_operator

_PRECISION = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16,
              "float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}


def generate_requests(B: int, L: int, E: int, offset_start: int, alpha: float = 1.0, weighted: bool = False):
    """One table's (indices, offsets, per_sample_weights) -- reference ``:93-135``."""
    n = B * L
    if alpha == 0:
        indices = torch.arange(0, n).long() % L
    elif alpha <= 0.5:
        indices = torch.arange(0, n).long() % E
    elif alpha <= 1.0:
        indices = torch.randint(low=0, high=E, size=(n,), dtype=torch.int64)
    else:
        indices = torch.as_tensor(np.random.zipf(a=alpha, size=n)).long() % E
    ends = offset_start + (torch.arange(1, B + 1, dtype=torch.int64) * L)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), ends]) if offset_start == 0 else ends
    weights = torch.randn(n, dtype=torch.float32) if weighted else None
    return indices, offsets, weights


def generate_batched_request(num_tables: int, rows, batch_size: int, pooling_factors, alpha: float = 1.0,
                             weighted: bool = False, device="cuda"):
    """All tables concatenated the way the reference's data generator does (``:191-208``)."""
    rows = rows if isinstance(rows, (list, tuple)) else [rows] * num_tables
    pooling_factors = pooling_factors if isinstance(pooling_factors, (list, tuple)) else [pooling_factors] * num_tables
    idx, off, wts, start = [], [], [], 0
    for t in range(num_tables):
        i, o, w = generate_requests(batch_size, pooling_factors[t], rows[t], start, float(alpha), weighted)
        idx.append(i)
        off.append(o)
        start = int(o[-1])
        if weighted:
            wts.append(w)
    dev = torch.device(device)
    return (torch.cat(idx).to(dev), torch.cat(off).to(dev), torch.cat(wts).to(dev) if weighted else None)


def _wd_mode(mode):
    """fbgemm ``WeightDecayMode`` given as enum member, int (0 NONE, 1 L2, 2 DECOUPLE) or name"""
    if mode is None:
        return None
    if hasattr(mode, "value"):
        mode = mode.value
    if isinstance(mode, str):
        mode = mode.lower().rsplit(".", 1)[-1]
    return mode


class SplitTableBatchedEmbeddingBagsCodegenOp(OperatorInterface):
    def __init__(self):
        super().__init__()
        self.op = None
        self.fwd_out: Optional[torch.Tensor] = None
        self.grad_in: Optional[torch.Tensor] = None
        self._request = None

    def build(self, num_tables: int, rows, dims, pooling: int, weighted: bool, weights_precision: str,
              optimizer: str, lr: float = 0.01, eps: float = 1.0e-8, weight_decay: float = 0.0,
              weight_decay_mode=None):
        from ... import BatchedEmbeddingBagMI355

        rows_list = rows if isinstance(rows, list) else [rows] * num_tables
        dims_list = dims if isinstance(dims, list) else [dims] * num_tables
        if int(pooling) != 0:
            raise ValueError("only PoolingMode.SUM (0) is implemented on the MI355X path")
        opt = str(optimizer).lower()
        if opt in ("sgd", "exact_sgd"):
            opt = "sgd"
        elif opt in ("exact_row_wise_adagrad", "exact_rowwise_adagrad", "rowwise_adagrad", "row_wise_adagrad"):
            opt = "rowwise_adagrad"   # fbgemm OptimType.EXACT_ROWWISE_ADAGRAD ("exact_row_wise_adagrad")
        else:
            raise ValueError(f"optimizer {optimizer!r}: the MI355X backward fuses plain SGD and exact row-wise Adagrad only")
        if not str(self.device).startswith(("cuda", "rocm")):
            raise ValueError(f"Unknown compute device {self.device} (the MI355X operator needs a ROCm device)")
        dev = "cuda" + str(self.device)[4:] if str(self.device).startswith("rocm") else str(self.device)
        self.weighted = weighted
        self.op = BatchedEmbeddingBagMI355(rows_list, dims_list, dtype=_PRECISION[str(weights_precision).lower()],
                                           device=dev, init="uniform_dlrm", learning_rate=lr, fused_update=True,
                                           optimizer=opt, eps=eps, weight_decay=weight_decay,
                                           weight_decay_mode=_wd_mode(weight_decay_mode),
                                           stochastic_rounding=True)   # the reference's fixed choice (:291)

    def cleanup(self):
        self.op = None
        self.grad_in = None
        self.fwd_out = None
        self._request = None

    def forward(self, *args, **kwargs):
        indices, offsets = args[0], args[1]
        psw = args[2] if len(args) > 2 else None
        self._request = (indices, offsets, psw)
        self.fwd_out = self.op.lookup(indices, offsets, psw)
        return self.fwd_out

    def create_grad(self):
        self.grad_in = torch.ones_like(self.fwd_out)

    def backward(self, grad=None):
        if grad is None:
            if self.grad_in is None:
                self.create_grad()
            grad = self.grad_in
        indices, offsets, psw = self._request
        self.op.optimizer_step_(grad, indices, offsets, per_sample_weights=psw)


register_operator("SplitTableBatchedEmbeddingBagsCodegen", SplitTableBatchedEmbeddingBagsCodegenOp())
