#!/usr/bin/env python3
"""Round 6: kernel time of the single-table forward at the reference driver's small batches (14 M x 128 fp32, nnz 30, batch 512 .. 16384:
train/compute/pt/dataset.py:56-82) for row-load batches of 2 / 4 / 8 (pm_set_tuning(unroll)).  Run under
`rocprofv3 --kernel-trace`: the kernel's duration per (UNROLL template argument, grid) is read from the trace by
tools/r6_small_batch_parse.py -- at these sizes the host's issue rate hides the kernel from a wall clock."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd  # noqa: E402
from param_amd.indices import fixed_offsets, init_indices  # noqa: E402

dev = torch.device("cuda:0")
features, D, nnz = 14_000_000, 128, 30
emb = param_amd.EmbeddingBagMI355(features, D, mode="sum", device=dev)
emb.weight.requires_grad_(False)
for batch in (512, 1024, 2048, 4096, 8192, 16384):
    idx = init_indices(0.0, features, batch, nnz).to(dev)
    off = fixed_offsets(batch, nnz, device=dev)
    for unroll in (0, 2, 4, 8):          # 0 = the library's own choice
        param_amd.set_tuning(unroll=unroll)
        for _ in range(60):
            emb(idx, off)
        torch.cuda.synchronize()
param_amd.set_tuning()
