#!/usr/bin/env python3
"""Accuracy of the sorted backward's chunk-partial order on a HOT row: one row looked up thousands of times with the same
gradient value.  The sequential fp32 sum (the oracle's order) loses low bits as the running sum grows; the GPU adds
per-chunk partial sums in chunk order, which is closer to the fp64 truth."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import param_amd
from oracle.embbag_oracle import COracle

dev = torch.device("cuda:0")
n, D = 5147, 128
m = param_amd.BatchedEmbeddingBagMI355([64], D, device=dev, init="normal", seed=1, fused_update=False)
idx = torch.full((n,), 3, dtype=torch.int64, device=dev)          # every bag looks up row 3 once
off = torch.arange(n + 1, dtype=torch.int64, device=dev)
g = torch.full((n, D), 0.1, device=dev) + torch.arange(D, device=dev) * 1e-3
(dense,) = m.dense_grad(g, idx, off, batch=n)
gpu = dense[3].double().cpu().numpy()
truth = g.double().cpu().numpy().sum(0)
ref = COracle().bwd_f32(np.zeros((64, D), np.float32), idx.cpu().numpy(), np.arange(n), g.cpu().numpy())[3].astype(np.float64)
print(json.dumps({"adds": n, "max|gpu - fp64|": float(np.abs(gpu - truth).max()), "max|sequential fp32 - fp64|": float(np.abs(ref - truth).max()),
                  "sum": float(truth.max())}))
