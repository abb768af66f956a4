#!/usr/bin/env python3
"""Why was torch.nn.EmbeddingBag on the GPU box's host 9x faster with autograd ON than under no_grad (round 2's cpu_baseline)?
One table 10 M x 128 fp32, batch 8192, pooling 20, 8 index sets in turn (the bench's CPU sample), reference protocol
(perf_counter around `steps` calls after warm-ups).  Modes are run in both orders, with the same step count, and the ops each
mode dispatches are listed from a 3-step profile."""
import json
import os
import statistics
import sys
import time

import torch

R, D, B, L, K = 10_000_000, 128, 8192, 20, 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
W = torch.empty(R, D).normal_()
sets = [torch.randint(0, R, (B * L,)) for _ in range(K)]
off = torch.arange(B, dtype=torch.int64) * L
emb = torch.nn.EmbeddingBag(R, D, mode="sum", _weight=W)
k = [0]


def step():
    k[0] += 1
    return emb(sets[k[0] % K], off)


def measure(n):
    for _ in range(3):
        step()
    reps = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        reps.append((time.perf_counter() - t0) / n)
    med = statistics.median(reps)
    return {"ms_per_step": med * 1e3, "G_lookups_s": B * L / med / 1e9, "spread": (max(reps) - min(reps)) / med, "reps_ms": [r * 1e3 for r in reps]}


def ops(ctx):
    with ctx, torch.autograd.profiler.profile() as prof:
        for _ in range(3):
            step()
    return sorted(((e.key, round(e.cpu_time_total / 3 / 1e3, 3)) for e in prof.key_averages()), key=lambda x: -x[1])[:6]


print(json.dumps({"threads": torch.get_num_threads(), "cpus": os.cpu_count(), "steps": steps}), flush=True)
order = [("grad_on", torch.enable_grad), ("no_grad", torch.no_grad), ("inference_mode", torch.inference_mode),
         ("no_grad", torch.no_grad), ("grad_on", torch.enable_grad)]
for tag, ctx in order:
    with ctx():
        r = measure(steps)
    print(json.dumps({"mode": tag, **r}), flush=True)
for tag, ctx in order[:3]:
    print(json.dumps({"mode": tag, "ops_ms": ops(ctx())}), flush=True)
# the same with the thread count re-set before each mode (what round 2's child did)
for tag, ctx in (("grad_on", torch.enable_grad), ("no_grad", torch.no_grad)):
    torch.set_num_threads(torch.get_num_threads())
    with ctx():
        r = measure(steps)
    print(json.dumps({"mode": tag + "+set_num_threads", **r}), flush=True)
# requires_grad off on the weight: the other way PARAM could have been run
emb.weight.requires_grad_(False)
with torch.enable_grad():
    print(json.dumps({"mode": "grad_on, weight.requires_grad=False", **measure(steps)}), flush=True)
