#!/usr/bin/env python3
"""Host facts that decide how stable a CPU EmbeddingBag timing can be on the GPU box (cgroup CPU quota, affinity, load), then
the reference engine at several thread counts (same protocol as bench.py's cpu child: 8 index sets in turn)."""
import json, os, statistics, sys, time
import torch

def rd(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"

facts = {"cpu.max": rd("/sys/fs/cgroup/cpu.max"), "cpuset.cpus.effective": rd("/sys/fs/cgroup/cpuset.cpus.effective"),
         "cpu.stat": rd("/sys/fs/cgroup/cpu.stat").replace("\n", "; "), "loadavg": rd("/proc/loadavg"), "affinity": len(os.sched_getaffinity(0)),
         "cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(), "OMP": {k: v for k, v in os.environ.items() if k.startswith(("OMP", "GOMP", "KMP", "MKL"))},
         "parallel_info": torch.__config__.parallel_info().replace("\n", " | ")[:400]}
print(json.dumps(facts), flush=True)
R, D, B, L, K = 10_000_000, 128, 8192, 20, 8
W = torch.empty(R, D).normal_()
sets = [torch.randint(0, R, (B * L,)) for _ in range(K)]
off = torch.arange(B, dtype=torch.int64) * L
emb = torch.nn.EmbeddingBag(R, D, mode="sum", _weight=W)
k = [0]
def step():
    k[0] += 1
    return emb(sets[k[0] % K], off)
for nthr in (1, 8, 16, 32, 64, 96, 128):
    torch.set_num_threads(nthr)
    for _ in range(3):
        step()
    reps = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(64):
            step()
        reps.append((time.perf_counter() - t0) / 64)
    med = statistics.median(reps)
    print(json.dumps({"threads": nthr, "ms_per_step": round(med * 1e3, 4), "G_lookups_s": round(B * L / med / 1e9, 4),
                      "spread": round((max(reps) - min(reps)) / med, 3), "min_ms": round(min(reps) * 1e3, 4)}), flush=True)
print(json.dumps({"cpu.stat_after": rd("/sys/fs/cgroup/cpu.stat").replace("\n", "; ")}), flush=True)
