#!/bin/bash
# first visit to the GPU box: environment facts, smoke, gpu tests, bench, tuning sweep
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{ rocm-smi --showmeminfo vram 2>&1 | head -8; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; nproc; free -g | head -2; } > gpurun_out/env.txt 2>&1
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 --bwd > gpurun_out/bench_r1a.json 2> gpurun_out/bench_r1a.err; echo "bench rc=$?" >> gpurun_out/bench_r1a.err
timeout 1200 python tools/sweep.py --steps 15 --bwd --out gpurun_out/sweep_r1a.jsonl > gpurun_out/sweep_r1a.log 2>&1; echo "sweep rc=$?" >> gpurun_out/sweep_r1a.log
tail -3 gpurun_out/smoke.log; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_r1a.json; tail -3 gpurun_out/bench_r1a.err; tail -5 gpurun_out/sweep_r1a.log
