#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2c_prof
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r2c_prof -o bwd -- python tools/bwd_probe.py --configs "0,1,1,0" --iters 10 > gpurun_out/r2c_probe.jsonl 2> gpurun_out/r2c_probe.err
find gpurun_out/r2c_prof -name "*kernel_stats*" | head; f=$(find gpurun_out/r2c_prof -name "*kernel_stats.csv" | head -1); head -30 "$f"
cat gpurun_out/r2c_probe.jsonl
