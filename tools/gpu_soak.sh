#!/bin/bash
# soak: the fuzz files with many seeds (random requests incl. 16-bit tables through the sorted apply, Adagrad, random backward tunings)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PARAM_AMD_FUZZ_SEEDS=${1:-1500} timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_sort.py tests/test_rowquant.py -x -q -m gpu > gpurun_out/soak.log 2>&1
echo "rc=$?"; grep -E "passed|failed|error" gpurun_out/soak.log | tail -3
