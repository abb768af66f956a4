#!/bin/bash
# tools/r5_soak.sh [fuzz seeds] [hybrid seeds]: the round's final kernels under many random requests (fuzz: 16-bit tables, Adagrad, random
# backward tunings; hybrid: mid-size requests hybrid vs sorted vs oracle), once as built and once with the rows dealt out for every request
mkdir -p gpurun_out/r5_soak
PARAM_AMD_FUZZ_SEEDS=${1:-600} PARAM_AMD_HYBRID_FUZZ_SEEDS=${2:-100} timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_hybrid.py tests/test_gpu_join_tiles.py -x -q -m gpu > gpurun_out/r5_soak/default.log 2>&1
echo "default rc=$?"; grep -E "passed|failed|error" gpurun_out/r5_soak/default.log | tail -2
PARAM_AMD_HYB_PART=1 PARAM_AMD_HYBRID_FUZZ_SEEDS=${2:-100} timeout 900 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu > gpurun_out/r5_soak/part1.log 2>&1
echo "dealt-out rc=$?"; grep -E "passed|failed|error" gpurun_out/r5_soak/part1.log | tail -2
