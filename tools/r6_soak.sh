#!/bin/bash
# tools/r6_soak.sh [fuzz seeds] [hybrid seeds] [mixed-dim seeds]: the round's final kernels under many random requests (fuzz: 16-bit tables,
# Adagrad, random backward tunings, short-bag requests through the compact flat-walk launch; hybrid: mid-size requests hybrid vs sorted vs
# oracle, left-overs in LDS; mixed dims: per-table lane groups vs one width vs oracle)
mkdir -p gpurun_out/r6_soak
PARAM_AMD_FUZZ_SEEDS=${1:-600} PARAM_AMD_HYBRID_FUZZ_SEEDS=${2:-100} PARAM_AMD_MIXED_SEEDS=${3:-300} timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_hybrid.py tests/test_gpu_rest.py tests/test_gpu_join_tiles.py tests/test_gpu_mixed_dims.py -x -q -m gpu > gpurun_out/r6_soak/default.log 2>&1
echo "soak rc=$?"; grep -E "passed|failed|error" gpurun_out/r6_soak/default.log | tail -2
