#!/bin/bash
# fused key building in the first radix pass: parity (sort + parity files), timing with / without
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r2z_fused_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r2z_fused_tests.log | tail -3
