#!/bin/bash
# round 2, visit t: forward with quantised output -- parity, whole GPU suite, headline bench (no regression in the fp32 forward)
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rowquant.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r2t_rowquant_tests.log
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r2t_gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 5 --only-headline > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err
cat gpurun_out/r2t_rowquant_tests.log gpurun_out/r2t_gpu_suite.log gpurun_out/r2t_bench.json
tail -3 gpurun_out/r2t_bench.err
