#!/bin/bash
# tools/r5_visit9.sh: rows dealt out to per-slice queues before the mark (hyb_part_kernel): parity, then PARAM_AMD_HYB_PART = 0 / 1 taking turns
t=${1:-r5_v9}; mkdir -p gpurun_out/$t
timeout 900 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_blocked.py tests/test_gpu_fuzz.py tests/test_gpu_join_tiles.py -q -x > gpurun_out/$t/tests.log 2>&1; tail -3 gpurun_out/$t/tests.log
PARAM_AMD_HYB_PART=1 timeout 900 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_fuzz.py -q -x > gpurun_out/$t/tests_part1.log 2>&1; tail -3 gpurun_out/$t/tests_part1.log
for rep in 1 2; do
for part in 0 1; do
  echo "== PARAM_AMD_HYB_PART=$part (rep $rep)"
  PARAM_AMD_HYB_PART=$part timeout 300 python tools/r4_bwd_probe.py --tables 48 --settings 2 --requests uniform 2>&1 | tail -1 | cut -c1-250
  PARAM_AMD_HYB_PART=$part timeout 300 python tools/r4_bwd_probe.py --tables 64 --dtype bf16 --settings 2 --requests uniform 2>&1 | tail -1 | cut -c1-250
  PARAM_AMD_HYB_PART=$part PROBE_TABLES=8 timeout 300 python tools/r5_rank_shape_probe.py 2>&1 | grep '"hybrid": 1' | grep uniform | cut -c1-200
  PARAM_AMD_HYB_PART=$part PROBE_TABLES=12 timeout 300 python tools/r5_rank_shape_probe.py 2>&1 | grep '"hybrid": 1' | grep uniform | cut -c1-200
done
done 2>&1 | tee gpurun_out/$t/ab.log
