#!/bin/bash
# tools/r5_visit5.sh: the sort's deeper look-back trips -- sort / hybrid / segsort tests, then backward timelines (Criteo, fp32 Zipf + uniform)
t=${1:-r5_v5}; mkdir -p gpurun_out/$t
timeout 900 python -m pytest tests/test_gpu_segsort.py tests/test_gpu_sort.py tests/test_gpu_hybrid.py -x -q -m gpu > gpurun_out/$t/pytest_sort.log 2>&1; tail -3 gpurun_out/$t/pytest_sort.log
TL_ROWS=12 PROBE_SETTINGS=-1 PROBE_ARGS="--workload criteo --batch 8192" bash tools/r4_timeline.sh ${t}_criteo uniform,zipf1.05
TL_ROWS=12 PROBE_SETTINGS=0 PROBE_ARGS="--tables 48" bash tools/r4_timeline.sh ${t}_fp32 zipf1.05,uniform
