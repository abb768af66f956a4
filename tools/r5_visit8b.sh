#!/bin/bash
# tools/r5_visit8b.sh: the parity suites that touch the sorted apply, on the tile-joining library
t=${1:-r5_v8b}; mkdir -p gpurun_out/$t
timeout 1500 python -m pytest tests/test_gpu_join_tiles.py tests/test_gpu_parity.py tests/test_gpu_hybrid.py tests/test_gpu_blocked.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q > gpurun_out/$t/tests.log 2>&1; tail -6 gpurun_out/$t/tests.log
