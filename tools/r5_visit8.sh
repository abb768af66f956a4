#!/bin/bash
# tools/r5_visit8.sh: tile-level partial sums (join_tiles) -- parity tests, then the old library (build/libparam_amd_base.so) and the
# new one taking turns on this box: rank shape N = 8 (Zipf head rows of 51 K lookups), benchmark shape, Criteo
t=${1:-r5_v8}; mkdir -p gpurun_out/$t
timeout 1500 python -m pytest tests/test_gpu_join_tiles.py tests/test_gpu_parity.py tests/test_gpu_hybrid.py tests/test_gpu_blocked.py tests/test_gpu_fuzz.py -q -x -k "not full_size and not 2_31" > gpurun_out/$t/tests.log 2>&1; tail -4 gpurun_out/$t/tests.log
for rep in 1 2; do
for lib in build/libparam_amd_base.so param_amd/libparam_amd.so; do
  echo "== $lib (rep $rep)"
  PARAM_AMD_LIB=$lib PROBE_TABLES=8 timeout 300 python tools/r5_rank_shape_probe.py 2>&1 | grep zipf | cut -c1-260
  PARAM_AMD_LIB=$lib timeout 300 python tools/r4_bwd_probe.py --tables 48 --settings 2 --requests zipf1.05 2>&1 | tail -1 | cut -c1-300
  PARAM_AMD_LIB=$lib timeout 300 python tools/r4_bwd_probe.py --workload criteo --settings 2 --requests uniform,zipf1.05 2>&1 | tail -2 | cut -c1-300
done
done 2>&1 | tee gpurun_out/$t/ab.log
