#!/bin/bash
# tools/r5_visit15.sh: pass 0's status rows turned into inclusive prefixes by the scan kernel (one-trip look-back walks in pass 0): parity of
# the sort suites, then the library before / after taking turns
t=${1:-r5_v15}; mkdir -p gpurun_out/$t
timeout 1200 python -m pytest tests/test_gpu_segsort.py tests/test_gpu_sort.py tests/test_gpu_hybrid.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -x > gpurun_out/$t/tests.log 2>&1; tail -2 gpurun_out/$t/tests.log
for rep in 1 2 3; do
for lib in build/libparam_amd_base.so param_amd/libparam_amd.so; do
  echo "== $lib (rep $rep)"
  PARAM_AMD_LIB=$lib timeout 300 python tools/r4_bwd_probe.py --tables 48 --settings 2 --requests zipf1.05 2>&1 | tail -1 | cut -c1-230
  PARAM_AMD_LIB=$lib timeout 300 python tools/r4_bwd_probe.py --workload criteo --settings 2 --requests uniform,zipf1.05 2>&1 | tail -2 | cut -c1-230
done
done 2>&1 | tee gpurun_out/$t/ab.log
