#!/bin/bash
# tools/r5_visit14.sh: the mark kernel's scan without the bounds predicate on whole batches: parity, then the library before / after taking turns
t=${1:-r5_v14}; mkdir -p gpurun_out/$t
timeout 900 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_blocked.py -q -x > gpurun_out/$t/tests.log 2>&1; tail -2 gpurun_out/$t/tests.log
for rep in 1 2 3; do
for lib in build/libparam_amd_base.so param_amd/libparam_amd.so; do
  echo "== $lib (rep $rep)"
  PARAM_AMD_LIB=$lib timeout 300 python tools/r4_bwd_probe.py --tables 48 --settings 2 --requests uniform 2>&1 | tail -1 | cut -c1-220
  PARAM_AMD_LIB=$lib timeout 300 python tools/r4_bwd_probe.py --tables 64 --dtype bf16 --settings 2 --requests uniform 2>&1 | tail -1 | cut -c1-220
done
done 2>&1 | tee gpurun_out/$t/ab.log
