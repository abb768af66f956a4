#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python -X faulthandler -m pytest tests -x -v -m gpu > gpurun_out/r2x_run$i.log 2>&1
  echo "run $i rc=$? $(grep -c PASSED gpurun_out/r2x_run$i.log) passed"
  if grep -q "Fatal Python error\|Aborted\|Segmentation\|core dumped" gpurun_out/r2x_run$i.log; then
    grep -n "PASSED\|FAILED" gpurun_out/r2x_run$i.log | tail -3
    grep -n -B5 -A45 "Fatal Python error" gpurun_out/r2x_run$i.log | cut -c1-250 | head -120
  fi
done
