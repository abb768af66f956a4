#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
rm -f gpurun_out/r2r_sweep.jsonl
for cfg in "0 0" "24 0" "16 0" "40 0" "0 2" "0 8" "24 2"; do set -- $cfg
BPB=$1 UNROLL=$2 timeout 600 python tools/stage_probe.py >> gpurun_out/r2r_sweep.jsonl 2>> gpurun_out/r2r.err
done
cat gpurun_out/r2r_sweep.jsonl
