#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
rm -f gpurun_out/r2k_stage.jsonl
for cfg in "1 0" "1 16" "1 64" "0 0" "0 16"; do set -- $cfg
PARAM_AMD_FWD_STAGE=$1 BPB=$2 timeout 600 python tools/stage_probe.py >> gpurun_out/r2k_stage.jsonl 2>> gpurun_out/r2k.err
done
cat gpurun_out/r2k_stage.jsonl
