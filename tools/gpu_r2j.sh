#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
PARAM_AMD_FWD_STAGE=1 timeout 600 python tools/stage_probe.py > gpurun_out/r2j_stage.jsonl 2> gpurun_out/r2j.err
PARAM_AMD_FWD_STAGE=0 timeout 600 python tools/stage_probe.py >> gpurun_out/r2j_stage.jsonl 2>> gpurun_out/r2j.err
PARAM_AMD_FWD_STAGE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or batched or seeded" 2>&1 | tail -3
cat gpurun_out/r2j_stage.jsonl; tail -2 gpurun_out/r2j.err
