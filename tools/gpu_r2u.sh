#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/fwd_quant_probe.py > gpurun_out/r2u_fwd_quant_probe.jsonl 2> gpurun_out/r2u_fwd_quant_probe.err
cat gpurun_out/r2u_fwd_quant_probe.jsonl; tail -5 gpurun_out/r2u_fwd_quant_probe.err
