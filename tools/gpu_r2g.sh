#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for nt in 2 1 0; do
timeout 600 python tools/bwd_probe.py --configs "0,1,1,1;0,1,1,2" --nt $nt >> gpurun_out/r2g_policy.jsonl 2>> gpurun_out/r2g.err
done
cat gpurun_out/r2g_policy.jsonl; tail -3 gpurun_out/r2g.err
