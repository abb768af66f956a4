#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python tools/bwd_probe.py --configs "0,1,1,0;0,1,0,0" --layout tbd > gpurun_out/r2d_tbd.jsonl 2> gpurun_out/r2d_tbd.err
timeout 600 python tools/bwd_probe.py --configs "0,1,1,0;0,1,0,0" --layout tbd --dtype bf16 --tables 64 > gpurun_out/r2d_tbd_bf16.jsonl 2>> gpurun_out/r2d_tbd.err
cat gpurun_out/r2d_tbd.jsonl gpurun_out/r2d_tbd_bf16.jsonl; tail -3 gpurun_out/r2d_tbd.err
