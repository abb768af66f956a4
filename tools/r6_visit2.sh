#!/bin/bash
# Round 6, GPU visit 2: phases of hyb_rest_kernel (experiment build) + bf16 kernel timeline
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6_v2
mkdir -p $O
PARAM_AMD_LIB=$PWD/build/libparam_amd_exp.so timeout 300 python tools/r6_rest_trace.py --dtype fp32 --tables 48 > $O/rest_trace_fp32.json 2> $O/rest_trace_fp32.err
cat $O/rest_trace_fp32.json; tail -2 $O/rest_trace_fp32.err
PARAM_AMD_LIB=$PWD/build/libparam_amd_exp.so timeout 300 python tools/r6_rest_trace.py --dtype bf16 --tables 64 > $O/rest_trace_bf16.json 2> $O/rest_trace_bf16.err
cat $O/rest_trace_bf16.json; tail -2 $O/rest_trace_bf16.err
