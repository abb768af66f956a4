#!/bin/bash
# Round 6, GPU visit 3: staging kernel + LDS kernel (same stream / side stream): parity, A/B, phases, timeline
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6_v3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rest.py tests/test_gpu_hybrid.py -x -q -m gpu > $O/pytest_rest_hybrid.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_rest_hybrid.txt
tail -8 $O/pytest_rest_hybrid.txt
timeout 600 python tools/r6_rest_probe.py --dtype fp32 --tables 48 --requests uniform > $O/rest_ab_fp32.jsonl 2> $O/rest_ab_fp32.err
cut -c1-150 $O/rest_ab_fp32.jsonl; tail -3 $O/rest_ab_fp32.err
timeout 600 python tools/r6_rest_probe.py --dtype bf16 --tables 64 --requests uniform > $O/rest_ab_bf16.jsonl 2> $O/rest_ab_bf16.err
cut -c1-150 $O/rest_ab_bf16.jsonl; tail -3 $O/rest_ab_bf16.err
timeout 600 python tools/r6_rest_probe.py --dtype fp32 --tables 48 --requests zipf1.05 --reps 2 > $O/rest_ab_fp32_zipf.jsonl 2> $O/rest_ab_fp32_zipf.err
cut -c1-150 $O/rest_ab_fp32_zipf.jsonl
for r in 1 2; do
PARAM_AMD_LIB=$PWD/build/libparam_amd_exp.so timeout 300 python tools/r6_rest_trace.py --dtype fp32 --tables 48 --rest $r > $O/rest_trace_fp32_m$r.json 2> $O/rest_trace_fp32_m$r.err
cat $O/rest_trace_fp32_m$r.json
done
PARAM_AMD_LIB=$PWD/build/libparam_amd_exp.so timeout 300 python tools/r6_rest_trace.py --dtype bf16 --tables 64 --rest 1 > $O/rest_trace_bf16_m1.json 2> $O/rest_trace_bf16_m1.err
cat $O/rest_trace_bf16_m1.json
cd /tmp
for r in 1 2; do
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_fp32_m$r -o rest -- python $GRAFT_REPO_ROOT/tools/r6_rest_probe.py --dtype fp32 --tables 48 --requests uniform --settings $r --reps 1 --iters 20 > $GRAFT_REPO_ROOT/$O/prof_fp32_m$r.log 2>&1
done
