#!/bin/bash
# Round 6, GPU visit 1: the LDS left-over kernel -- parity first, then same-box A/B and a kernel trace.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6_v1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rest.py tests/test_gpu_hybrid.py -x -q -m gpu > $O/pytest_rest_hybrid.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_rest_hybrid.txt
tail -15 $O/pytest_rest_hybrid.txt
timeout 600 python tools/r6_rest_probe.py --dtype fp32 --tables 48 > $O/rest_ab_fp32.jsonl 2> $O/rest_ab_fp32.err
tail -12 $O/rest_ab_fp32.jsonl; tail -3 $O/rest_ab_fp32.err
timeout 600 python tools/r6_rest_probe.py --dtype bf16 --tables 64 > $O/rest_ab_bf16.jsonl 2> $O/rest_ab_bf16.err
tail -12 $O/rest_ab_bf16.jsonl; tail -3 $O/rest_ab_bf16.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_fp32 -o rest -- python $GRAFT_REPO_ROOT/tools/r6_rest_probe.py --dtype fp32 --tables 48 --requests uniform --settings 1 --reps 1 --iters 20 > $GRAFT_REPO_ROOT/$O/prof_fp32.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_fp32 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {}'
find $O/prof_fp32 -name "*.csv" ! -name "*kernel_stats.csv" -size +8M -delete
