#!/bin/bash
# round 6, visit 11: the hybrid tests on the chip-filling rule, the mark kernel with 16-load batches (kernel trace), the backward A/B probe
O=gpurun_out/r6_v13; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_rest.py tests/test_gpu_blocked.py tests/test_gpu_join_tiles.py tests/test_gpu_segsort.py tests/test_gpu_sort.py tests/test_capi_symbols.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 600 python tools/r6_rest_probe.py --dtype fp32 --tables 48 --settings 1 > $O/bwd_fp32.jsonl 2> $O/err.txt; cut -c1-200 $O/bwd_fp32.jsonl
timeout 600 python tools/r6_rest_probe.py --dtype bf16 --tables 64 --settings 1 --requests uniform > $O/bwd_bf16.jsonl 2>> $O/err.txt; cut -c1-200 $O/bwd_bf16.jsonl
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o k -- python $R/tools/r6_rest_probe.py --dtype fp32 --tables 48 --requests uniform --settings 1 --reps 1 --iters 20 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-160; cp $f $O/kernel_stats_fp32_bwd_uniform.csv
find $O/prof -type f -size +2M -delete
