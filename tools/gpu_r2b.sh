#!/bin/bash
# round-2 visit B: own radix sort + backward tuning matrix
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -25) > $O/r2b_sort_tests.log
(timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_sort.py 2>&1 | tail -15) > $O/r2b_pytest.log
timeout 900 python tools/bwd_probe.py > $O/r2b_bwd_probe.jsonl 2> $O/r2b_bwd_probe.err
timeout 600 python tools/bwd_probe.py --dtype bf16 --tables 64 --iters 6 > $O/r2b_bwd_probe_bf16.jsonl 2> $O/r2b_bwd_probe_bf16.err
tail -5 $O/r2b_sort_tests.log; tail -5 $O/r2b_pytest.log; cat $O/r2b_bwd_probe.jsonl; tail -3 $O/r2b_bwd_probe.err; cat $O/r2b_bwd_probe_bf16.jsonl; tail -3 $O/r2b_bwd_probe_bf16.err
