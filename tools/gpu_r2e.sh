#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -25) > $O/r2e_sort_tests.log
(timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_sort.py 2>&1 | tail -25) > $O/r2e_pytest.log
timeout 900 python tools/bwd_probe.py > $O/r2e_bwd_probe.jsonl 2> $O/r2e_bwd_probe.err
timeout 600 python tools/bwd_probe.py --dtype bf16 --tables 64 --iters 6 --configs "0,1,1,1;0,1,1,2" > $O/r2e_bwd_probe_bf16.jsonl 2> $O/r2e_bwd_probe_bf16.err
tail -7 $O/r2e_sort_tests.log; tail -7 $O/r2e_pytest.log; cat $O/r2e_bwd_probe.jsonl; tail -3 $O/r2e_bwd_probe.err; cat $O/r2e_bwd_probe_bf16.jsonl; tail -3 $O/r2e_bwd_probe_bf16.err
