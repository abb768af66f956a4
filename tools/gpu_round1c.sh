#!/bin/bash
# third GPU visit: sorted backward parity + timing, clean rocprofv3 stats of the bench command
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_c.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_c.log
timeout 900 python bench.py --steps 50 --warmup 5 --bwd > $OUT/bench_r1c.json 2> $OUT/bench_r1c.err; echo "bench rc=$?" >> $OUT/bench_r1c.err
cd /tmp; export TMPDIR=/tmp
# same command as the bench line (headline Zipf launches only: --no-uniform) under kernel-trace + stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_c -o bench -- \
    python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-uniform > $OUT/prof_bench_c.json 2> $OUT/prof_bench_c.err
echo "stats rc=$?" >> $OUT/prof_bench_c.err
# backward under kernel-trace + stats (sort kernels + apply kernel durations)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_bwd -o bench -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-uniform --bwd > $OUT/prof_bench_bwd.json 2> $OUT/prof_bench_bwd.err
cd $REPO
tail -5 $OUT/pytest_gpu_c.log; cat $OUT/bench_r1c.json | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value', r['value'], 'roofline', r['roofline']['achieved'], r['roofline']['frac'], 'traffic', r['roofline']['traffic'])
print('uniform', r.get('uniform')); print('bwd', r.get('bwd_scatter_add')); print('cpu', {k:v for k,v in r.get('cpu_baseline',{}).items() if k!='modes'})
print('cpu modes', r.get('cpu_baseline',{}).get('modes'))"
tail -2 $OUT/bench_r1c.err
head -12 $OUT/prof_stats_bwd/bench_kernel_stats.csv | cut -c1-200
