#!/bin/bash
# second GPU visit: rocprofv3 stats + PMC passes, plain bench, diagnostics
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "TCC_|FETCH_SIZE|WRITE_SIZE|TCP_T" | head -150 > $OUT/counters.txt 2>&1

# 1. kernel-trace + stats of the bench command itself
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- \
    python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof_bench.err
echo "stats rc=$?" >> $OUT/prof_bench.err

# 2. PMC passes (each in its own run, kernel-trace only)
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- \
      python $REPO/tools/pmc_probe.py --manifest $OUT/pmc_manifest.json > $OUT/pmc_$i.log 2>&1
  echo "pmc pass $i ($c) rc=$?" >> $OUT/pmc_passes.txt
done

cd $REPO
# 3. plain bench (no profiler attached), with backward
timeout 900 python bench.py --steps 50 --warmup 5 --bwd > $OUT/bench_r1b.json 2> $OUT/bench_r1b.err; echo "bench rc=$?" >> $OUT/bench_r1b.err
# 4. diagnostics
timeout 900 python tools/diag.py > $OUT/diag_r1b.jsonl 2> $OUT/diag_r1b.err; echo "diag rc=$?" >> $OUT/diag_r1b.err
# 5. tests again (library changed)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_b.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_b.log

find $OUT/prof_stats $OUT/pmc_1 -type f | head -20
cat $OUT/pmc_passes.txt; tail -2 $OUT/prof_bench.err; cat $OUT/bench_r1b.json | cut -c1-600; tail -3 $OUT/pytest_gpu_b.log; cat $OUT/diag_r1b.jsonl; tail -3 $OUT/diag_r1b.err
du -sh $OUT
