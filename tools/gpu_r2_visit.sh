#!/bin/bash
# round-2 full visit: GPU tests, smoke, the bench lines (twice: CPU-baseline stability), the same command under
# rocprofv3 --kernel-trace --stats, PMC passes (separate runs per counter group), N>1 code path on one GPU, bf16 line.
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $OUT/${TAG}_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $OUT/${TAG}_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line_run2.json 2> $OUT/${TAG}_bench_line_run2.err
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $OUT/${TAG}_bench_line_bf16_T64.json 2> $OUT/${TAG}_bench_bf16.err
timeout 600 python bench.py --steps 20 --warmup 5 --workload criteo --no-cpu-baseline > $OUT/${TAG}_bench_line_criteo.json 2> $OUT/${TAG}_bench_criteo.err
timeout 600 python bench.py --dist-debug --tables 26 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_distdebug_26tables.json 2> $OUT/${TAG}_dd26.err
timeout 600 python bench.py --dist-debug --workload criteo --steps 10 --warmup 3 --no-cpu-baseline --lookup-cus 224 > $OUT/${TAG}_distdebug_criteo_cu224.json 2> $OUT/${TAG}_ddc.err
cd /tmp; export TMPDIR=/tmp
# the dominant kernel alone: every embbag_fwd_kernel launch of these two runs is the headline launch (Zipf) / the
# roofline-defining launch (uniform), so the kernel's average in the stats file is directly comparable with the bench line
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_zipf -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --only-headline > $OUT/${TAG}_headline_zipf_under_rocprofv3.json 2> $OUT/${TAG}_prof_zipf.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_uniform -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --only-headline --alpha 0 > $OUT/${TAG}_headline_uniform_under_rocprofv3.json 2> $OUT/${TAG}_prof_uniform.err
# the whole default command (forward both layouts and both distributions, backward, fwd+bwd): per-kernel totals
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_bench -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_under_rocprofv3.json 2> $OUT/${TAG}_prof_bench.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc/pmc_$i -o pmc -- \
      python $REPO/tools/pmc_probe.py --bwd --manifest $OUT/${TAG}_pmc/pmc_manifest.json > $OUT/${TAG}_pmc_$i.log 2>&1
done
cd $REPO
tail -3 $OUT/${TAG}_pytest.log; cat $OUT/${TAG}_smoke.log
for f in bench_line bench_line_run2 bench_line_bf16_T64 bench_line_criteo distdebug_26tables distdebug_criteo_cu224 bench_line_under_rocprofv3 headline_zipf_under_rocprofv3 headline_uniform_under_rocprofv3; do echo "== $f: $(head -c 300 $OUT/${TAG}_$f.json)"; done
find $OUT/${TAG}_prof_bench -name "*kernel_stats.csv" | head -2; ls $OUT/${TAG}_pmc
