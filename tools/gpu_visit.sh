#!/bin/bash
# One visit to the GPU box: gpu tests, smoke, the bench lines, rocprofv3 stats + PMC passes.
# Usage (from the build container): gpurun --timeout 3000 -- 'bash tools/gpu_visit.sh [tag]'
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
timeout 900 python bench.py --bwd > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 900 python bench.py --bwd --dtype bf16 --no-cpu-baseline > $OUT/${TAG}_bench_bf16.json 2>> $OUT/${TAG}_bench.err
timeout 900 python bench.py --workload criteo --bwd --no-cpu-baseline > $OUT/${TAG}_bench_criteo.json 2>> $OUT/${TAG}_bench.err
timeout 900 python bench.py --dist-debug --steps 20 --no-cpu-baseline > $OUT/${TAG}_bench_distdebug.json 2>> $OUT/${TAG}_bench.err
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_fwd -o bench -- \
    python $REPO/bench.py --no-cpu-baseline --no-uniform > $OUT/${TAG}_bench_under_rocprof.json 2>> $OUT/${TAG}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_bwd -o bench -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-uniform --bwd > /dev/null 2>> $OUT/${TAG}_bench.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc/pmc_$i -o pmc -- \
      python $REPO/tools/pmc_probe.py --bwd --manifest $OUT/${TAG}_pmc/pmc_manifest.json > $OUT/${TAG}_pmc_$i.log 2>&1
done
cd $REPO
tail -3 $OUT/${TAG}_pytest_gpu.log; tail -1 $OUT/${TAG}_smoke.log
python - <<PY
import json
for f in ("bench", "bench_bf16", "bench_criteo", "bench_distdebug"):
    try:
        r = json.load(open("$OUT/${TAG}_%s.json" % f))
        print(f, "value %.4g" % r["value"], "frac %.3f" % r["roofline"]["frac"], "uniform", r.get("uniform", {}).get("frac"),
              "bwd", r.get("bwd_scatter_add", {}).get("frac"), "fwd+bwd", r.get("fwd_bwd_step", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
