#!/bin/bash
# fourth GPU visit: all gpu tests (incl. driver/plug-in layers), bench, bwd PMC, bf16 bench
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_d.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_d.log
timeout 900 python bench.py --steps 50 --warmup 5 --bwd > $OUT/bench_r1d.json 2> $OUT/bench_r1d.err; echo "bench rc=$?" >> $OUT/bench_r1d.err
timeout 900 python bench.py --steps 30 --warmup 5 --bwd --dtype bf16 --no-cpu-baseline > $OUT/bench_r1d_bf16.json 2> $OUT/bench_r1d_bf16.err
timeout 900 python bench.py --steps 30 --warmup 5 --bwd --alpha 0 --no-uniform --no-cpu-baseline > $OUT/bench_r1d_uniform.json 2> $OUT/bench_r1d_uniform.err
cd /tmp; export TMPDIR=/tmp
i=10
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- \
      python $REPO/tools/pmc_probe.py --bwd --manifest $OUT/pmc_manifest_bwd.json > $OUT/pmc_$i.log 2>&1
  echo "pmc pass $i ($c) rc=$?" >> $OUT/pmc_passes_d.txt
done
cd $REPO
tail -8 $OUT/pytest_gpu_d.log
for f in bench_r1d bench_r1d_bf16 bench_r1d_uniform; do python - <<EOF
import json
try:
    r=json.load(open("$OUT/$f.json"))
    print("$f", "value %.3g" % r["value"], "frac %.3f" % r["roofline"]["frac"], "T", r["config"]["tables_per_gpu"], "uniform", r.get("uniform",{}).get("frac"),
          "bwd", {k:(round(v,5) if isinstance(v,float) else v) for k,v in r.get("bwd_scatter_add",{}).items() if k!="method"})
except Exception as e:
    print("$f failed", e); print(open("$OUT/$f.err").read()[-1500:])
EOF
done
cat $OUT/pmc_passes_d.txt
