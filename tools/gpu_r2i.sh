#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd $REPO
(PARAM_AMD_FUZZ_SEEDS=400 timeout 1500 python -m pytest tests/test_gpu_sort.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -12) > $OUT/r2i_soak.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r2i_bench_line.json 2> $OUT/r2i_bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r2i_bench_line_run2.json 2> $OUT/r2i_bench_line_run2.err
tail -4 $OUT/r2i_soak.log
for f in bench_line bench_line_run2; do python - <<EOF
import json
d=json.load(open("$OUT/r2i_$f.json"))
c=d["cpu_baseline"]
print("$f", "value", d["value"]/1e9, "roof", d["roofline"]["frac"], "cpu", c["value"], c.get("best_mode"), c.get("sample","")[:80])
for t,r in c.get("children",{}).items():
    print("  ", t, {m:(round(x["lookups_per_s"]/1e6,1), round(x["spread"],3), x["threads"]) for m,x in r.get("modes",{}).items()} if "modes" in r else r, r.get("c_oracle_1core"))
EOF
done
tail -3 $OUT/r2i_bench_line.err
