#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; cd $REPO
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-bwd > $OUT/r2n_bench_$i.json 2> $OUT/r2n_bench_$i.err
python - <<EOF
import json
d=json.load(open("$OUT/r2n_bench_$i.json")); c=d["cpu_baseline"]
print("run $i value", round(d["value"]/1e9,2), "frac", round(d["roofline"]["frac"],4), "cpu", round(c["value"]/1e6,1), c.get("best_mode"), {m:(round(x["lookups_per_s"]/1e6,1), round(x["spread"],2), x["threads"]) for m,x in c["child"].get("modes",{}).items()}, c["child"].get("c_oracle_1core"))
EOF
done
