#!/bin/bash
TAG=${1:-r02b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line_run2.json 2> $OUT/${TAG}_bench_line_run2.err
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_zipf -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --only-headline > $OUT/${TAG}_headline_zipf_under_rocprofv3.json 2> $OUT/${TAG}_prof_zipf.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_uniform -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --only-headline --alpha 0 > $OUT/${TAG}_headline_uniform_under_rocprofv3.json 2> $OUT/${TAG}_prof_uniform.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_full -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_full_under_rocprofv3.json 2> $OUT/${TAG}_prof_full.err
cd $REPO
for f in bench_line bench_line_run2; do python - <<EOF
import json
d=json.load(open("$OUT/${TAG}_$f.json"))
c=d["cpu_baseline"]
print("$f", "value", d["value"]/1e9, "roof", d["roofline"]["frac"], "cpu", c["value"], c.get("best_mode"))
for t,r in c.get("children",{}).items():
    print("  ", t, {m:(round(x["lookups_per_s"]/1e6,1), round(x["spread"],3)) for m,x in r.get("modes",{}).items()} if "modes" in r else r)
EOF
done
for k in zipf uniform full; do f=$(find $OUT/${TAG}_prof_$k -name "*kernel_stats.csv" | head -1); echo "== $k $f"; head -4 "$f" | cut -c1-260; done
