#!/bin/bash
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcn8_$c; rm -rf $d
  (cd /tmp && PROBE_TABLES=8 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o pmc -- python $GRAFT_REPO_ROOT/tools/r5_rank_shape_probe.py > /dev/null 2> $d.log)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "pm::" in n and "fill_random" not in n:
        k = n.replace("void ", "").replace("pm::(anonymous namespace)::", "").split("<")[0].split("(")[0]
        agg[k].append(float(r["Counter_Value"]))
scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
for k, v in agg.items():
    # the probe runs uniform (hybrid off: 20 steps, on: 20 steps) then zipf: print the distinct per-launch values' median groups
    v2 = sorted(v)
    print(sys.argv[2], k, "launches", len(v), "MB per launch (min / median / max)", round(v2[0] * 1024 * scale / 1e6, 1), round(v2[len(v2) // 2] * 1024 * scale / 1e6, 1), round(v2[-1] * 1024 * scale / 1e6, 1))
PY
done
