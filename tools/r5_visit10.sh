#!/bin/bash
# tools/r5_visit10.sh: per-kernel times of the hybrid backward with the rows dealt out first, rank shape N = 8 and benchmark shape
t=${1:-r5_v10}; out=gpurun_out/$t; mkdir -p $out; export TMPDIR=/tmp
for cfg in "8 1" "8 0" "24 1"; do set -- $cfg
  d=/tmp/${t}_$1_$2; rm -rf $d
  (cd /tmp && PARAM_AMD_HYB_PART=$2 PROBE_TABLES=$1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python $GRAFT_REPO_ROOT/tools/r5_rank_shape_probe.py > $d.log 2>&1)
  f=$(find $d -name "*kernel_stats.csv" | head -1); cp $f $out/T$1_part$2.kernel_stats.csv
  echo "== tables $1 part $2"; python - $f <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "pm::" not in n or "fill_random" in n: continue
    m = re.search(r"(\w+_kernel)", n)
    print(f'{m.group(1):28s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
PY
done 2>&1 | tee $out/summary.txt
