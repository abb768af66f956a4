#!/bin/bash
# tools/r5_visit12.sh: what makes the bag-major apply slower at the N = 8 rank shape -- tables, bags per table, flagged share: kernel times
t=${1:-r5_v12}; out=gpurun_out/$t; mkdir -p $out; export TMPDIR=/tmp
for cfg in "8 65536 10000000" "8 32768 10000000" "16 65536 10000000" "8 65536 20000000" "16 32768 10000000"; do set -- $cfg
  d=/tmp/${t}_$1_$2_$3; rm -rf $d
  (cd /tmp && PROBE_TABLES=$1 PROBE_BATCH=$2 PROBE_ROWS=$3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python $GRAFT_REPO_ROOT/tools/r5_rank_shape_probe.py > $d.log 2>&1)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== tables $1 bags/table $2 rows $3"; grep uniform $d.log | grep '"hybrid": 1' | cut -c1-330
  python - $f <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "pm::" in n and "fill_random" not in n:
        k = re.search("([a-z_0-9]+_kernel)", n).group(1)
        print("  %-26s calls %4s max_us %9.1f" % (k, r["Calls"], float(r["MaxNs"]) / 1e3))
PY
done 2>&1 | tee $out/summary.txt
