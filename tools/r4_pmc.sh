#!/bin/bash
# tools/r4_pmc.sh: separate rocprofv3 --pmc passes (no other trace domain) over tools/pmc_probe.py --bwd; summary -> profiles/r04_pmc_summary.json
out=gpurun_out/r4_pmc
mkdir -p "$out/pmc"
export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); d=/tmp/r4pmc_$i; rm -rf "$d"
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- \
      python "$GRAFT_REPO_ROOT/tools/pmc_probe.py" --bwd --manifest "$GRAFT_REPO_ROOT/$out/pmc/pmc_manifest.json" > "$d.log" 2>&1)
  mkdir -p "$out/pmc/pmc_$i"
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/pmc/pmc_$i/pmc_counter_collection.csv" || tail -5 "$d.log"
done
python tools/parse_pmc.py "$out/pmc" r04 > "$out/pmc_summary.json" 2> "$out/pmc_parse.err"
cp profiles/r04_pmc_summary.json profiles/pmc_traffic.json "$out/"
python - "$out/pmc_summary.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("calibration",r["calibration"])
for k,v in r["kernels"].items():
    print(k,{x:(round(v[x],4) if isinstance(v[x],float) else v[x]) for x in ("hbm_bytes_per_launch","algorithmic_bytes","hbm_over_algorithmic","l2_hit_rate","fetch_bytes_calibrated","write_bytes_calibrated") if x in v})
PY
