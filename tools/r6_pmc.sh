#!/bin/bash
# tools/r6_pmc.sh [tag] [workloads]: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one counter per run, --kernel-trace only --
# never combined with another trace domain) over tools/r6_pmc_probe.py, one phase of one workload per process.
# Summary: tools/r5_parse_pmc.py gpurun_out/<tag> -> profiles/r06_pmc_summary.json + keys for profiles/pmc_traffic.json.
tag=${1:-r6_pmc}; wls=${2:-fp32,bf16,criteo,mixed}
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
run() { # counter workload phase
  d=/tmp/${tag}_$2_$3_$1; rm -rf "$d"
  (cd /tmp && timeout 600 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d "$d" -o pmc -- \
      python "$GRAFT_REPO_ROOT/tools/r6_pmc_probe.py" --workload $2 --phase $3 > "$GRAFT_REPO_ROOT/$out/$2.$3.$1.manifest.json" 2> "$d.log")
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/$2.$3.$1.csv"; else echo "no counters: $2 $3 $1"; tail -3 "$d.log"; fi
}
for c in FETCH_SIZE WRITE_SIZE; do run $c any calib; done
for w in ${wls//,/ }; do for ph in fwd_uniform fwd_zipf bwd_uniform bwd_zipf; do for c in FETCH_SIZE WRITE_SIZE; do run $c $w $ph; done; done; done
python tools/r5_parse_pmc.py "$out" > "$out/pmc_summary.json" 2> "$out/parse.err"; tail -3 "$out/parse.err"
python - "$out/pmc_summary.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("calibration", r["calibration"])
for k, v in r["phases"].items():
    print(k, {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("fabric_bytes_per_step", "algorithmic_bytes_per_step", "traffic_over_algorithmic")})
PY
