#!/bin/bash
# tools/r6_kstats.sh [tag] [workloads]: rocprofv3 --kernel-trace --stats of ONE phase of ONE workload per run (tools/r6_pmc_probe.py, 10 steps):
# per-kernel average durations from which every fraction of the bench line's bf16 / Criteo / fp32 blocks can be recomputed
# -> gpurun_out/<tag>/<workload>.<phase>.kernel_stats.csv (+ a readable summary); copy to profiles/r06_kstats_*.
tag=${1:-r6_kstats}; wls=${2:-fp32,bf16,criteo,mixed}
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
for w in ${wls//,/ }; do for ph in fwd_uniform fwd_zipf bwd_uniform bwd_zipf; do
  d=/tmp/${tag}_${w}_$ph; rm -rf "$d"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o ks -- python "$GRAFT_REPO_ROOT/tools/r6_pmc_probe.py" --workload $w --phase $ph --iters 10 > "$GRAFT_REPO_ROOT/$out/$w.$ph.manifest.json" 2> "$d.log")
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/$w.$ph.kernel_stats.csv"; else echo "no stats: $w $ph"; tail -3 "$d.log"; fi
done; done
python - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
summary = {}
for p in sorted(glob.glob(os.path.join(out, "*.kernel_stats.csv"))):
    w, ph = os.path.basename(p).split(".")[:2]
    man = json.loads(open(os.path.join(out, f"{w}.{ph}.manifest.json")).read().strip().splitlines()[-1])
    ks, tot = {}, 0.0
    for r in csv.DictReader(open(p)):
        if "pm::" not in r["Name"] or "fill_random" in r["Name"]:
            continue
        name = r["Name"].replace("void ", "").replace("pm::(anonymous namespace)::", "").split("(")[0][:90]
        per_step = float(r["TotalDurationNs"]) / man["iters"] / 1e3
        ks[name] = {"calls_per_step": int(r["Calls"]) / man["iters"], "us_per_step": round(per_step, 2), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
        tot += per_step
    summary[f"{w}.{ph}"] = {"kernel_us_per_step": round(tot, 2), "algorithmic_bytes_per_step": man["algorithmic_bytes_per_step"],
                            "frac_of_8TBps_from_kernel_time": round(man["algorithmic_bytes_per_step"] / (tot * 1e-6) / 8e12, 4), "kernels": ks}
json.dump(summary, open(os.path.join(out, "kstats_summary.json"), "w"), indent=1)
for k, v in summary.items():
    print(k, v["kernel_us_per_step"], "us/step  frac", v["frac_of_8TBps_from_kernel_time"])
PY
