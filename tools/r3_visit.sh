#!/bin/bash
# one GPU visit of round 3: tools/r3_visit.sh <tag> <what...>   (what: tests | segtests | exp | bench | prof | pmc ...)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/r3_$tag
mkdir -p "$out"
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    segtests)
      timeout 900 python -m pytest tests/test_gpu_segsort.py tests/test_gpu_sort.py -q -m gpu -x > "$out/segtests.log" 2>&1
      tail -15 "$out/segtests.log" ;;
    cfgtests)
      timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu > "$out/cfgtests.log" 2>&1
      tail -15 "$out/cfgtests.log" ;;
    tests)
      timeout 1500 python -m pytest tests -q -m gpu > "$out/tests.log" 2>&1
      tail -15 "$out/tests.log" ;;
    exp)
      timeout 600 python tools/r3_bwd_exp.py --iters 10 > "$out/bwd_exp.jsonl" 2> "$out/bwd_exp.err"
      cat "$out/bwd_exp.jsonl"; tail -3 "$out/bwd_exp.err" ;;
    exp:*)
      # exp:<runs>:<policies>[:dtype:tables]
      IFS=: read -r _ runs pols dt tb <<< "$what"
      timeout 600 python tools/r3_bwd_exp.py --iters 10 --runs "$runs" --policies "$pols" --dtype "${dt:-fp32}" --tables "${tb:-48}" > "$out/bwd_exp_${runs}_${pols}_${dt:-fp32}.jsonl" 2> "$out/bwd_exp.err"
      python - "$out/bwd_exp_${runs}_${pols}_${dt:-fp32}.jsonl" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    r=json.loads(ln)
    tag=" ".join(f"{k}={r[k]}" for k in ("sort","mode","digit_rot","phases") if k in r)
    print(f'{tag:28s} pol {r["row_policy"]} {r["indices"]:9s} {r["dtype"]} sort {r["sort_ms"]:.3f} apply {r["apply_ms"]:.3f} total {r["total_ms"]:.3f} frac {r["alg_frac_total"]:.3f} apply_frac {r["alg_frac_apply"]:.3f}')
PY
      tail -3 "$out/bwd_exp.err" ;;
    expbf16)
      timeout 600 python tools/r3_bwd_exp.py --iters 10 --dtype bf16 --tables 64 --policies 0 > "$out/bwd_exp_bf16.jsonl" 2> "$out/bwd_exp_bf16.err"
      cat "$out/bwd_exp_bf16.jsonl"; tail -3 "$out/bwd_exp_bf16.err" ;;
    bench)
      timeout 900 python bench.py > "$out/bench_line.json" 2> "$out/bench.err"
      python - "$out/bench_line.json" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",r["value"],"roof",r["roofline"]["frac"],"bwd",{k:r["bwd_scatter_add"].get(k) for k in ("avg_s_sort_plus_apply","avg_s_apply_only","avg_s_sort","alg_frac")})
u=r["bwd_scatter_add"].get("uniform",{}); print("bwd uniform",{k:u.get(k) for k in ("avg_s_sort_plus_apply","avg_s_apply_only","avg_s_sort","frac","apply_only_frac")})
print("fwd_bwd",r["fwd_bwd_step"].get("avg_s"), r["fwd_bwd_step"].get("uniform",{}).get("avg_s")); print("cpu",r.get("cpu_baseline",{}).get("value"))
PY
      ;;
    prof:*)
      # prof:<runs>:<requests>  e.g. prof:seg0:uniform -- kernel trace of one sort flavour on one index distribution
      IFS=: read -r _ runs reqs <<< "$what"
      d=/tmp/r3prof_${runs}_${reqs}
      rm -rf "$d"
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python "$GRAFT_REPO_ROOT/tools/r3_bwd_exp.py" --iters 10 --policies 0 --runs "$runs" --requests "$reqs" > "$d.out" 2> "$d.err")
      f=$(find "$d" -name "*kernel_stats.csv" | head -1)
      if [ -n "$f" ]; then cp "$f" "$out/prof_${runs}_${reqs}_kernel_stats.csv"; echo "== $runs $reqs"; head -22 "$f" | cut -c1-220; else echo "no stats for $runs $reqs"; tail -5 "$d.err"; ls -R "$d" | head; fi
      ;;
    official)
      # the round's record: GPU tests, smoke, the default bench line twice in a row, bf16 / Criteo lines, the N>1 path on a 1-rank
      # RCCL group (26 tables, Criteo), rocprofv3 kernel stats of the headline launches and of the whole default command
      timeout 1500 python -m pytest tests -q -m gpu > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
      timeout 900 python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
      timeout 900 python bench.py > "$out/bench_line_run2.json" 2> "$out/bench_line_run2.err"
      timeout 900 python bench.py --dtype bf16 --no-cpu-baseline > "$out/bench_line_bf16_T64.json" 2> "$out/bench_bf16.err"
      timeout 900 python bench.py --workload criteo --no-cpu-baseline > "$out/bench_line_criteo.json" 2> "$out/bench_criteo.err"
      timeout 900 python bench.py --dist-debug --tables 26 --no-cpu-baseline --steps 20 > "$out/distdebug_26tables.json" 2> "$out/dd26.err"
      timeout 900 python bench.py --dist-debug --workload criteo --no-cpu-baseline --steps 20 > "$out/distdebug_criteo.json" 2> "$out/ddc.err"
      for v in zipf:--only-headline uniform:--only-headline,--alpha,0 full:--no-cpu-baseline; do
        name=${v%%:*}; bargs=${v#*:}
        d=/tmp/r3off_$name; rm -rf "$d"
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 ${bargs//,/ } > "$GRAFT_REPO_ROOT/$out/${name}_under_rocprofv3.json" 2> "$d.err")
        f=$(find "$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${name}_kernel_stats.csv"
      done
      python - "$out" <<'PY'
import json,sys,os
o=sys.argv[1]
def ld(n):
    try: return json.loads(open(os.path.join(o,n)).read().strip().splitlines()[-1])
    except Exception as e: return {"error":str(e)}
for n in ("bench_line.json","bench_line_run2.json","bench_line_bf16_T64.json","bench_line_criteo.json"):
    r=ld(n)
    if "error" in r: print(n,r); continue
    b=r.get("bwd_scatter_add",{}); u=b.get("uniform",{}); c=r.get("cpu_baseline") or {}
    print(f"{n}: value {r['value']/1e9:.2f} G  roof {r['roofline']['frac']:.3f}  bwd zipf {b.get('avg_s_sort_plus_apply',0)*1e3:.3f} ms (sort {b.get('avg_s_sort',0)*1e3:.3f}) alg {b.get('alg_frac',0):.3f} | uniform {u.get('avg_s_sort_plus_apply',0)*1e3:.3f} ms frac {u.get('frac',0):.3f} apply {u.get('apply_only_frac',0):.3f} sort {u.get('avg_s_sort',0)*1e3:.3f} | fwd+bwd {r.get('fwd_bwd_step',{}).get('avg_s',0)*1e3:.3f} / {r.get('fwd_bwd_step',{}).get('uniform',{}).get('avg_s',0)*1e3:.3f} ms | cpu {c.get('value')} cores {c.get('cores')} unstable {c.get('unstable')}")
for n in ("distdebug_26tables.json","distdebug_criteo.json"):
    r=ld(n)
    if "error" in r: print(n,r); continue
    print(n, "tables", r["config"]["tables_total"], "value %.2f G" % (r["value"]/1e9), r["all_to_all"].get("selfcheck",{}).get("a2a_selfcheck"), "fwd_bwd", r.get("fwd_bwd_step",{}).get("avg_s_pipelined"))
PY
      ;;
    fwd:*)
      # fwd:<ENV=val>;<ENV=val>...   the headline forward (Zipf value, uniform roofline fraction, other layout) under an environment setting
      IFS=: read -r _ cfgs <<< "$what"
      for c in ${cfgs//;/ }; do
        env $c timeout 300 python bench.py --no-bwd --no-cpu-baseline --steps 30 > "$out/fwd_${c}.json" 2> "$out/fwd.err"
        python - "$out/fwd_${c}.json" "$c" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); o=r.get("other_layout",{})
print(f"fwd [{sys.argv[2]:30s}] zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.4f} ({r['roofline']['avg_launch_s']*1e6:.1f} us) | other layout zipf {o.get('zipf_lookups_per_s',0)/1e9:.2f} G/s uniform {o.get('uniform_frac',0):.4f}")
PY
      done ;;
    criteofwd:*)
      # criteofwd:<bench args with , for spaces>;<...>   forward of the Criteo workload under bench.py tuning flags
      IFS=: read -r _ cfgs <<< "$what"
      IFS=';' read -ra arr <<< "$cfgs"
      for c in "${arr[@]}"; do
        timeout 300 python bench.py --workload criteo --no-bwd --no-cpu-baseline --steps 30 ${c//,/ } > "$out/criteofwd.json" 2> "$out/criteofwd.err"
        python - "$out/criteofwd.json" "$c" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"criteo fwd [{sys.argv[2]:28s}] zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.3f} ({r['roofline']['avg_launch_s']*1e6:.1f} us)")
PY
      done ;;
    distdebug:*)
      # distdebug:<name>:<bench args with , for spaces>   the N>1 code path on a 1-rank RCCL group
      IFS=: read -r _ name bargs <<< "$what"
      timeout 900 python bench.py --dist-debug --no-cpu-baseline ${bargs//,/ } > "$out/distdebug_${name}.json" 2> "$out/distdebug_${name}.err"
      python - "$out/distdebug_${name}.json" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("workload:", r["config"]["workload"][:160])
print("value", r["value"], "tables", r["config"]["tables_total"], "a2a", {k: r["all_to_all"].get(k) for k in ("avg_s","algbw_GBps","busbw_GBps","rccl_ranks","backend","busbw_over_xgmi_bound")})
print("selfcheck", r["all_to_all"].get("selfcheck"))
print("overlap", r["overlap"]); print("fwd_bwd", {k: r.get("fwd_bwd_step",{}).get(k) for k in ("avg_s_pipelined","avg_s_serial","overlap_eff")})
PY
      tail -2 "$out/distdebug_${name}.err" ;;
    profbench:*)
      # profbench:<name>:<bench args with , for spaces>   rocprofv3 kernel stats of a bench.py command
      IFS=: read -r _ name bargs <<< "$what"
      d=/tmp/r3profbench_$name; rm -rf "$d"
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python "$GRAFT_REPO_ROOT/bench.py" ${bargs//,/ } > "$GRAFT_REPO_ROOT/$out/profbench_${name}.json" 2> "$d.err")
      f=$(find "$d" -name "*kernel_stats.csv" | head -1)
      if [ -n "$f" ]; then cp "$f" "$out/profbench_${name}_kernel_stats.csv"; python - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "pm::" in n and "fill_random" not in n:
        short=re.sub(r"\(.*","",n.replace("void pm::(anonymous namespace)::","").replace("pm::(anonymous namespace)::",""))
        print(f'{short[:72]:72s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
PY
      else echo "no stats"; tail -5 "$d.err"; fi ;;
    criteo:*)
      # criteo:<ENV=val>;<ENV=val>...   forward (+ backward with :bwd suffix on the step name) of the Criteo workload under an environment setting
      IFS=: read -r _ cfgs <<< "$what"
      for c in ${cfgs//;/ }; do
        env $c timeout 300 python bench.py --workload criteo --no-cpu-baseline --steps 30 > "$out/criteo_${c}.json" 2> "$out/criteo.err"
        python - "$out/criteo_${c}.json" "$c" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b=r.get("bwd_scatter_add",{}); u=b.get("uniform",{})
print(f"criteo {sys.argv[2]:32s}: fwd zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.3f} ({r['roofline']['avg_launch_s']*1e6:.1f} us) | bwd zipf {b.get('avg_s_sort_plus_apply',0)*1e6:.0f} us (sort {b.get('avg_s_sort',0)*1e6:.0f}) uniform {u.get('avg_s_sort_plus_apply',0)*1e6:.0f} us frac {u.get('frac',0):.3f} (sort {u.get('avg_s_sort',0)*1e6:.0f})")
PY
      done ;;
    pmc)
      # separate rocprofv3 --pmc passes (no other trace domain), one counter group each; only the CSVs come home
      mkdir -p "$out/pmc"
      i=0
      for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i+1)); d=/tmp/r3pmc_$i; rm -rf "$d"
        (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- \
            python "$GRAFT_REPO_ROOT/tools/pmc_probe.py" --bwd --manifest "$GRAFT_REPO_ROOT/$out/pmc/pmc_manifest.json" > "$d.log" 2>&1)
        mkdir -p "$out/pmc/pmc_$i"
        f=$(find "$d" -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && cp "$f" "$out/pmc/pmc_$i/pmc_counter_collection.csv" || tail -5 "$d.log"
      done
      python tools/parse_pmc.py "$out/pmc" r03_tmp > "$out/pmc_summary.json" 2> "$out/pmc_parse.err"
      python - "$out/pmc_summary.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("calibration",r["calibration"])
for k,v in r["kernels"].items():
    print(k,{x:(round(v[x],4) if isinstance(v[x],float) else v[x]) for x in ("hbm_bytes_per_launch","algorithmic_bytes","hbm_over_algorithmic","l2_hit_rate","fetch_bytes_calibrated","write_bytes_calibrated") if x in v})
PY
      ;;
    *) echo "unknown step $what" ;;
  esac
done
