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
    *) echo "unknown step $what" ;;
  esac
done
