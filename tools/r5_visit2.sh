#!/bin/bash
# tools/r5_visit2.sh: persistent forward, occupancy / tile sweep (same box, one process per unroll)
out=gpurun_out/${1:-r5_v2}; mkdir -p "$out"; export TMPDIR=/tmp
C="classic;2,3,1,4,2;2,3,1,4,3;2,3,1,4,4;2,3,1,4,5;2,8,1,4,4;2,2,2,4,2;2,2,2,4,3;2,2,2,4,4;2,2,2,4,5;2,3,2,4,3;2,4,2,4,3;2,2,4,4,1;2,2,4,4,2;2,2,4,4,3;2,3,4,4,2;2,3,1,7,1;2,3,1,7,2;2,3,1,7,3;2,3,2,7,1;2,3,2,7,2;2,2,4,7,1;2,2,4,7,2"
for u in 0 4; do
  timeout 600 python tools/r5_fwd_ab.py --rounds 1 --layouts tbd --unroll $u --configs "$C" > "$out/sweep_u$u.jsonl" 2> "$out/sweep_u$u.err"; echo "== unroll $u rc=$?"
  python - "$out/sweep_u$u.jsonl" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
print("bit-identity failures:", [r for r in rows if r.get("bit_identical_to_classic") is False])
agg = collections.defaultdict(dict)
for r in rows:
    if "avg_launch_us" in r: agg[r["config"]][r["indices"]] = r["avg_launch_us"]
for k in agg: print(k, agg[k])
PY
done
