#!/bin/bash
# round 6, visit 10: the bench line's Criteo blocks with the flat-walk forward's old grid / the compact launch taking turns (same box);
# the five 40 M-row Criteo tables' backward sorted / hybrid
O=gpurun_out/r6_v11; mkdir -p $O
for rep in 1 2; do for c in 0 1; do
PARAM_AMD_FLAT_COMPACT=$c timeout 700 python bench.py --no-cpu-baseline > $O/bench_compact${c}_rep$rep.json 2>> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench_compact${c}_rep$rep.json"))
print("compact", $c, "rep", $rep, {k: {x: round(v * 1e6, 1) for x, v in d[k]["fwd"].items() if "launch_s" in x} for k in ("criteo", "criteo_mixed")})
PY
done; done
timeout 300 python tools/r6_mixed_probe.py --subsets 128 --backward --hybrid 1,2,1,2 > $O/big5_hybrid.jsonl 2> $O/err.txt
grep mixed_bwd $O/big5_hybrid.jsonl | cut -c1-330
