#!/bin/bash
# round 6, visit 6: grid of the compact flat-walk forward (workgroups walking the tile order) and its tile target, Criteo tables (all 128 / mixed dims)
O=gpurun_out/r6_v7; mkdir -p $O
for rep in 1 2; do
for w in 512 1024 2048 3072 4096 6144 8192 16384 0; do
PARAM_AMD_FLAT_COMPACT=$w timeout 300 python tools/r6_mixed_probe.py --subsets all,all128 >> $O/compact_grid.jsonl 2>> $O/err.txt
done
done
for tg in 128 192 384 512; do
for w in 1024 4096; do
PARAM_AMD_FLAT_TARGET=$tg PARAM_AMD_FLAT_COMPACT=$w timeout 300 python tools/r6_mixed_probe.py --subsets all,all128 >> $O/compact_target.jsonl 2>> $O/err.txt
done
done
python - <<'PY'
import json
for f in ("compact_grid", "compact_target"):
    for l in open(f"gpurun_out/r6_v7/{f}.jsonl"):
        d = json.loads(l)
        if d["hint"] == 1:
            print(f, "grid", d["compact"], "target", d["flat_target"], d["subset"], d["indices"], d["us"], d["alg_frac"])
PY
