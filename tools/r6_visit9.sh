#!/bin/bash
# round 6, visit 9: the full GPU suite on the product + alternates split, the flat-walk grid A/B again (prefix in dynamic LDS), the bench line
O=gpurun_out/r6_v10; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
timeout 600 python tools/r6_flat_grid_probe.py --grids 0,1,2048,4096,8192 > $O/flat_grid.jsonl 2> $O/err.txt
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_v10/flat_grid.jsonl"):
    r = json.loads(l)
    d[(r["workload"], r["indices"], r["grid"])].append(r["us"])
for k in sorted(d):
    print(k, d[k])
PY
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 700 $O/bench.json
