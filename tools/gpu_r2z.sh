#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2z_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r2z_pytest.log | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --workload criteo --no-cpu-baseline > gpurun_out/r2z_bench_criteo.json 2> gpurun_out/r2z_bench_criteo.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r2z_bench_criteo.json").read().strip().splitlines()[-1])
print(d["value"]/1e9, d["roofline"]["frac"], d.get("bwd_scatter_add",{}).get("avg_s_sort_plus_apply"))
P
