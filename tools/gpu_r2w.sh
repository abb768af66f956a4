#!/bin/bash
# round 2, late visit: whole GPU suite, smoke, the default bench line twice (refreshes profiles/r02_bench_line*.json)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2w_pytest.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r2w_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2w_bench_line.json 2> gpurun_out/r2w_bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2w_bench_line_run2.json 2> gpurun_out/r2w_bench_line_run2.err
grep -E "passed|failed|error" gpurun_out/r2w_pytest.log | tail -3; cat gpurun_out/r2w_smoke.log
for f in r2w_bench_line r2w_bench_line_run2; do python - <<P
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"]/1e9,2), d["roofline"]["frac"], {k:(round(v["avg_s_sort_plus_apply"]*1e3,3), round(v["uniform"]["avg_s_sort_plus_apply"]*1e3,3)) for k,v in d.items() if k=="bwd_scatter_add"}, d["cpu_baseline"]["value"])
P
done
