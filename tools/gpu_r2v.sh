#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python bench.py --dist-debug --tables 26 --steps 10 --warmup 3 --no-cpu-baseline --a2a-bitwidth 8 --grad-bitwidth 16 > gpurun_out/r2v_dd26_q8.json 2> gpurun_out/r2v_dd26_q8.err
timeout 600 python bench.py --dist-debug --tables 26 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2v_dd26_fp32.json 2> gpurun_out/r2v_dd26_fp32.err
timeout 600 python bench.py --dist-debug --workload criteo --steps 10 --warmup 3 --no-cpu-baseline --a2a-bitwidth 16 > gpurun_out/r2v_ddc_q16.json 2> gpurun_out/r2v_ddc_q16.err
for f in r2v_dd26_q8 r2v_dd26_fp32 r2v_ddc_q16; do python - <<P
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], json.dumps(d.get("all_to_all")), json.dumps({k:v for k,v in d.get("fwd_bwd_step",{}).items() if k!="what"}))
except Exception as e:
    print("$f", "ERR", e); print(open("gpurun_out/$f.err").read()[-1500:])
P
done
