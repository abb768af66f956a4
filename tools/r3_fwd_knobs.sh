#!/bin/bash
# forward knobs re-measured with the load batch real (round 3): bags per workgroup, non-temporal row loads, unroll at each
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_fwdknobs; mkdir -p $out
run() {
  timeout 300 python bench.py --no-bwd --no-cpu-baseline --steps 30 "$@" > $out/line.json 2> $out/err.txt
  python - $out/line.json "$*" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"fwd [{sys.argv[2]:44s}] zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.4f} ({r['roofline']['avg_launch_s']*1e6:.1f} us)")
except Exception as e:
    print("failed", sys.argv[2], e)
PY
}
run
run --bags-per-block 16
run --bags-per-block 64
run --nt-loads 1
run --unroll 4 --bags-per-block 64
run --xcd-affine 0
run --dtype bf16
run --dtype bf16 --unroll 4
run --dtype bf16 --unroll 1
run --workload criteo
run --workload criteo --bags-per-block 16
run --workload criteo --bags-per-block 64
