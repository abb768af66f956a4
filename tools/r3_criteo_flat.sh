#!/bin/bash
# Criteo forward: flat-walk kernel (default) vs the 8-bag tiles it replaces (PARAM_AMD_FWD_FLAT=0); tile target / cap sweep
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_criteo_flat; mkdir -p $out
run() {
  env $1 timeout 300 python bench.py --workload criteo --no-bwd --no-cpu-baseline --steps 30 > $out/line.json 2> $out/err.txt
  python - $out/line.json "$*" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"criteo fwd [{sys.argv[2]:52s}] zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.4f} ({r['roofline']['avg_launch_s']*1e6:.1f} us)")
except Exception as e:
    print("failed", sys.argv[2], e)
PY
}
run PARAM_AMD_FWD_FLAT=0
run "PARAM_AMD_FWD_FLAT=1"
for t in 128 256 1024; do run "PARAM_AMD_FLAT_TARGET=$t"; done
for b in 16 32 128; do run "PARAM_AMD_FLAT_BAGS=$b"; done
run "PARAM_AMD_FLAT_TARGET=256 PARAM_AMD_FLAT_BAGS=32"
run "PARAM_AMD_FLAT_TARGET=1024 PARAM_AMD_FLAT_BAGS=128"
