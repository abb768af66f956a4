cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_shortbags; mkdir -p $out
run() {
  env $1 timeout 300 python bench.py --no-bwd --no-cpu-baseline --steps 30 ${@:2} > $out/line.json 2> $out/err.txt
  python - $out/line.json "$*" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"fwd [{sys.argv[2]:56s}] zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.4f} ({r['roofline']['avg_launch_s']*1e6:.1f} us)")
except Exception as e:
    print("failed", sys.argv[2], e); print(open(sys.argv[1].replace('line.json','err.txt')).read()[-600:])
PY
}
for L in 1 2 4; do
  run PARAM_AMD_FWD_FLAT=1 --pooling $L --batch 65536
  run PARAM_AMD_FWD_FLAT=2 --pooling $L --batch 65536
done
