cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_v45; mkdir -p $out
for u in 1 2 3 4 6 8; do
  timeout 300 python bench.py --no-bwd --no-cpu-baseline --steps 30 --unroll $u > $out/fwd_u$u.json 2> $out/fwd.err
  python - $out/fwd_u$u.json $u <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); o=r.get("other_layout",{})
print(f"fwd unroll {sys.argv[2]}: zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.4f} ({r['roofline']['avg_launch_s']*1e6:.1f} us)")
PY
done
for u in 1 2 4; do
  timeout 300 python bench.py --workload criteo --no-bwd --no-cpu-baseline --steps 30 --unroll $u > $out/cfwd_u$u.json 2> $out/fwd.err
  python - $out/cfwd_u$u.json $u <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"criteo fwd unroll {sys.argv[2]}: zipf {r['value']/1e9:.2f} G/s ({r['roofline']['zipf']['avg_launch_s']*1e6:.1f} us) uniform frac {r['roofline']['frac']:.3f} ({r['roofline']['avg_launch_s']*1e6:.1f} us)")
PY
done
