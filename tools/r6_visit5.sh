#!/bin/bash
# round 6, visit 5: the flat-walk forward as one resident set of workgroups (flat_compact): parity, then same-box A/B against the old grid
O=gpurun_out/r6_v6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mixed_dims.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not backward and not bwd and not adagrad" > $O/pytest_fwd.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_fwd.txt
for c in 1 0 1 0; do
PARAM_AMD_FLAT_COMPACT=$c timeout 300 python tools/r6_mixed_probe.py --subsets all,all128,narrow,128 >> $O/compact_ab.jsonl 2>> $O/compact_ab.err
done
for w in 1024 1536 2560 3072 4096; do
PARAM_AMD_FLAT_COMPACT=$w timeout 300 python tools/r6_mixed_probe.py --subsets all,all128 >> $O/compact_grid.jsonl 2>> $O/compact_ab.err
done
python - <<'PY'
import json
for f in ("compact_ab", "compact_grid"):
    for l in open(f"gpurun_out/r6_v6/{f}.jsonl"):
        d = json.loads(l)
        if d["hint"] == 1:
            print(f, d["compact"], d["subset"], d["indices"], d["us"], d["alg_frac"])
PY
