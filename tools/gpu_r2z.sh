#!/bin/bash
# experiment: 8192-element sort tiles (512 threads) vs 4096 (256 threads)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r2z_rs512.jsonl
for v in "" _rs512; do
  echo "{\"lib\": \"$v\"}" >> gpurun_out/r2z_rs512.jsonl
  PARAM_AMD_LIB=$PWD/param_amd/libparam_amd$v.so timeout 600 python tools/bwd_probe.py --configs "0,1,1,1" >> gpurun_out/r2z_rs512.jsonl 2>> gpurun_out/r2z.err
done
PARAM_AMD_LIB=$PWD/param_amd/libparam_amd_rs512.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sorted_backward or full_size_zipf" 2>&1 | tail -3
python - <<'P'
import json
for ln in open("gpurun_out/r2z_rs512.jsonl"):
    d=json.loads(ln)
    print(d.get("lib") if "lib" in d else (d["indices"], round(d["sort_ms"],4), round(d["apply_ms"],3), round(d["total_ms"],3), round(d["alg_frac_total"],3)))
P
