#!/bin/bash
# round 2, visit s: row quantisers -- parity tests, quantised sweep on a 1-rank RCCL group, throughput probe
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rowquant.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r2s_rowquant_tests.log
timeout 600 python -m pytest tests/test_gpu_drivers.py -q -m gpu -x -k "quantised" 2>&1 | tail -15 > gpurun_out/r2s_sweep_test.log
timeout 300 python tools/rowquant_probe.py > gpurun_out/r2s_rowquant_probe.jsonl 2> gpurun_out/r2s_rowquant_probe.err
timeout 300 python -m param_amd.comms.pt.comms --master-ip 127.0.0.1 --b 1M --e 256M --f 4 --n 20 --w 5 --z 1 --collective all_to_allv --device rocm --bitwidth 8 --quant-a2a-embedding-dim 128 > gpurun_out/r2s_sweep_bw8.log 2>&1
cat gpurun_out/r2s_rowquant_tests.log gpurun_out/r2s_sweep_test.log gpurun_out/r2s_rowquant_probe.jsonl gpurun_out/r2s_sweep_bw8.log
tail -5 gpurun_out/r2s_rowquant_probe.err
