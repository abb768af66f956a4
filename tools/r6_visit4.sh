#!/bin/bash
# round 6, visit 4: mixed-dim parity, where the mixed forward's time goes, the reference's driver shapes, host time per call
O=gpurun_out/r6_v5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mixed_dims.py -m gpu -x -q > $O/pytest_mixed.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_mixed.txt
timeout 300 python tools/r6_mixed_probe.py --backward > $O/mixed_probe.jsonl 2> $O/mixed_probe.err; echo "probe rc=$?"
PARAM_AMD_FLAT_TARGET=512 timeout 300 python tools/r6_mixed_probe.py --subsets all,narrow > $O/mixed_probe_t512.jsonl 2>> $O/mixed_probe.err
PARAM_AMD_FLAT_TARGET=128 timeout 300 python tools/r6_mixed_probe.py --subsets all,narrow > $O/mixed_probe_t128.jsonl 2>> $O/mixed_probe.err
cat $O/mixed_probe*.jsonl | cut -c1-260
timeout 300 python tools/r6_host_call_probe.py > $O/host_call.json 2> $O/host_call.err; cat $O/host_call.json
timeout 300 python -m param_amd.compute.pt.driver --steps 50 --warmups 5 --device gpu emb -d A --json > $O/driver_A.txt 2>&1
grep -v "^{" $O/driver_A.txt | head -30
