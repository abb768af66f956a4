#!/bin/bash
# round 6, visit 8: forward parity on the compact flat-walk kernel + small-request batches, then the bench line
O=gpurun_out/r6_v9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mixed_dims.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_drivers.py tests/test_gpu_blocked.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 900 $O/bench.json
timeout 300 python -m param_amd.compute.pt.driver --steps 50 --warmups 5 --device gpu emb -d A --json > $O/driver_A.txt 2>&1
timeout 300 python -m param_amd.compute.pt.driver --steps 50 --warmups 5 --device gpu emb -d B --json > $O/driver_B.txt 2>&1
grep -v "^{" $O/driver_A.txt | tail -18
