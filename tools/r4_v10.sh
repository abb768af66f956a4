mkdir -p gpurun_out/r4_v13
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r4_v13/gpu_tests.txt 2>&1; echo tests rc $?; tail -3 gpurun_out/r4_v13/gpu_tests.txt
echo "== r3 tree fp32"
(cd build/r3_tree && timeout 300 python tools/r3_bwd_exp.py --policies 3 --runs seg0 --iters 15 2>/dev/null | cut -c1-300)
echo "== HEAD: hybrid 0 / 1"
timeout 300 python tools/r4_bwd_probe.py --settings 0,1 2>/dev/null | cut -c1-360
