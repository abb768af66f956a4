mkdir -p gpurun_out/r4_v8
echo "== r3 tree (cb110fc): bf16 T=64 and fp32 T=48, seg0 sort, policy 3"
(cd build/r3_tree && timeout 300 python tools/r3_bwd_exp.py --dtype bf16 --tables 64 --policies 3 --runs seg0 --iters 15 2>/dev/null | cut -c1-300)
(cd build/r3_tree && timeout 300 python tools/r3_bwd_exp.py --policies 3 --runs seg0 --iters 15 2>/dev/null | cut -c1-300)
echo "== HEAD: hybrid 0 / 1 (steady state: hint)"
timeout 300 python tools/r4_bwd_probe.py --settings 0,1 --dtype bf16 --tables 64 2>/dev/null | cut -c1-360
timeout 300 python tools/r4_bwd_probe.py --settings 0,1 2>/dev/null | cut -c1-360
timeout 600 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -3
