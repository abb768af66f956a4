mkdir -p gpurun_out/r4_v7
echo "== old lib bf16"; PARAM_AMD_LIB=build/ab/lib_cb110fc.so timeout 300 python tools/r4_bwd_probe.py --settings 0 --dtype bf16 --tables 64 --requests uniform 2>gpurun_out/r4_v7/e1.err | cut -c1-330
echo "== new lib bf16 hybrid off"; timeout 300 python tools/r4_bwd_probe.py --settings 0 --dtype bf16 --tables 64 --requests uniform 2>gpurun_out/r4_v7/e2.err | cut -c1-330
echo "== small batches, hybrid off (main kernel on few pairs)"
for b in 128 256 512; do tools/r4_prof.sh r4_v7/prof_b$b 0 uniform --batch $b > /dev/null; grep -h "main_kernel\|lookback\|hist_all" gpurun_out/r4_v7/prof_b$b/kernel_stats_pm.txt; done
echo "== zipf hybrid 2 vs 0 fp32"; timeout 300 python tools/r4_bwd_probe.py --settings 0,2 --requests zipf1.05 2>/dev/null | cut -c1-330
