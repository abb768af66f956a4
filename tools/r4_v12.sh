mkdir -p gpurun_out/r4_v12
tools/r4_prof.sh r4_v12/prof_head_sort 0 uniform,zipf1.05 > /dev/null
cat gpurun_out/r4_v12/prof_head_sort/kernel_stats_pm.txt
echo "== r3 tree"
export TMPDIR=/tmp
(cd build/r3_tree && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3 -o p -- python tools/r3_bwd_exp.py --policies 3 --runs seg0 --iters 10 > /dev/null 2>&1)
find /tmp/prof_r3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4_v12/r3_kernel_stats.csv \;
python tools/r4_kstats.py stats gpurun_out/r4_v12/r3_kernel_stats.csv
