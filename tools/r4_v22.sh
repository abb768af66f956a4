echo "== HEAD 4096-tile"; timeout 300 python tools/r4_bwd_probe.py --settings 0 2>/dev/null | cut -c1-300
echo "== 2048-tile"; PARAM_AMD_LIB=build/libparam_amd_t2048.so timeout 300 python tools/r4_bwd_probe.py --settings 0 2>/dev/null | cut -c1-300
echo "== 2048-tile criteo"; PARAM_AMD_LIB=build/libparam_amd_t2048.so timeout 300 python tools/r4_bwd_probe.py --settings 0 --workload criteo 2>/dev/null | cut -c1-300
echo "== visible stores: parity tests"; PARAM_AMD_LIB=build/libparam_amd_visible.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_hybrid.py -x -q -m gpu -k "backward or sorted or adagrad or fuzz or random or hybrid or update" 2>&1 | tail -2
PARAM_AMD_LIB=build/libparam_amd_t2048.so timeout 900 python -m pytest tests/test_gpu_segsort.py tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -2
