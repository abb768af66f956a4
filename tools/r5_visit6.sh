#!/bin/bash
t=${1:-r5_v6}; mkdir -p gpurun_out/$t
PARAM_AMD_LIB=build/libparam_amd_exp.so timeout 600 python tools/r5_sort_trace.py --workload criteo --requests uniform > gpurun_out/$t/trace_criteo.jsonl 2> gpurun_out/$t/trace_criteo.err; tail -2 gpurun_out/$t/trace_criteo.err
PARAM_AMD_LIB=build/libparam_amd_exp.so timeout 600 python tools/r5_sort_trace.py --workload tables --requests uniform,zipf1.05 > gpurun_out/$t/trace_tables.jsonl 2> gpurun_out/$t/trace_tables.err; tail -2 gpurun_out/$t/trace_tables.err
cat gpurun_out/$t/trace_criteo.jsonl gpurun_out/$t/trace_tables.jsonl
