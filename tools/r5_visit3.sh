#!/bin/bash
# tools/r5_visit3.sh: kernel timelines (rocprofv3 --kernel-trace) of the backward for the three uniform-index workloads the review names
t=${1:-r5_v3}
TL_ROWS=16 PROBE_SETTINGS=-1 PROBE_ARGS="--workload criteo --batch 8192" bash tools/r4_timeline.sh ${t}_criteo uniform,zipf1.05
TL_ROWS=16 PROBE_SETTINGS=-1 PROBE_ARGS="--tables 48" bash tools/r4_timeline.sh ${t}_fp32 uniform
TL_ROWS=16 PROBE_SETTINGS=-1 PROBE_ARGS="--tables 64 --dtype bf16" bash tools/r4_timeline.sh ${t}_bf16 uniform
