#!/bin/bash
# tools/r5_visit7.sh: the sorted apply on the hybrid path's left-overs (225 K pairs): main kernel time by apply tile
t=${1:-r5_v7}
for tile in 256 512 1024; do
  TL_ROWS=5 PROBE_SETTINGS=-1 PROBE_ARGS="--tables 48" bash tools/r4_timeline.sh ${t}_t$tile uniform PARAM_AMD_BWD_TILE=$tile | grep -E "bwd_sorted_main|bwd_sorted_fixup|total_ms" | cut -c1-200
done
