#!/bin/bash
# round 6, visit 7: in-process A/B of the flat-walk forward's launch shape; small-batch kernel times (row-load batch 2 / 4 / 8) under the kernel trace
O=gpurun_out/r6_v8; mkdir -p $O
timeout 600 python tools/r6_flat_grid_probe.py > $O/flat_grid.jsonl 2> $O/err.txt
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_v8/flat_grid.jsonl"):
    r = json.loads(l)
    d[(r["workload"], r["indices"], r["grid"])].append(r["us"])
for k in sorted(d):
    print(k, d[k])
PY
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/small -o small -- python $R/tools/r6_small_batch_probe.py > $R/$O/small.log 2>&1
cd $R
f=$(find $O/small -name "*kernel_trace.csv" | head -1); python tools/r6_small_batch_parse.py $f | tee $O/small_batch_kernel_us.txt
find $O/small -type f -size +2M -delete
timeout 300 python tools/r6_host_call_probe.py > $O/host_call.json 2>> $O/err.txt; cat $O/host_call.json
