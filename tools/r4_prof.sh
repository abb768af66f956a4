#!/bin/bash
# tools/r4_prof.sh <tag> <settings> <requests> [extra args]: rocprofv3 kernel stats + trace of tools/r4_bwd_probe.py for one setting
tag=$1; settings=$2; reqs=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $GRAFT_REPO_ROOT/tools/r4_bwd_probe.py --iters 10 --settings "$settings" --requests "$reqs" "$@" > $out/probe.jsonl 2> $out/probe.err)
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find /tmp/prof_$tag -name "*kernel_trace.csv" -exec cp {} /tmp/prof_$tag/trace.csv \;
python $GRAFT_REPO_ROOT/tools/r4_kstats.py stats $out/kernel_stats.csv > $out/kernel_stats_pm.txt
python $GRAFT_REPO_ROOT/tools/r4_kstats.py trace /tmp/prof_$tag/trace.csv 45 > $out/timeline_tail.txt
cat $out/probe.jsonl; cat $out/kernel_stats_pm.txt; cat $out/timeline_tail.txt
