#!/bin/bash
# tools/r4_timeline.sh <tag> [requests] [KEY=VAL ...]: kernel timeline (rocprofv3 --kernel-trace) of the last backward steps of
# tools/r4_bwd_probe.py: where the time between the kernels of one sort + apply goes.  Extra KEY=VAL pairs are exported first.
tag=${1:-r4_tl}; reqs=${2:-uniform,zipf1.05}; shift; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
for rq in ${reqs//,/ }; do
  d=/tmp/tl_${tag}_$rq; rm -rf "$d"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$d" -o tl -- python "$GRAFT_REPO_ROOT/tools/r4_bwd_probe.py" --settings ${PROBE_SETTINGS:-1} --requests $rq --iters 4 --layout tbd ${PROBE_ARGS} > "$GRAFT_REPO_ROOT/$out/$rq.log" 2> "$d.err")
  f=$(find "$d" -name "*kernel_trace.csv" | head -1)
  python tools/r4_kstats.py trace "$f" ${TL_ROWS:-14} > "$out/$rq.timeline.txt"
  echo "== $tag $rq $*"; tail -1 "$out/$rq.log" | cut -c1-330; cat "$out/$rq.timeline.txt"
done
