#!/bin/bash
# tools/r4_sort_timeline.sh <tag> [lib ...]: tools/r4_sort_probe.py under rocprofv3 --kernel-trace, once per library given
# (PARAM_AMD_LIB; "default" = the in-tree one): the sort's kernels, last sort of each request.
tag=${1:-r4_st}; shift; libs=${@:-default}
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
for lib in $libs; do
  name=$(basename "$lib" .so)
  if [ "$lib" = default ]; then unset PARAM_AMD_LIB; else export PARAM_AMD_LIB=$GRAFT_REPO_ROOT/$lib; fi
  for rq in uniform zipf1.05; do
    d=/tmp/st_${tag}_${name}_$rq; rm -rf "$d"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$d" -o st -- python "$GRAFT_REPO_ROOT/tools/r4_sort_probe.py" --requests $rq > "$GRAFT_REPO_ROOT/$out/${name}_$rq.log" 2> "$d.err")
    f=$(find "$d" -name "*kernel_trace.csv" | head -1)
    echo "== $name $rq $(tail -1 "$out/${name}_$rq.log")"
    python tools/r4_kstats.py trace "$f" 7 | tee "$out/${name}_$rq.timeline.txt"
  done
done
