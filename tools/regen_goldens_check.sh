#!/bin/bash
# tools/regen_goldens_check.sh: every tests/golden/gen_*.py re-run in a scratch copy of the repo (build container only: the
# generators import /root/reference), the regenerated fixtures compared with the committed ones.  Prints one line per fixture;
# exit code 1 if any differs.  (gen_adagrad_fbgemm.py needs fbgemm_gpu, absent here: skipped.)
set -u
src=$(cd "$(dirname "$0")/.." && pwd)
work=$(mktemp -d /tmp/regen.XXXXXX)
mkdir -p "$work/repo"
(cd "$src" && tar cf - --exclude=.git --exclude=gpurun_out --exclude=build --exclude='__pycache__' .) | (cd "$work/repo" && tar xf -)
cd "$work/repo"
rc=0
for g in tests/golden/gen_*.py; do
  case "$g" in *gen_adagrad_fbgemm.py) echo "skip  $g (needs fbgemm_gpu)"; continue;; esac
  if ! timeout 1200 python "$g" > "$work/$(basename "$g").log" 2>&1; then echo "FAIL  $g (see $work/$(basename "$g").log)"; rc=1; fi
done
python - "$src" "$work/repo" <<'PY'
import json, os, sys
import numpy as np
src, new = sys.argv[1], sys.argv[2]
bad = 0
for root, _, files in os.walk(os.path.join(src, "tests", "golden")):
    for f in sorted(files):
        if f.endswith(".py") or "__pycache__" in root:
            continue
        a = os.path.join(root, f)
        b = os.path.join(new, os.path.relpath(a, src))
        if not os.path.exists(b):
            print("MISSING", os.path.relpath(a, src)); bad += 1; continue
        if f.endswith(".npz"):
            x, y = np.load(a), np.load(b)
            same = sorted(x.files) == sorted(y.files) and all(np.array_equal(x[k], y[k], equal_nan=x[k].dtype.kind == "f") for k in x.files)
        else:
            same = open(a, "rb").read() == open(b, "rb").read()
        print("same " if same else "DIFF ", os.path.relpath(a, src))
        bad += 0 if same else 1
sys.exit(1 if bad else 0)
PY
[ $? -ne 0 ] && rc=1
echo "scratch copy: $work"
exit $rc
