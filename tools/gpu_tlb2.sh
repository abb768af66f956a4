#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=40
for c in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $REPO/tools/pmc_tlb_probe2.py > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($c) rc=$?"
done
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmc_4[1-4]/pmc_counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "embbag_fwd_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for c, rows in per.items():
        rows.sort()
        parts = [rows[1:4], rows[5:8], rows[9:12]]
        s = f"{c:40s}"
        for name, p in zip(("rowstores", "burst", "burst_confined"), parts):
            v = sum(x for _, x, _ in p) / max(1, len(p)); t = sum(x for _, _, x in p) / max(1, len(p))
            s += f" | {name} {v:16.1f} ({t/1e3:7.1f} us)"
        print(s)
PY
