#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=20
for c in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $REPO/tools/pmc_tlb_probe.py > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($c) rc=$?"
done
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmc_2[1-5]/pmc_counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "embbag_fwd_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for c, rows in per.items():
        rows.sort()
        full = rows[1:4]; conf = rows[5:8]
        f = sum(v for _, v, _ in full) / max(1, len(full)); g = sum(v for _, v, _ in conf) / max(1, len(conf))
        tf = sum(t for _, _, t in full) / max(1, len(full)); tg = sum(t for _, _, t in conf) / max(1, len(conf))
        print(f"{c:48s} full {f:16.1f} ({tf/1e3:7.1f} us)   confined {g:16.1f} ({tg/1e3:7.1f} us)")
PY
