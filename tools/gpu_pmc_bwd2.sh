#!/bin/bash
# PMC study of the sorted apply kernel (bwd_sorted_main_kernel): request latencies, stall and occupancy counters, one
# rocprofv3 run per counter group (tools/pmc_probe.py --bwd: 3 uniform launches, then 3 Zipf launches)
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=60
for c in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_32B_sum" \
         "TCC_REQ_sum TCC_TAG_STALL_sum" "TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum" "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcb_$i -o pmc -- python $REPO/tools/pmc_probe.py --bwd --reps 3 > $OUT/pmcb_$i.log 2>&1
  echo "pass $i ($c) rc=$?"
done
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/pmcb_*/pmc_counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "bwd_sorted_main_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for c, rows in per.items():
        rows.sort()
        h = len(rows) // 2
        s = f"{c:36s}"
        for name, p in (("uniform", rows[:h]), ("zipf", rows[h:])):
            v = sum(x for _, x, _ in p) / max(1, len(p)); t = sum(x for _, _, x in p) / max(1, len(p))
            s += f" | {name} {v:18.1f} ({t/1e3:7.1f} us, n={len(p)})"
        print(s)
PY
