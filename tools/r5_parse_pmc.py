#!/usr/bin/env python3
"""Summarise tools/r5_pmc.sh's passes: per (workload, phase) the L2 -> fabric bytes per step (FETCH_SIZE + WRITE_SIZE of every
library kernel of the phase, scaled by the calibration pass of the same visit) against the phase's algorithmic bytes, and the same
per kernel.  Counter rows: rocprofv3 counter_collection.csv (Kernel_Name, Counter_Name, Counter_Value), KiB units for both counters.

    python tools/r5_parse_pmc.py gpurun_out/<tag>  > pmc_summary.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src = sys.argv[1]


def rows_of(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return out


def short(name):
    name = name.replace("void ", "").replace("pm::(anonymous namespace)::", "").replace("pm::", "")
    m = re.match(r"([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(src, f"any.calib.{c}.csv")
    man = json.load(open(os.path.join(src, f"any.calib.{c}.manifest.json")))
    rows = rows_of(p)
    if c == "WRITE_SIZE":
        vals = [v for n, v in rows if "fill_random_kernel" in n]
        per = sum(vals) / man["iters"]
    else:
        vals = [v for n, v in rows if "reduce_kernel" in n]
        per = sum(vals) / man["iters"]
    calib[c] = {"counter_KiB_per_8GiB": per, "scale": (8 << 30) / (per * 1024.0) if per > 0 else None}
result = {"calibration": calib, "phases": {}, "units": "bytes; counters are KiB x 1024 x scale",
          "method": "one counter and one (workload, phase) per rocprofv3 run (--pmc X --kernel-trace, nothing else); sums over every pm:: kernel of "
                    "the run except fill_random (table initialisation), divided by the run's step count"}
for man_path in sorted(glob.glob(os.path.join(src, "*.FETCH_SIZE.manifest.json"))):
    base = os.path.basename(man_path)
    wl, ph = base.split(".")[0], base.split(".")[1]
    if ph == "calib":
        continue
    try:
        man = json.loads(open(man_path).read().strip().splitlines()[-1])
    except Exception as exc:
        result["phases"][f"{wl}.{ph}"] = {"error": f"manifest: {exc}"}
        continue
    rec = {"algorithmic_bytes_per_step": man["algorithmic_bytes_per_step"], "steps": man["iters"], "kernels": {}}
    total = 0.0
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(src, f"{wl}.{ph}.{c}.csv")
        if not os.path.exists(p) or calib[c]["scale"] is None:
            rec[c] = None
            continue
        per_k = defaultdict(float)
        calls = defaultdict(int)
        for n, v in rows_of(p):
            if "pm::" not in n or "fill_random" in n:
                continue
            per_k[short(n)] += v
            calls[short(n)] += 1
        tot = sum(per_k.values()) * 1024.0 * calib[c]["scale"] / man["iters"]
        rec[c + "_bytes_per_step"] = tot
        total += tot
        for k, v in per_k.items():
            rec["kernels"].setdefault(k, {})[c + "_bytes_per_step"] = v * 1024.0 * calib[c]["scale"] / man["iters"]
            rec["kernels"][k]["launches_per_step"] = calls[k] / man["iters"]
    rec["fabric_bytes_per_step"] = total
    rec["traffic_over_algorithmic"] = total / man["algorithmic_bytes_per_step"]
    result["phases"][f"{wl}.{ph}"] = rec
print(json.dumps(result, indent=1))
