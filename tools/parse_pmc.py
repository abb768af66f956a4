#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/pmc_probe.py into profiles/.

Reads gpurun_out/pmc_<i>/pmc_counter_collection.csv (+ pmc_manifest.json), labels the
dispatches of interest by kernel name and launch order, averages each counter per label and
writes:
  profiles/<tag>_pmc_summary.json   per label: counters, algorithmic bytes, calibrated HBM bytes
  profiles/pmc_traffic.json         {workload key: {"hbm_bytes_per_launch": ...}} read by bench.py

Calibration (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB derived from the
L2's fabric-side request counters; their scale is calibrated here on two kernels with KNOWN
traffic (8 GiB written by fill_random_kernel, 8 GiB read by a torch sum) and the same correction
factor is applied to the embedding kernels.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
man = json.load(open(os.path.join(src, "pmc_manifest.json")))
reps = man["reps"]


def label_rows(rows):
    """rows: dispatches of one counter in launch order -> {label: [values]}"""
    out = defaultdict(list)
    n_fwd = 0
    n_fill = 0
    n_main = n_fix = n_uni = n_mark = 0
    for name, val in rows:
        if "fill_random_kernel" in name:
            n_fill += 1
            # the first T fills initialise the tables; the calibration fills are the LAST `reps` large ones
            out["_fill_all"].append(val)
        elif "reduce_kernel" in name and "sum" in name.lower():
            out["_reduce_all"].append(val)
        elif "embbag_fwd_kernel" in name:
            n_fwd += 1
            if n_fwd <= reps + 1:
                if n_fwd > 1:
                    out["fwd_uniform"].append(val)
            elif n_fwd > reps + 2:
                out["fwd_zipf"].append(val)
        elif "embbag_bwd_kernel" in name:
            out["bwd_atomic_uniform"].append(val)
        elif "bwd_unique_kernel" in name:             # round 4, hybrid backward: the bag-major apply (empty under Zipf: no table qualifies)
            n_uni += 1
            out["bwd_unique_uniform" if n_uni <= reps else "bwd_unique_zipf"].append(val)
        elif "hyb_mark_kernel" in name:
            n_mark += 1
            out["bwd_mark_uniform" if n_mark <= reps else "bwd_mark_zipf"].append(val)
        elif "bwd_sorted_main_kernel" in name:        # the sorted apply: `reps` uniform steps (hybrid: the flagged lookups only), then `reps` Zipf
            n_main += 1
            out["bwd_uniform" if n_main <= reps else "bwd_zipf"].append(val)
        elif "bwd_sorted_fixup_kernel" in name:
            n_fix += 1
            out["bwd_fixup_uniform" if n_fix <= reps else "bwd_fixup_zipf"].append(val)
        elif "radix_sort" in name and "onesweep" in name:
            out["bwd_sort_pass_rocprim"].append(val)
        elif "seg_scatter_kernel" in name:
            out["bwd_sort_scatter_pass_seg"].append(val)
        elif "seg_hist_kernel" in name:
            out["bwd_sort_hist_pass_seg"].append(val)
        elif "rs_scatter_kernel" in name:
            out["bwd_sort_scatter_pass"].append(val)
        elif "rs_hist_kernel" in name:
            out["bwd_sort_hist_pass"].append(val)
    if out["_fill_all"]:
        out["calib_write"] = out["_fill_all"][-reps:]
    if out["_reduce_all"]:
        # torch splits the 8 GiB sum into equal 32-bit-indexable pieces (4 x 2 GiB here): the
        # pieces are the dispatches within 1 % of the largest one
        top = max(out["_reduce_all"])
        big = [v for v in out["_reduce_all"] if top > 0 and v > 0.99 * top]
        pieces = max(1, len(big) // reps)
        out["calib_read"] = [v * pieces for v in big]  # per 8 GiB sum
    return {k: v for k, v in out.items() if not k.startswith("_")}


summary = defaultdict(dict)
for path in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
    per_counter = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            per_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    for cname, rows in per_counter.items():
        rows.sort()
        for label, vals in label_rows([(n, v) for _, n, v in rows]).items():
            if vals:
                summary[label][cname] = sum(vals) / len(vals)

calib = {}
if "FETCH_SIZE" in summary.get("calib_read", {}):
    calib["fetch_scale"] = man["calib_bytes"] / (summary["calib_read"]["FETCH_SIZE"] * 1024)
if "WRITE_SIZE" in summary.get("calib_write", {}):
    calib["write_scale"] = man["calib_bytes"] / (summary["calib_write"]["WRITE_SIZE"] * 1024)

res = {"manifest": {k: v for k, v in man.items() if k != "order"}, "calibration": calib, "kernels": {}}
for label, c in summary.items():
    e = dict(c)
    if "FETCH_SIZE" in c:
        e["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
        e["fetch_bytes_calibrated"] = c["FETCH_SIZE"] * 1024 * calib.get("fetch_scale", 1.0)
    if "WRITE_SIZE" in c:
        e["write_bytes_raw"] = c["WRITE_SIZE"] * 1024
        e["write_bytes_calibrated"] = c["WRITE_SIZE"] * 1024 * calib.get("write_scale", 1.0)
    if "fetch_bytes_calibrated" in e and "write_bytes_calibrated" in e:
        e["hbm_bytes_per_launch"] = e["fetch_bytes_calibrated"] + e["write_bytes_calibrated"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if label.startswith("fwd"):
        e["algorithmic_bytes"] = man["alg_bytes_fwd"]
    if label.startswith("bwd") and "alg_bytes_bwd" in man:
        e["algorithmic_bytes"] = man["alg_bytes_bwd"]
    if "hbm_bytes_per_launch" in e and "algorithmic_bytes" in e:
        e["hbm_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes"]
    res["kernels"][label] = e

# round 4: under uniform indices the backward's apply is the bag-major kernel + the sorted apply of the flagged lookups
ks = res["kernels"]
if "bwd_unique_uniform" in ks and "hbm_bytes_per_launch" in ks["bwd_unique_uniform"]:
    parts = [ks[k] for k in ("bwd_unique_uniform", "bwd_uniform", "bwd_fixup_uniform", "bwd_mark_uniform") if "hbm_bytes_per_launch" in ks.get(k, {})]
    tot = sum(p_["hbm_bytes_per_launch"] for p_ in parts)
    ks["bwd_uniform_step"] = {"what": "bag-major apply + sorted apply of the flagged lookups + fix-up + mark kernel (the key sort's small kernels excluded)",
                              "hbm_bytes_per_launch": tot, "fetch_bytes_calibrated": sum(p_.get("fetch_bytes_calibrated", 0) for p_ in parts),
                              "write_bytes_calibrated": sum(p_.get("write_bytes_calibrated", 0) for p_ in parts),
                              "algorithmic_bytes": man.get("alg_bytes_bwd"), "hbm_over_algorithmic": tot / man["alg_bytes_bwd"] if man.get("alg_bytes_bwd") else None}

os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
m = man
for label, alpha in (("fwd_uniform", 0.0), ("fwd_zipf", 1.05)):
    k = res["kernels"].get(label, {})
    if "hbm_bytes_per_launch" in k:
        key = f"T{m['T']}_R{m['R']}_D{m['D']}_B{m['B']}_L{m['L']}_a{alpha}_fp32"
        traffic[key] = {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc_summary.json",
                        "fetch_scale": calib.get("fetch_scale"), "write_scale": calib.get("write_scale")}
json.dump(traffic, open(traffic_path, "w"), indent=1)
print(json.dumps(res, indent=1))
