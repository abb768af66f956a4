#!/usr/bin/env python3
"""Round 6: the reference driver's published shapes (train/compute/pt/dataset.py:56-82; README.md:75-78) on the final tree, with two
columns the reference's table does not have: the algorithmic-bytes fraction of the 8 TB/s HBM peak (per lookup D*e + 8 read, per
bag 8 read + D*4 written) and the KERNEL's own duration from a rocprofv3 --kernel-trace of the same command (the step time of the small
batches is the host's issue rate, not the kernel).

usage: r6_driver_table.py <driver stdout with --json> <kernel_trace.csv of the same command> <warmups> <steps>"""
import csv
import json
import statistics
import sys

out_txt, trace, warmups, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = [json.loads(l) for l in open(out_txt) if l.startswith("{")]
k = []
for r in csv.DictReader(open(trace)):
    name = r.get("Kernel_Name") or r.get("kernel_name")
    if "embbag_fwd" in name:
        k.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
k.sort()
per = warmups + steps
assert len(k) == per * len(rows), (len(k), per, len(rows))
print("-" * 118)
print("    Features    embdim    nnz     batch      Time(s)/step   Data(MB)   BW(GB/s)   alg GB/s  frac of 8 TB/s   kernel us   step us")
print("-" * 118)
for i, r in enumerate(rows):
    ku = statistics.median(d for _, d in k[i * per + warmups:(i + 1) * per])
    mb = r["batch"] * r["nnz"] * r["embdim"] * 4 / 1e6
    print("{:10},  {:6},  {:6},  {:8},    {:10.6f}, {:10.1f},  {:8.3f}   {:8.1f}      {:6.3f}       {:8.2f}  {:8.2f}".format(
        r["features"], r["embdim"], r["nnz"], r["batch"], r["s_per_step"], mb, mb / r["s_per_step"] / 1e3, r["algorithmic_GBps"],
        r["hbm_roofline_frac"], ku, r["s_per_step"] * 1e6))
