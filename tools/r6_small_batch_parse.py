#!/usr/bin/env python3
"""median kernel duration per (kernel instantiation, grid) from a rocprofv3 --kernel-trace CSV (tools/r6_small_batch_probe.py)"""
import csv
import re
import statistics
import sys
from collections import defaultdict

d = defaultdict(list)
order = []
for row in csv.DictReader(open(sys.argv[1])):
    name = row.get("Kernel_Name") or row.get("kernel_name")
    if "embbag_fwd" not in name:
        continue
    m = re.search(r"embbag_fwd\w*<([^>]*)>", name)
    grid = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0) // int(row.get("Workgroup_Size_X") or row.get("Workgroup_Size") or 256)
    key = (m.group(1) if m else name[:60], grid)
    if key not in d:
        order.append(key)
    d[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for key in order:
    v = d[key]
    print(f"{key[0]:40s} workgroups {key[1]:6d}  launches {len(v):4d}  median {statistics.median(v):8.2f} us  min {min(v):8.2f}")
