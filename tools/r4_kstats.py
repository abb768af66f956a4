#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs: `r4_kstats.py stats <kernel_stats.csv>` prints this library's kernels (average / min / max us);
`r4_kstats.py trace <kernel_trace.csv> [n]` prints the last n kernel dispatches as a timeline (start, duration, queue, name) so that
overlap between streams can be read off."""
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("void ", "").replace("pm::(anonymous namespace)::", "").replace("pm::", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^(]*>)?", name)
    if not m:
        return name[:70]
    tmpl = (m.group(2) or "").replace("(anonymous namespace)::", "")
    return (m.group(1) + tmpl)[:70]


def main():
    mode, path = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(path)))
    if mode == "stats":
        for r in rows:
            if "pm::" in r["Name"] and "fill_random" not in r["Name"]:
                print(f"{short(r['Name']):72s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  "
                      f"max {float(r['MaxNs']) / 1e3:8.1f}")
    else:
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
        rows = [r for r in rows if "pm::" in r["Kernel_Name"] and "fill_random" not in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[-n:]
        t0 = int(rows[0]["Start_Timestamp"])
        for r in rows:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            print(f"{s / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
