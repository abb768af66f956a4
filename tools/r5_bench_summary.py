#!/usr/bin/env python3
"""prints the numbers of a bench line the round's targets are stated in"""
import json
import sys

r = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
ro = r["roofline"]
print("value G lookups/s", round(r["value"] / 1e9, 3), "| roofline frac", round(ro["frac"], 4) if ro.get("frac") else None, "launches", ro.get("launches_timed"),
      "traffic/alg", round(ro.get("traffic_over_algorithmic", 0), 4))
if "other_layout" in r:
    o = r["other_layout"]
    print("other layout", o["output_layout"], "zipf G", round(o["zipf_lookups_per_s"] / 1e9, 3), "uniform frac", round(o.get("uniform_frac", 0), 4))
if "bwd_scatter_add" in r:
    b = r["bwd_scatter_add"]
    print("bwd zipf ms", round(b["avg_s_sort_plus_apply"] * 1e3, 4), "alg_frac", round(b["alg_frac"], 4), "whole key sort ms", round(b["avg_s_whole_key_sort"] * 1e3, 4), b.get("sort"))
    if "uniform" in b:
        u = b["uniform"]
        print("bwd uniform ms", round(u["avg_s_sort_plus_apply"] * 1e3, 4), "frac", round(u["frac"], 4), "traffic/alg", round(u.get("traffic_over_algorithmic", 0), 4),
              "whole key sort ms", round(u["avg_s_whole_key_sort"] * 1e3, 4), u.get("sort"))
    f = r["fwd_bwd_step"]
    print("fwd+bwd zipf ms", round(f["avg_s"] * 1e3, 4), "uniform ms", round(f.get("uniform", {}).get("avg_s", 0) * 1e3, 4), "frac", round(f.get("uniform", {}).get("frac", 0), 4))
for k in ("bf16_T64", "criteo"):
    if k in r and "fwd" in r[k]:
        x = r[k]
        print(k, "fwd zipf G", round(x["fwd"]["zipf_lookups_per_s"] / 1e9, 2), "uniform frac", round(x["fwd"]["uniform_frac"], 4),
              "| bwd uniform ms", round(x["bwd_scatter_add"]["uniform"]["avg_s_sort_plus_apply"] * 1e3, 4), "frac", round(x["bwd_scatter_add"]["uniform"]["frac"], 4),
              "zipf ms", round(x["bwd_scatter_add"]["zipf"]["avg_s_sort_plus_apply"] * 1e3, 4))
    elif k in r:
        print(k, r[k])
c = r.get("cpu_baseline") or {}
print("cpu_baseline G", round((c.get("value") or 0) / 1e9, 3), "cores", c.get("cores"), "mode", c.get("value_mode"), "unstable", c.get("unstable"),
      {k: round(v / 1e9, 3) for k, v in (c.get("modes_lookups_per_s") or {}).items()})
