#!/usr/bin/env python3
"""print the numbers of a bench.py JSON line that the round's targets are stated in"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("fwd zipf G lookups/s", round(d["value"] / 1e9, 2), "| uniform frac", round(r["frac"], 4), "| traffic/alg", r.get("traffic_over_algorithmic"))
if "other_layout" in d:
    o = d["other_layout"]
    print("other layout", o["output_layout"], "zipf G", round(o["zipf_lookups_per_s"] / 1e9, 2), "uniform frac", round(o.get("uniform_frac", 0), 4))
if "bwd_scatter_add" in d:
    b = d["bwd_scatter_add"]
    print("bwd zipf ms", round(b["avg_s_sort_plus_apply"] * 1e3, 4), "apply", round(b["avg_s_apply_only"] * 1e3, 4), "whole key sort ms", round(b.get("avg_s_whole_key_sort", 0) * 1e3, 4), b.get("sort"))
    if "uniform" in b:
        u = b["uniform"]
        print("bwd uniform ms", round(u["avg_s_sort_plus_apply"] * 1e3, 4), "frac", round(u["frac"], 4), "apply_only_frac", round(u["apply_only_frac"], 4), "whole key sort ms", round(u.get("avg_s_whole_key_sort", 0) * 1e3, 4), u.get("sort"))
    f = d["fwd_bwd_step"]
    print("fwd+bwd ms zipf", round(f["avg_s"] * 1e3, 4), "uniform", round(f.get("uniform", {}).get("avg_s", 0) * 1e3, 4), "frac", f.get("uniform", {}).get("frac"))
for k in ("bf16_T64", "criteo"):
    if k in d:
        x = d[k]
        if "fwd" not in x:
            print(k, x)
            continue
        bu, bz = x["bwd_scatter_add"]["uniform"], x["bwd_scatter_add"]["zipf"]
        print(k, "fwd zipf G", round(x["fwd"]["zipf_lookups_per_s"] / 1e9, 2), "uniform frac", round(x["fwd"]["uniform_frac"], 4), "| bwd uniform ms",
              round(bu["avg_s_sort_plus_apply"] * 1e3, 4), "frac", round(bu["frac"], 4), "hyb", bu["hybrid_tables"], "| bwd zipf ms",
              round(bz["avg_s_sort_plus_apply"] * 1e3, 4), "| fwd+bwd uniform ms", round(bu["fwd_bwd_step_s"] * 1e3, 4))
c = d.get("cpu_baseline") or {}
print("cpu", {k: c.get(k) for k in ("value", "cores", "unstable", "spread", "repeats_kept", "repeats_dropped_throttled", "value_best_repeat")})
if "all_to_all" in d:
    print("a2a", {k: d["all_to_all"].get(k) for k in ("avg_s", "busbw_GBps", "rccl_ranks", "selfcheck")})
    print("overlap", d.get("overlap"))
    print("fwd_bwd_step", {k: v for k, v in d.get("fwd_bwd_step", {}).items() if k != "what"})
