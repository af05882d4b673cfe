#!/usr/bin/env python3
"""Per-kernel average duration over bench.py's timed region in a rocprofv3 --kernel-trace csv: the region is the last K rounds,
and a round begins with a match-finding launch (`k_resolve_blocks4`, either instantiation) — every launch that starts at or after
the K-th last of those is counted. (The first warm-up round of a collection matches most of its targets again in small units,
DESIGN §4b: the whole-process --stats average mixes those launches in, and so did "the last K launches of each kernel".)
Usage: timed_stats.py DIR K"""
import collections
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if "k_resolve_blocks" in r["Kernel_Name"]]
t0 = starts[-k] if len(starts) >= k else 0
d, every = collections.defaultdict(list), collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    every[n].append(us)
    if int(r["Start_Timestamp"]) >= t0:
        d[n].append(us)
out = {n: {"launches_total": len(every[n]), "launches_timed": len(v), "avg_us_timed": round(sum(v) / len(v), 1), "avg_us_all": round(sum(every[n]) / len(every[n]), 1)}
       for n, v in d.items() if n.startswith(("swk::", "void swk::"))}
res = [v for n, vs in d.items() if "k_resolve_blocks" in n for v in vs]
out["_timed_region"] = {"rounds": k, "match_finding_launches": len(res), "k_resolve_blocks4_avg_us": round(sum(res) / max(1, len(res)), 1)}
json.dump(out, sys.stdout, indent=1)
