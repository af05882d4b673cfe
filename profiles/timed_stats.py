#!/usr/bin/env python3
"""Per-kernel average duration over the LAST k launches of each kernel in a rocprofv3 --kernel-trace csv:
bench.py's timed region is its last `--steps` rounds (the warm-up round also launches smaller retries of
dissimilar contigs, which the whole-process --stats average mixes in). Usage: timed_stats.py DIR K"""
import collections
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2])
d = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {n: {"launches_total": len(v), "avg_us_last_%d" % k: round(sum(v[-k:]) / len(v[-k:]), 1), "avg_us_all": round(sum(v) / len(v), 1)}
       for n, v in d.items() if n.startswith(("swk::", "void swk::"))}
json.dump(out, sys.stdout, indent=1)
