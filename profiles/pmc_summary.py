#!/usr/bin/env python3
"""Per-kernel mean / max of every counter in a rocprofv3 `--pmc ... --kernel-trace --output-format csv` directory (one row
per dispatch and counter in *_counter_collection.csv). With a second argument K also the mean over the LAST K launches of
each kernel (bench.py's timed steps: the launches before them are warm-up rounds). Prints JSON."""
import collections
import csv
import glob
import json
import sys

last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {}
    for c, v in cs.items():
        e = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)}
        if last:
            e["mean_last_%d" % last] = sum(v[-last:]) / len(v[-last:])
            e["values"] = v                      # per launch, in dispatch order
        out[k][c] = e
json.dump(out, sys.stdout, indent=1)
