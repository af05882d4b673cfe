#!/usr/bin/env python3
"""Per-kernel mean / max of every counter in a rocprofv3 `--pmc ... --kernel-trace --output-format csv`
directory (one row per dispatch and counter in *_counter_collection.csv). Prints JSON."""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in cs.items()}
    vg = None
json.dump(out, sys.stdout, indent=1)
