#!/usr/bin/env python3
"""One bench step from a rocprofv3 --kernel-trace csv: per dispatch start, end and stream (queue)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resolve_blocks" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("q%-3s %-34s start %8.1f end %8.1f dur %8.1f" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-34:], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print("step %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
