#!/usr/bin/env python3
"""One bench step from a rocprofv3 --kernel-trace csv: per dispatch start, duration and the idle gap before it."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_probe" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = prev = int(rows[i0]["Start_Timestamp"])
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s start %8.1f us dur %8.1f gap %7.1f" % (r["Kernel_Name"].split("(")[0][-34:], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    busy += e - s
    prev = e
tot = int(rows[i1]["Start_Timestamp"]) - t0
print("step %.1f us, kernels busy %.1f us, idle %.1f us" % (tot / 1e3, busy / 1e3, (tot - busy) / 1e3))
