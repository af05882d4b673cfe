#!/usr/bin/env python3
"""Per step of a rocprofv3 --kernel-trace csv of the bench: the resolve launch's duration, when the stitch ended relative to it, the
insertion, and the step; then, in full, the slowest step."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resolve_blocks" in r["Kernel_Name"]]
worst, worst_i = 0, None
for a, b in zip(idx[:-1], idx[1:]):
    t0 = int(rows[a]["Start_Timestamp"])
    step = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    res = (int(rows[a]["End_Timestamp"]) - t0) / 1e3
    st = [r for r in rows[max(0, a - 40):b] if "k_stitch" in r["Kernel_Name"] and int(r["End_Timestamp"]) > t0 and int(r["Start_Timestamp"]) < int(rows[a]["End_Timestamp"]) + 300000]
    ste = (int(st[0]["End_Timestamp"]) - t0) / 1e3 if st else -1
    ins = [r for r in rows[a:b] if "k_insert_multi<true" in r["Kernel_Name"]]
    ie = (int(ins[0]["End_Timestamp"]) - t0) / 1e3 if ins else -1
    print("step %7.1f  resolve %7.1f  stitch end %7.1f  insert end %7.1f" % (step, res, ste, ie))
    gap = step - ie
    if gap > worst and a > idx[6] and step < 4000:
        worst, worst_i = gap, (a, b)
a, b = worst_i
t0 = int(rows[a]["Start_Timestamp"])
print("--- the step with the longest idle stretch behind its insertion")
for r in rows[max(0, a - 6):b + 8]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("q%-3s %-34s start %8.1f end %8.1f dur %8.1f" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-34:], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
