#!/usr/bin/env python3
"""A rocprofv3 --kernel-trace csv of one `mbgc-hip c` run: how busy the device was between the first and the last match-finding
launch (union of the kernels' intervals over that span), the kernels' totals, and one round from the middle of the run.
usage: timeline_tool.py <rocprof output dir> [round index | -1 = the middle one]"""
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resolve_blocks" in r["Kernel_Name"]]
a, b = int(rows[idx[0]]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows[idx[0]:])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[idx[0]:])
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for r in rows[idx[0]:]:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1]
    d = tot.setdefault(n, [0, 0])
    d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[1] += 1
print(json.dumps({"span_ms": round((b - a) / 1e6, 2), "device_busy_ms": round(busy / 1e6, 2), "busy_fraction": round(busy / (b - a), 3),
                  "match_finding_launches": len(idx),
                  "kernel_ms": {k: [round(v[0] / 1e6, 2), v[1]] for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]}}))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
k = len(idx) // 2 if which < 0 else which
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("q%-3s %-34s start %8.1f end %8.1f dur %8.1f" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-34:], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print("round %d: %.1f us" % (k, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
