#!/usr/bin/env python3
"""Prints per-kernel average durations from a rocprofv3 `--kernel-trace --stats --output-format csv` directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-36s calls %4s avg %9.1f us  %6s%%" % (r["Name"].split("(")[0][-36:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:6]))
