#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 `*kernel_stats.csv` (development aid): python tools/kstats.py DIR [rows]."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-72s %6s %10.2f ms %9.1f us" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
