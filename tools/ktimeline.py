#!/usr/bin/env python3
"""Kernel timeline of the LAST forward of a rocprofv3 --kernel-trace run of tools/prof_forward_batch.py (development aid):
python tools/ktimeline.py DIR [anchor substring, default max_pool] [kernels shown before the anchor, default 3]; the column
after the start time is the stream / queue id."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "max_pool"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]][-1]
t0 = int(rows[i0]["Start_Timestamp"])
tot = 0.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in rows[max(i0 - back, 0):]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%9.1f %8.1f q%-3s %-46s grid=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("odx::", "").replace("void ", "")[:46], r["Grid_Size_X"]))
print("busy %.1f us" % tot)
