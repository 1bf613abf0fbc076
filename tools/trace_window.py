#!/usr/bin/env python3
"""Kernels of a rocprofv3 kernel-trace CSV from the LAST launch whose name contains argv[2] to the end of the trace (or to
the last launch containing argv[3]): per-kernel launches / total / average, the window's span and its idle time.
Usage: python tools/trace_window.py <x_kernel_trace.csv> <first-kernel substring> [<last-kernel substring>]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]][-1]
last = len(rows) - 1
if len(sys.argv) > 3:
    last = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]][-1]
win = rows[first:last + 1]
t0, t1 = int(win[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in win)
tot = {}
busy, cur_end = 0, t0
for r in win:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("odx::", "")[:90]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    e = tot.setdefault(nm, [0, 0])
    e[0] += 1
    e[1] += b - a
    if b > cur_end:
        busy += b - max(a, cur_end)
        cur_end = b
print("window: %.3f ms, %d launches, GPU busy %.3f ms (idle %.3f)" % ((t1 - t0) / 1e6, len(win), busy / 1e6, (t1 - t0 - busy) / 1e6))
print("| kernel | launches | total us | avg us |")
print("|---|---|---|---|")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("| `%s` | %d | %.1f | %.1f |" % (k, v[0], v[1] / 1e3, v[1] / 1e3 / v[0]))
