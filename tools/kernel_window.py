#!/usr/bin/env python3
"""Timeline of the kernels of a rocprofv3 rocpd .db between the LAST occurrence of a kernel whose name contains argv[2]
and the end of the trace: start, gap to the previous end, duration, grid, name; then totals.  Development aid."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start"))
first = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]][-1]
t0, prev_end, tot = rows[first][1], rows[first][1], {}
for r in rows[first:]:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("odx::", "")[:70]
    d = (r[2] - r[1]) / 1e3
    e = tot.setdefault(nm, [0, 0.0])
    e[0] += 1
    e[1] += d
    print("%8.1f us  +%7.1f  dur %7.1f  grid %d %d %d  %s" % ((r[1] - t0) / 1e3, (r[1] - prev_end) / 1e3, d, r[3] // r[6], r[4], r[5], nm))
    prev_end = max(prev_end, r[2])
print("window: %.3f ms" % ((max(r[2] for r in rows[first:]) - t0) / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %4d launches %9.1f us" % (k, v[0], v[1]))
