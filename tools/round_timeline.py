#!/usr/bin/env python3
"""One Minibootstrap round (between two consecutive batched factorisation chains) of a rocprofv3 rocpd .db: busy / idle
time of the GPU, the idle gaps longer than 50 us with the kernels around them, and the totals per kernel.

    python tools/round_timeline.py OUT/x_results.db [rounds back from the end, default 3]
"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(c.execute("select name,start,end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "row_sqnorm_f64_batched" in r[0]]
a, b = marks[-back - 1], marks[-back]
seg = rows[a:b]
t0, t1 = seg[0][1], rows[b][1]
print("round: %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("odx::", "")[:60]


busy_end, idle, tot = t0, 0, {}
for i, (n, s, e) in enumerate(seg):
    if s > busy_end:
        gap = s - busy_end
        idle += gap
        if gap > 50e3:
            print("  idle %7.1f us at %8.1f us: after %s, before %s" % (gap / 1e3, (busy_end - t0) / 1e3, short(seg[i - 1][0]), short(n)))
    busy_end = max(busy_end, e)
    v = tot.setdefault(short(n), [0, 0.0])
    v[0] += 1
    v[1] += (e - s) / 1e3
if t1 > busy_end:
    idle += t1 - busy_end
    print("  idle %7.1f us at the end of the round (after %s)" % ((t1 - busy_end) / 1e3, short(seg[-1][0])))
print("idle: %.2f ms" % (idle / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-62s %5d launches %9.1f us" % (k, v[0], v[1]))
