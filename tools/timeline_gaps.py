#!/usr/bin/env python3
"""Where the time of a bench step goes, from a rocprofv3 --kernel-trace CSV (development aid, runs anywhere).

    python tools/timeline_gaps.py gpurun_out/tl/tl_kernel_trace.csv [--from-last N]

Takes the last N main-stream launches of the K_nM build kernel as class boundaries (default: every class after the
first third of the trace, i.e. the timed region) and prints, per class period: wall time, the main stream's busy time
split by kernel family, its idle time (gaps between consecutive main-stream kernels) and what the side streams ran in the
meantime.  "Main stream" = the stream the build kernel runs on."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(odx::)?(\w+)(<[^>]*>)?\s*(\(|$)", name)
    n = name
    if "odx::" in name:
        n = name.split("odx::")[1].split("(")[0]
    elif "rocclr" in name:
        n = name.split("(")[0]
    else:
        n = "torch:" + re.sub(r".*native::", "", name)[:40]
    return n


def main():
    path = sys.argv[1]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"] + "/" + r["Queue_Id"], short(r["Kernel_Name"])))
    rows.sort()
    builds = [r for r in rows if r[3].startswith("gauss_knm_h2")]
    if not builds:
        print("no build kernels in the trace")
        return
    main_stream = builds[-1][2]
    # class periods: from one build start to the next
    starts = [b[0] for b in builds if b[2] == main_stream]
    starts = starts[len(starts) // 3:]
    print("main stream %s, %d class periods analysed" % (main_stream, len(starts) - 1))
    tot = defaultdict(float)
    for i in range(len(starts) - 1):
        a, b = starts[i], starts[i + 1]
        mine = [r for r in rows if r[2] == main_stream and a <= r[0] < b]
        side = [r for r in rows if r[2] != main_stream and r[1] > a and r[0] < b]
        busy = defaultdict(float)
        idle = defaultdict(float)
        prev_end, prev_name = a, "start"
        for s, e, _, n in mine:
            busy[n] += (e - s) / 1e6
            if s > prev_end:
                idle[prev_name + " -> " + n] += (s - prev_end) / 1e6
            prev_end, prev_name = max(prev_end, e), n
        if b > prev_end:
            idle[prev_name + " -> next build"] += (b - prev_end) / 1e6
        sb = defaultdict(float)
        for s, e, _, n in side:
            sb[n] += (min(e, b) - max(s, a)) / 1e6
        wall = (b - a) / 1e6
        tot["wall"] += wall
        for k, v in busy.items():
            tot["busy:" + k] += v
        for k, v in idle.items():
            tot["idle:" + k] += v
        for k, v in sb.items():
            tot["side:" + k] += v
    n = len(starts) - 1
    print("per class, ms (mean over %d):" % n)
    print("  wall %.2f" % (tot["wall"] / n))
    for pref in ("busy:", "idle:", "side:"):
        items = sorted(((v / n, k) for k, v in tot.items() if k.startswith(pref)), reverse=True)
        s = sum(v for v, _ in items)
        print("  %s total %.2f" % (pref, s))
        for v, k in items[:14]:
            print("      %8.3f  %s" % (v, k[len(pref):]))


if __name__ == "__main__":
    main()
