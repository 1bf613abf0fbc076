import csv, glob, sys, collections
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last potrf run: find last potrf_diag sequences; take the last 79 diag kernels' window
idx = [i for i, r in enumerate(rows) if "potrf_diag" in r["Kernel_Name"]]
first = idx[-79]
sel = rows[first:idx[-1] + 3]
t0 = int(sel[0]["Start_Timestamp"])
print("window %.2f ms" % ((int(sel[-1]["End_Timestamp"]) - t0) / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
prev_end = t0
gaps = 0.0
for r in sel[:60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_X"])
    print("%8.1f us  +%6.1f  dur %7.1f  wgs %6d  q%s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, g, r["Queue_Id"], r["Kernel_Name"].split("(")[0][-28:]))
    prev_end = max(prev_end, e)
