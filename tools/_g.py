import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob("gpurun_out/g%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "gemm_nt_kernel<double" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(tag, k, {n: "%.4g" % (sum(x) / len(x)) for n, x in v.items()}, "launches", len(next(iter(v.values()))))
    ft = glob.glob("gpurun_out/g%s/**/*kernel_trace.csv" % tag, recursive=True)
    if ft:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(ft[0])) if "gemm_nt_kernel<double" in r["Kernel_Name"]]
        print(tag, "durations ms:", ["%.2f" % x for x in d[:12]])
