#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the markdown committed under profiles/."""
import collections
import csv
import glob
import json
import os
import statistics
import sys


def robust(x):
    """Per-counter median over the profiled dispatches.  A plain mean let one outlier sample (a GRBM_GUI_ACTIVE read of
    3e10 on one dispatch in round 1) print a "43.76 GHz" clock into the committed summary."""
    return statistics.median(x)


d = sys.argv[1]
print("# rocprofv3 summary (%s)\n" % os.path.basename(d))
try:
    b = json.loads(open(os.path.join(d, "bench_n1.json")).read().strip().splitlines()[-1])
    print("bench.py (un-profiled): value %.1f %s, %.1f ms/step, roofline %s\n" % (b["value"], b["unit"], b["ms_per_step"], json.dumps(b["roofline"])))
    for key in ("roofline_hbm", "roofline_mfma", "roofline_second_family"):
        if key in b:
            print("%s: %s\n" % (key, json.dumps(b[key])))
    print("phases (ms/step, rank 0): %s\n" % json.dumps(b["phases_ms_per_step_rank0"]))
    print("cpu_baseline: %s\n" % json.dumps(b.get("cpu_baseline")))
except Exception as e:  # noqa: BLE001
    print("(no bench line: %s)\n" % e)
stats = os.path.join(d, "bench_kernel_stats.csv")
if os.path.exists(stats):
    print("## `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline` (top kernels)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in list(csv.DictReader(open(stats)))[:16]:
        print("| `%s` | %s | %.1f | %.1f | %s |" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print()
print("## PMC passes on one launch of each hot kernel at the headline shard shape (n=1e6, M=1e4, D=1024)\n")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "pmc_*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "odx::gauss_knm" in k or "odx::gauss_mmv" in k or "odx::knm_pass" in k:      # incl. knm_pass2_kernel
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(d, "pmc_*kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k in agg:
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
for k, v in agg.items():
    c = {n: robust(x) for n, x in v.items()}
    ms = statistics.median(dur[k]) if dur[k] else float("nan")
    line = "* `%s`: %.2f ms/launch" % (k, ms)
    ghz = c["GRBM_GUI_ACTIVE"] / 8 / ms / 1e6 if ("GRBM_GUI_ACTIVE" in c and ms == ms) else None
    if ghz is not None and not 0.3 < ghz < 3.0:
        line += " (GRBM_GUI_ACTIVE sample implausible: %.3g cycles; clock and MFMA-busy not derived)" % c["GRBM_GUI_ACTIVE"]
        c.pop("GRBM_GUI_ACTIVE")
    elif ghz is not None:
        line += " at %.2f GHz" % ghz
    if "FETCH_SIZE" in c:
        # guide: on gfx950 FETCH_SIZE (KiB) counts 64 B per 128-B request for wide coalesced reads -> x2
        rd = 2 * c["FETCH_SIZE"] * 1024
        wr = c.get("WRITE_SIZE", 0.0) * 1024
        line += "; HBM read %.2f GB (FETCH_SIZE x2 correction), write %.2f GB" % (rd / 1e9, wr / 1e9)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        line += "; MFMA busy %.1f %% (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8))" % (
            100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0))
    if "SQ_LDS_BANK_CONFLICT" in c:
        line += "; LDS bank-conflict cycles %.3g of %.3g LDS-active" % (c["SQ_LDS_BANK_CONFLICT"], c.get("SQ_LDS_IDX_ACTIVE", 0))
    print(line)
    print("  raw: " + json.dumps({n: float("%.5g" % x) for n, x in c.items()}))

print("""
## Reading notes

* The preconditioner kernels (`gemm_nt_kernel<double, ..>`, `potrf_diag_kernel`, `trsm128_kernel`, `trtri_diag_kernel`,
  `kmm_epilogue_kernel`) run on side streams beside the main stream's kernels.  Their durations in the stats table are
  first-wave-to-last-wave times that include waiting for a free CU behind the main stream's workgroups (a
  `trsm128_kernel` takes 14 us and a `potrf_diag_kernel` 75 us on an idle GPU): the table's percentages add up to more
  than the wall time and say nothing about how much of the GPU those kernels used.
* `knm_pass_kernel`: algorithmic bytes per launch = n x M x 4 (40.0 GB at n = 1e6, M = 1e4); the FETCH_SIZE counter
  (doubled per the gfx950 correction of the guide) gives the same number: K_nM is read exactly once per pass.
  `knm_pass2_kernel` (one launch per class: the CG step whose periodic full residual rides along) reads the same
  40 GB once and forms two products from them.
* The Gaussian kernels issue 3 f16 MFMAs per algorithmic product (two-term f16 split); MFMA-busy and the clock the chip
  holds under them (GRBM_GUI_ACTIVE / 8 / time) are in the lines above: 62-63 % at ~1.9 GHz on the 256 x 256 tile core, i.e.
  ~1.25 PFLOP/s of f16 MFMA issued on random data.  Their FETCH_SIZE counts L2 misses, most of them served by the Infinity
  Cache (unique input: 4.1 GB of packed X + 41 MB of packed Z per launch).  In the bench table the same kernels take
  longer per launch than alone: the f64 MFMA work of the look-ahead (class-batched) preconditioners runs beside them.
* `gauss_knm_h2w256_kernel<true>` is the build with the fit's right-hand side K' (y / n) fused in (what the bench runs);
  the PMC passes launch the plain build `<false>`.
""")
