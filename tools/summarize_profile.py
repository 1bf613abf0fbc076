#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the markdown committed under profiles/:
the bench line, the kernel-stats table of the profiled bench, one line per hot kernel from the PMC passes (HBM bytes with
the guide's gfx950 FETCH_SIZE correction, clock, MFMA busy, wait shares, LDS conflicts), and the same for the rest of the
path (x_<tag>_kernel_stats.csv / xpmc_<tag>_* of part 2: feature forward, RLS chain, Minibootstrap)."""
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


def short(name):
    return name.split("(")[0].replace("void ", "").replace("odx::", "")


def stats_table(path, rows=16, title=None, skip=()):
    if not os.path.exists(path):
        return
    if title:
        print(title)
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in [r for r in csv.DictReader(open(path)) if not any(w in r["Name"] for w in skip)][:rows]:
        print("| `%s` | %s | %.1f | %.1f | %s |" % (short(r["Name"])[-70:], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print()


def pmc_lines(d, prefix, want):
    """One line per kernel whose name contains one of `want`, from <prefix>*counter_collection.csv / *kernel_trace.csv."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, prefix + "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if any(w in k for w in want):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(d, prefix + "*kernel_trace.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k in agg:
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]] or [0])):
        c = {n: robust(x) for n, x in v.items()}
        ms = statistics.median(dur[k]) if dur[k] else float("nan")
        line = "* `%s`: %.3f ms/launch (median of %d)" % (k[-90:], ms, len(dur[k]))
        ghz = c["GRBM_GUI_ACTIVE"] / 8 / ms / 1e6 if ("GRBM_GUI_ACTIVE" in c and ms == ms and ms > 0) else None
        if ghz is not None and ms < 0.3:
            c.pop("GRBM_GUI_ACTIVE")            # (the guide: the quotient reads high on dispatches shorter than ~0.3 ms)
        elif ghz is not None and not 0.3 < ghz < 3.0:
            line += " (GRBM_GUI_ACTIVE sample implausible: %.3g cycles; clock and MFMA-busy not derived)" % c["GRBM_GUI_ACTIVE"]
            c.pop("GRBM_GUI_ACTIVE")
        elif ghz is not None:
            line += " at %.2f GHz" % ghz
        if "FETCH_SIZE" in c:
            # guide: on gfx950 FETCH_SIZE (KiB) counts 64 B per 128-B request for wide coalesced reads -> x2
            rd = 2 * c["FETCH_SIZE"] * 1024
            wr = c.get("WRITE_SIZE", 0.0) * 1024
            line += "; HBM read %.3f GB (FETCH_SIZE x2 correction), write %.3f GB" % (rd / 1e9, wr / 1e9)
            if ms == ms and ms > 0:
                line += " = %.2f TB/s" % ((rd + wr) / ms / 1e9)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            line += "; MFMA busy %.1f %% (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8))" % (
                100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0))
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            line += "; of the wave cycles %.0f %% wait (memory / barrier), %.0f %% issue stall, %.0f %% issuing" % (
                100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"])
        if "SQ_LDS_BANK_CONFLICT" in c:
            line += "; LDS bank-conflict cycles %.3g of %.3g LDS-active" % (c["SQ_LDS_BANK_CONFLICT"], c.get("SQ_LDS_IDX_ACTIVE", 0))
        print(line)
        print("  raw: " + json.dumps({n: float("%.5g" % x) for n, x in c.items()}))
    print()


def marker_windows(path, marker="axpby"):
    """Per-kernel totals of the launches between consecutive PAIRS of marker kernels of a kernel-trace CSV (tools/prof_forward.py
    brackets one steady-state image of each network with them)."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    out = []
    for a, b in zip(marks[0::2], marks[1::2]):
        win = rows[a + 1:b]
        if not win:
            continue
        tot = collections.defaultdict(lambda: [0, 0])
        busy, cur = 0, int(win[0]["Start_Timestamp"])
        for r in win:
            s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            t = tot[short(r["Kernel_Name"])[-70:]]
            t[0] += 1
            t[1] += e0 - s0
            if e0 > cur:
                busy += e0 - max(s0, cur)
                cur = e0
        span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
        out.append((len(win), span / 1e6, busy / 1e6, sorted(tot.items(), key=lambda kv: -kv[1][1])))
    return out


d = sys.argv[1]
print("# rocprofv3 summary (%s)\n" % os.path.basename(d.rstrip("/")))
try:
    bench = json.loads(open(os.path.join(d, "bench_n1.json")).read().strip().splitlines()[-1])
    print("bench.py (un-profiled): value %.1f %s, %.1f ms/step, dtype: %s\n" % (bench["value"], bench["unit"], bench["ms_per_step"], bench["dtype"]))
    for key in ("roofline", "roofline_hbm", "roofline_mfma"):
        if key in bench:
            print("%s: %s\n" % (key, json.dumps(bench[key])))
    print("phases (ms/step, rank 0): %s\n" % json.dumps(bench["phases_ms_per_step_rank0"]))
    print("cpu_baseline: %s\n" % json.dumps(bench.get("cpu_baseline")))
    for key in ("config2", "config2_bf16", "config4", "config5_shard", "config5_shard_f8", "rls", "forward", "forward_fpn", "detect",
                "minibootstrap"):
        if key in bench:
            print("%s: %s\n" % (key, json.dumps(bench[key])))
except Exception as e:  # noqa: BLE001
    print("(no bench line: %s)\n" % e)

stats_table(os.path.join(d, "bench_kernel_stats.csv"),
            title="## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (top kernels)\n")
if glob.glob(os.path.join(d, "pmc_*counter_collection.csv")):
    print("## PMC passes on three launches of each hot kernel at the headline shard shape (n = 1e6, M = 1e4, D = 1024; tools/prof_kernels.py)\n")
    pmc_lines(d, "pmc_", ("gauss_knm", "gauss_mmv", "knm_pass", "split_f16", "row_sqnorm"))
    print("""Reading notes

* The preconditioner kernels (`gemm_nt_kernel<double, ..>`, `potrf_diag_kernel`, `trsm128_kernel`, `trtri_diag_kernel`,
  `kmm_epilogue_kernel`) run on a side stream beside the main stream's kernels.  Their durations in the stats table are
  first-wave-to-last-wave times that include waiting for a free CU behind the main stream's workgroups: the table's
  percentages add up to more than the wall time and say nothing about how much of the GPU those kernels used.
* Pass kernels: algorithmic bytes per launch = the stored K_nM shard, n x M x 3 B = 30.0 GB in the 24-bit fixed-point storage
  large blocks get (n x M x 4 = 40 GB with ODX_KNM=f32).  FETCH_SIZE (doubled per the gfx950 correction of the guide, which is
  calibrated for 16-byte-per-lane loads; the compact pass loads 8 + 4 bytes per lane) should be read against that.
* The Gaussian kernels issue 3 f16 MFMAs per algorithmic product (two-term f16 split).  Their FETCH_SIZE counts L2 misses,
  most of them served by the Infinity Cache (unique input: 4.1 GB of packed X + 41 MB of packed Z per launch).  In the bench
  table the same kernels take longer per launch than alone: the f64 MFMA work of the look-ahead (class-batched)
  preconditioners runs beside them.
""")

for tag, title, want in (("forward", "feature forward: R-50-C4 and R-50-FPN, 10 images each (tools/prof_forward.py)",
                          ("gemm_h2", "split_f16", "roi_align", "nms_", "Cijk", "miopen", "MIOpen", "igemm", "conv", "naive", "Im2d2Col", "gemm")),
                         ("rls", "RLS chain: 30 regressors, n = 3e5, D = 1024 (tools/prof_rls.py, three repetitions)",
                          ("gemm_nt_kernel", "rls_", "potrf", "trsm", "trtri", "trmv")),
                         ("minibootstrap", "Minibootstrap, reference regime, default mode, two repetitions (tools/prof_minibootstrap.py)", ())):
    sp = os.path.join(d, "x_%s_kernel_stats.csv" % tag)
    tp = os.path.join(d, "x_%s_kernel_trace.csv" % tag)
    if tag == "forward" and os.path.exists(tp):
        print("## %s\n" % title)
        print("(The whole-process stats of this script are dominated by the convolution library's solver search on the first image of "
              "each network — every applicable solver, the naive reference one included, is run once per new shape; the tables below are "
              "ONE steady-state image of each network, cut out of the kernel trace between two marker launches.)\n")
        for (cnt, span, busy, tot), name in zip(marker_windows(tp), ("R-50-C4 (300 RoIs, conv5 head)", "R-50-FPN (1000 RoIs, fc6 / fc7)")):
            print("### %s: %d launches, %.2f ms first to last under the profiler, GPU busy %.2f ms\n" % (name, cnt, span, busy))
            print("| kernel | launches | total us | avg us |")
            print("|---|---|---|---|")
            for k, v in tot[:16]:
                print("| `%s` | %d | %.1f | %.1f |" % (k, v[0], v[1] / 1e3, v[1] / 1e3 / v[0]))
            print()
        if glob.glob(os.path.join(d, "xpmc_%s_*counter_collection.csv" % tag)):
            print("PMC (FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of the same script; medians over all launches of a kernel):\n")
            pmc_lines(d, "xpmc_%s_" % tag, ("gemm_h2", "split_f16", "roi_align", "nms_"))
        continue
    if os.path.exists(sp):
        stats_table(sp, rows=18, title="## %s\n" % title)
        if want and glob.glob(os.path.join(d, "xpmc_%s_*counter_collection.csv" % tag)):
            print("PMC (FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of the same script):\n")
            pmc_lines(d, "xpmc_%s_" % tag, want)

for tag, title in (("forward_b4", "R-50-C4 forward, FOUR 600 x 800 images per call (extract.forward_batch), f32, 15 calls (tools/prof_forward_batch.py 4 f32)"),
                   ("forward_b4_bf16", "the same in bf16 (compute_dtype = bfloat16: trunk stages, RPN head and conv5 head on 16-bit rows, odx_gemm_b16 / odx_gemm_b16_taps)"),
                   ("forward_b8", "R-50-C4 forward, EIGHT images per call, f32: trunk stages, RPN head and conv5 head as one chain of row GEMMs "
                                  "(tools/prof_forward_batch.py 8 f32)"),
                   ("forward_fpn_b8", "R-50-FPN forward, EIGHT images per call, f32: trunk stages and pyramid as row GEMMs, the proposal stage per "
                                      "level for the group (tools/prof_forward_batch.py 8 f32 fpn)")):
    sp = os.path.join(d, "x_%s_kernel_stats.csv" % tag)
    if os.path.exists(sp):
        stats_table(sp, rows=16, skip=("naive_conv",), title="## %s\n\n(whole-process statistics, 15 calls; the convolution library's "
                    "solver search on the first calls runs its naive reference kernels once per shape — left out of the table; the percentages "
                    "are of the process's total including them)\n" % title)

for tag, anchors in (("forward_b8", ("stem_pool_rows", "max_pool")), ("forward_fpn_b8", ("stem_pool_rows", "max_pool"))):
    tp = os.path.join(d, "x_%s_kernel_trace.csv" % tag)
    if os.path.exists(tp):
        import csv as _csv
        rows = sorted(_csv.DictReader(open(tp)), key=lambda r: int(r["Start_Timestamp"]))
        idx = []
        for anchor in anchors:                   # (the stem's fused tail; the library's pooling in profiles made before it)
            idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
            if idx:
                break
        if idx:
            i0 = idx[-1]
            t0 = int(rows[i0]["Start_Timestamp"])
            print("\n### kernel timeline of the last call of %s (us from the stem's pooling; launches of 20 us and more)\n" % tag)
            print("| start | us | kernel | grid |\n|---|---|---|---|")
            busy = 0.0
            for r in rows[max(i0 - 3, 0):]:
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                busy += dur
                if dur >= 20:
                    print("| %.0f | %.1f | `%s` | %s |" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur,
                                                         r["Kernel_Name"].replace("odx::", "").replace("void ", "")[:60], r["Grid_Size_X"]))
            print("\nGPU busy from the stem on: %.2f ms for the call's 8 images\n" % (busy / 1e3))
