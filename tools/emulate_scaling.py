#!/usr/bin/env python3
"""Per-rank compute of the headline job at 1 / 2 / 4 / 8 ranks, measured on ONE GPU (round-4 review, item 5): for each world
size W, `bench.py --emulate-world W --emulate-rank r` runs rank r's share — its N / W rows for all 30 classes (builds, passes,
scoring), the preconditioner chains of the classes it owns, collectives replaced by local copies of the right size — and
reports the rank's compute-only step time.  The table it prints (and writes as markdown) is a PREDICTION of the strong-scaling
curve, unmeasured on hardware: job time ~ the slowest rank's compute + the collectives (counted, priced with the guide's xGMI
figures).  Usage (GPU box): python tools/emulate_scaling.py [--out gpurun_out/r05_emulated_scaling.md] [--steps 1]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XGMI_LINK_GBS = 153.0          # /opt/skills/guides: per xGMI link, 7 links per GPU
COLL_LATENCY_US = 25.0         # a small RCCL collective over xGMI (latency-bound; assumption, stated in the table)


def run(world, rank, steps, exchange="lockstep"):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-extras",
           "--cg-exchange", exchange]
    if world > 1:
        cmd += ["--emulate-world", str(world), "--emulate-rank", str(rank)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, check=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_emulated_scaling.md"))
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--exchanges", default="lockstep,allreduce", help="lockstep: all-gather + reduce-scatter per iteration and batch, "
                    "one owner per class; allreduce: the replicated one-all-reduce-per-iteration form of the north star")
    args = ap.parse_args()
    rows = []
    for ex in args.exchanges.split(","):
        base = None
        for W in [int(w) for w in args.worlds.split(",")]:
            ranks = [0] if W == 1 else sorted({0, W - 1})
            res = {r: run(W, r, args.steps, ex) for r in ranks}
            slow = max(res.values(), key=lambda d: d["ms_per_step"])
            ms = slow["ms_per_step"]
            M, C, it = slow["config"]["M"], slow["config"]["classes"], 20
            b = slow["config"]["lockstep_batch"]
            if W == 1:
                t_coll = 0.0
            elif ex == "lockstep":
                # per lock-step batch: (it + 1 + 2 full-gradient) mmv exchanges of one all-gather + one reduce-scatter of (W, M) f64, plus
                # the right-hand-side reduce-scatter and the final gather; each rank sends / receives (W - 1) / W of the matrix over 7 links
                batches = (C + b - 1) // b
                ncoll = batches * (2 * (it + 2) + 2)
                t_coll = ncoll * (COLL_LATENCY_US * 1e-6 + W * M * 8 * (W - 1) / W / (7 * XGMI_LINK_GBS * 1e9))
            else:
                # per class: the right-hand side + it all-reduces of the (M,) f64 partial, one of them two vectors wide (the folded
                # full residual); a ring all-reduce moves 2 (W - 1) / W of the vector per rank
                ncoll = C * (it + 1)
                t_coll = ncoll * (COLL_LATENCY_US * 1e-6) + C * (it + 2) * M * 8 * 2 * (W - 1) / W / (7 * XGMI_LINK_GBS * 1e9)
            # the centres: one all-gather of the rows each rank owns per class, (W - 1) / W of the (M, D) f32 block per rank over one ring link
            centre = 0.0 if W == 1 else C * (M * 1024 * 4) * (W - 1) / W / (XGMI_LINK_GBS * 1e9)
            total = ms / 1e3 + t_coll + centre
            if base is None:
                base = total
            rows.append((ex, W, {r: d["ms_per_step"] for r, d in res.items()}, b, slow["phases_ms_per_step_rank0"], t_coll, centre, total,
                         base / total, base / total / W, slow["roofline_step"]["achieved"]))
            print(rows[-1], flush=True)
    with open(args.out, "w") as f:
        f.write("# Headline job at 1 / 2 / 4 / 8 ranks: per-rank compute measured on ONE MI355X (bench.py --emulate-world), PREDICTED job time\n\n")
        f.write("Predicted, unmeasured on more than one GPU.  `compute` = the emulated rank's step (all kernels of its shard + the chains it "
                "builds; collectives are local copies); `collectives` = their count x (%.0f us + bytes over 7 x %.0f GB/s links), `centres` = "
                "the all-gather of the centres' rows per class over one ring link — both ASSUMED, on the main stream (no overlap credited).  "
                "`lockstep`: one owner rank per class, all-gather + reduce-scatter per CG iteration and batch (what bench.py times); "
                "`allreduce`: every rank holds every class's factors and runs its M-sized algebra, ONE all-reduce per CG iteration "
                "(the north star's literal form).\n\n" % (COLL_LATENCY_US, XGMI_LINK_GBS))
        f.write("| exchange | ranks | compute per rank, ms (rank: ms) | classes per batch | K_nM builds / passes / scoring, ms | collectives, s | centres, s | predicted step, s | speed-up | efficiency | roofline_step (compute only) |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for ex, W, per, b, ph, tc, ce, tot, sp, eff, rs in rows:
            f.write("| %s | %d | %s | %d | %.0f / %.0f / %.0f | %.3f | %.3f | %.2f | %.2f | %.2f | %.2f |\n" % (
                ex, W, ", ".join("%d: %.0f" % kv for kv in sorted(per.items())), b, ph["knm"], ph["ktk"] + ph["ktk2"], ph["mmv"], tc, ce, tot, sp, eff, rs))
    print(open(args.out).read())


if __name__ == "__main__":
    main()
