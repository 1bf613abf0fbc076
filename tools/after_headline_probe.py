#!/usr/bin/env python3
"""Why does the Minibootstrap (and the harvest loop) read slower in the process that has just run the headline job than in a fresh
one (round-5 review, item 6: 0.51 against 0.43 s)?  One process: the default Minibootstrap FRESH, then behind the headline job
(`--classes` classes of it), then again after each of a list of interventions that undo one thing the job left behind — the
stream-choice cache, the job's timing events, the allocator's cached blocks, the library's helper streams, a rest.  Whatever
intervention brings the figure back names the cause.   Usage (GPU box): python tools/after_headline_probe.py [--classes 30]"""
import argparse
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import odx  # noqa: E402
from odx import streams as odx_streams  # noqa: E402
from odx.job import LockstepClassJob  # noqa: E402
from odx.solver import SolverOptions  # noqa: E402
from tools import bench_extras as bx  # noqa: E402


def mini(tag):
    r = bx.minibootstrap_extra(modes=(("default", None),))
    print("%-58s Minibootstrap %.3f s" % (tag, r["s_default"]), flush=True)
    torch.cuda.empty_cache()
    return r["s_default"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", type=int, default=30)
    ap.add_argument("--timers", type=int, default=1, help="1: HIP-event timers around every kernel family, as bench.py hangs them")
    ap.add_argument("--runs", type=int, default=1, help="times the job is run (bench.py: warm-up + steps); every run creates and "
                    "releases the chains' helper streams")
    ap.add_argument("--no-fresh", action="store_true", help="do not run the Minibootstrap in the fresh process first (bench.py does not)")
    ap.add_argument("--prewarm-streams", type=int, default=0, help="pick this many measured-distinct side streams in the fresh process")
    ap.add_argument("--keep-helpers", action="store_true", help="do not release the helper streams between the runs (only at the end)")
    args = ap.parse_args()
    be = odx.get_backend()
    dev = torch.device("cuda", 0)
    if args.prewarm_streams:
        print("side streams picked in the fresh process:", len(odx_streams.distinct(args.prewarm_streams)), flush=True)
    if not args.no_fresh:
        mini("fresh process:")
        mini("fresh process, again:")
    N, D, M, C = 1_000_000, 1024, 10_000, args.classes
    X = bench.synth_rows(0, N, D, 30, 1237, dev)
    row_ids = torch.arange(0, N, device=dev)
    cidx = [torch.from_numpy(i).to(dev) for i in bench.centre_indices(N, 30, M, 1237)][:C]
    ph = {k: bench.Phase() for k in ("knm", "ktk", "ktk2", "precond", "mmv")} if args.timers else None
    job = LockstepClassJob(be, X, N, M, lambda c: torch.where((row_ids % 30) == c, 1.0, -1.0).to(torch.float64), cidx, 15.0, 1e-5, 20,
                           SolverOptions(check_pivots=False))
    infos = []
    real_release = be.release_helper_streams
    if args.keep_helpers:
        be.release_helper_streams = lambda: None
    for r in range(args.runs):
        t0 = time.perf_counter()
        job.run(be.features(X), list(range(C)), phases=ph, infos=infos if ph is not None else None)
        torch.cuda.synchronize()
        print("headline job, %d classes, run %d: %.2f s" % (C, r, time.perf_counter() - t0), flush=True)
    be.release_helper_streams = real_release
    be.release_helper_streams()
    job.release()
    job = X = row_ids = cidx = None
    be.release_workspaces()
    torch.cuda.empty_cache()
    mini("behind the job (buffers released, as bench.py does):")
    mini("... again:")
    # ---- interventions, one at a time
    odx_streams._CACHE.clear()
    mini("stream-choice cache dropped (streams probed again):")
    ph = infos = None
    gc.collect()
    mini("the job's timing events and status words dropped:")
    be.release_helper_streams()
    mini("library helper streams released once more:")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    mini("synchronised, allocator cache emptied:")
    time.sleep(20.0)
    mini("after 20 s of rest:")
    odx_streams._CACHE.clear()
    mini("stream-choice cache dropped once more:")


if __name__ == "__main__":
    main()
