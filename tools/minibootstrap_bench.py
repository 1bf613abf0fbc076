#!/usr/bin/env python3
"""Wall time of the reference-regime training on the GPU box: `OnlineRegionClassifier.trainRegionClassifier` (Minibootstrap
hard-negative mining) for 30 classes x 10 negative batches of 2000 rows with the shipped iCWT
constants (M = 2000, sigma, lambda from experiments/configs/config_online_detection_icwt30.yaml), then the 30 RLS box
regressors — through the drop-in module files, exactly as experiments/run_experiment_* call them.  Development aid.

    python tools/minibootstrap_bench.py [--dim 2048] [--classes 30] [--iters 10]
"""
import argparse
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402
import yaml  # noqa: E402

from tests import dropin  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=30)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--positives", type=int, default=800)
    ap.add_argument("--sigma", type=float, default=15.0)
    ap.add_argument("--streams", type=int, default=0, help="opts['class_streams']: classes trained concurrently (0 = reference order)")
    args = ap.parse_args()
    C, D, IT = args.classes, args.dim, args.iters
    names = ["_background_"] + ["c%d" % i for i in range(C)]
    cfg = {"NUM_CLASSES": C + 1,
           "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                        "CLASSIFIER": {"lambda": 0.0001, "sigma": args.sigma, "M": 2000, "kernel_type": "gauss"}},
           "REGION_REFINER": {"opts": {"lambda": 1000}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(names)}}
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    g = torch.Generator(device="cuda").manual_seed(1)
    mu = torch.randn((C, D), device="cuda", generator=g)

    def data():
        pos = [mu[c] + 0.7 * torch.randn((args.positives, D), device="cuda", generator=g) for c in range(C)]
        neg = [[mu[(c + 1 + j) % C] * 0.5 + 0.8 * torch.randn((2000, D), device="cuda", generator=g) for j in range(IT)] for c in range(C)]
        return pos, neg

    u = dropin.load("py_od_utils")
    clf_mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    for rep in range(2):                      # the first repetition warms every kernel and allocation
        pos, neg = data()
        with redirect_stdout(io.StringIO()):
            stats = u.computeFeatStatistics_torch(pos, neg, features_dim=D, pos_fraction=0.8)
            clf = clf_mod.FALKONWrapper(cfg_path=path)
            orc = orc_mod.OnlineRegionClassifier(clf, pos, neg, stats, cfg_path=path)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            models = orc.trainRegionClassifier(opts={"class_streams": args.streams} if args.streams else None)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        fits = sum(1 for m in models if m is not None) * IT
        print("rep %d: %d classes x %d batches (D = %d, M = 2000, %d positives, class_streams = %d): %.2f s = %.1f ms per (class, batch) "
              "[1 fit + 2 predicts + cache bookkeeping]; host returned after %.2f s" % (rep, C, IT, D, args.positives, args.streams, dt, dt / fits * 1e3, t_host))
    # the 30 box regressors on 1e4 rows per class
    n = 10000 * C
    COXY = {"C": (torch.arange(n, device="cuda") % C + 1).float().reshape(-1, 1), "X": torch.randn((n, D), device="cuda", generator=g),
            "Y": torch.randn((n, 4), device="cuda", generator=g) * 0.1}
    rr = dropin.load("region_refiner").RegionRefiner(path)
    for rep in range(2):
        with redirect_stdout(io.StringIO()):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rr.trainRegionRefiner(COXY)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("rep %d: %d RLS box regressors on %d rows each, D = %d: %.2f s" % (rep, C, n // C, D, dt))


if __name__ == "__main__":
    main()
