#!/usr/bin/env python3
"""Where the wall time of one RLS training goes on the HOST (cProfile of a warm repetition) beside its GPU span."""
import cProfile
import io
import os
import pstats
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import time  # noqa: E402

import torch  # noqa: E402

import odx  # noqa: E402
from odx.rls import RegionRefinerTrainer  # noqa: E402

odx.get_backend()
n, D, C, lam = 300_000, 1024, 30, 1000.0
g = torch.Generator(device="cuda").manual_seed(1237)
X = torch.randn((n, D), generator=g, device="cuda") * 0.6 + 0.15
cls = (torch.arange(n, device="cuda") % C) + 1
Y = torch.randn((n, 4), generator=g, device="cuda") * 0.2
cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}, "REGION_REFINER": {"opts": {"lambda": lam}}}
COXY = {"C": cls.float().view(-1, 1), "O": None, "X": X, "Y": Y}


def once():
    with redirect_stdout(io.StringIO()):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        RegionRefinerTrainer(cfg, lam, False)(COXY)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3


for _ in range(3):
    once()
print("host returns after %.2f ms, GPU done after %.2f ms" % once())
pr = cProfile.Profile()
pr.enable()
once()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
