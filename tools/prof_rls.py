#!/usr/bin/env python3
"""The 30 box regressors of BASELINE config 3 (n = 3e5, D = 1024, lambda = 1000) through RegionRefinerTrainer, three times
(the last one is the profiled one), for `rocprofv3 --kernel-trace [--stats | --pmc ...] -- python tools/prof_rls.py`."""
import io
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

from odx.rls import RegionRefinerTrainer  # noqa: E402

n, D, C, lam = 300_000, 1024, 30, 1000.0
g = torch.Generator(device="cuda").manual_seed(1234 + 3)
X = torch.randn((n, D), generator=g, device="cuda") * 0.6 + 0.15
cls = (torch.arange(n, device="cuda") % C) + 1
Y = torch.randn((n, 4), generator=g, device="cuda") * 0.2
cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}, "REGION_REFINER": {"opts": {"lambda": lam}}}
COXY = {"C": cls.float().view(-1, 1), "O": None, "X": X, "Y": Y}
for _ in range(3):
    torch.cuda.synchronize()
    with redirect_stdout(io.StringIO()):
        RegionRefinerTrainer(cfg, lam, False)(COXY)
torch.cuda.synchronize()
