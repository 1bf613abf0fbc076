#!/usr/bin/env python3
"""fit_batch(30 classes, M = 2000, D = 2048, n = 4000) a few times, for rocprofv3 --kernel-trace (tools/chain_timeline.py
lists the last call: batched factors, 30 K_nM builds, one lock-step CG)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.falkon import GaussianKernel, InCoreFalkon, fit_batch  # noqa: E402
from odx.wrappers import CenterSelector  # noqa: E402

be = odx.get_backend()
C, M, D, n = 30, 2000, 2048, 4000
g = torch.Generator(device="cuda").manual_seed(0)
Xs = [torch.randn((n, D), device="cuda", generator=g) * (20.0 / D ** 0.5) for _ in range(C)]
ys = [torch.where(torch.arange(n) % 5 == 0, 1.0, -1.0).cuda() for _ in range(C)]
for _ in range(3):
    ests = [InCoreFalkon(kernel=GaussianKernel(15.0), penalty=1e-4, M=M, maxiter=20,
                         center_selection=CenterSelector(torch.arange(0, n, n // M)[:M])) for _ in range(C)]
    for e in ests:
        e.options.check = False
    fit_batch(ests, Xs, ys)
    torch.cuda.synchronize()
