#!/usr/bin/env python3
"""precond_batched(30 classes, M = 2000, D = 2048) a few times, for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
C, M, D = int(os.environ.get("ODX_C", 30)), int(os.environ.get("ODX_M", 2000)), 2048
g = torch.Generator(device="cuda").manual_seed(0)
Zfs = [be.features(torch.randn((M, D), device="cuda", generator=g) * (20.0 / D ** 0.5)) for _ in range(C)]
for _ in range(4):
    be.precond_batched(Zfs, 15.0, 1e-4, 1e-5)
torch.cuda.synchronize()
