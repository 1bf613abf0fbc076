#!/usr/bin/env python3
"""precond_batched(ODX_C classes, M = ODX_M, D = 1024) a few times at the headline's width, for rocprofv3 --kernel-trace --stats
(tools/chain_timeline.py reads the trace): where the f64 preconditioner chain of the headline job spends its time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
C, M, D = int(os.environ.get("ODX_C", 6)), int(os.environ.get("ODX_M", 10000)), int(os.environ.get("ODX_D", 1024))
g = torch.Generator(device="cuda").manual_seed(0)
Zfs = [be.features(torch.randn((M, D), device="cuda", generator=g) * (20.0 / D ** 0.5)) for _ in range(C)]
out = torch.empty((C, 4, M, (M + 1) // 2 * 2), dtype=torch.float64, device="cuda")
be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = int(os.environ.get("ODX_REPS", 3))
for _ in range(reps):
    Ps = be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
flop = C * (5.0 * M ** 3 / 3 + 2.0 * M * M * D)
print("chain of %d classes, M = %d, D = %d: %.1f ms = %.1f TF f64 (5 M^3 / 3 + 2 M^2 D flop per class); pivots %s" % (
    C, M, D, dt * 1e3, flop / dt / 1e12, [int(p.info.item()) for p in Ps]))
