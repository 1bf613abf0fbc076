#!/usr/bin/env python3
"""The class-batched preconditioner chain alone (idle GPU): B classes of M centres, HIP-event time per call.
Under `rocprofv3 --kernel-trace --stats` the per-kernel split of the same chain.  ODX_M (10000), ODX_B (6), ODX_D (1024)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
M, B, D = int(os.environ.get("ODX_M", 10000)), int(os.environ.get("ODX_B", 6)), int(os.environ.get("ODX_D", 1024))
X = torch.randn(B * M, D, device="cuda") * (20.0 / D ** 0.5)
Zfs = [be.features(X[b * M:(b + 1) * M]) for b in range(B)]
out = torch.empty((B, 4, M, (M + 1) // 2 * 2), dtype=torch.float64, device="cuda")
reps = int(os.environ.get("ODX_REPS", 3))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out)
ev[0].record()
for i in range(reps):
    Ps = be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]
print("precond_batched B=%d M=%d D=%d: %.1f ms per call, %.1f ms per class, %.1f TFLOP/s of f64 at 5/3 M^3 + M^2 D per class; info %s"
      % (B, M, D, ms, ms / B, B * (5.0 / 3.0 * M ** 3 + float(M) * M * D) / ms / 1e9, [int(p.info) for p in Ps]))
