#!/usr/bin/env python3
"""A/B of two builds of the 256 x 256 Gaussian tile core IN ONE PROCESS (the guide's rule: never compare across processes or
boxes): ODX_H2_STAGING=reg (operands staged through registers) against =dma (LDS-DMA), alternating, K_nM build with the
fused right-hand side and fused scoring at the headline shard shape.  ODX_N / ODX_M / ODX_D override the shape; any other
environment switch can be A/B'd the same way: ODX_AB_VAR=NAME ODX_AB_A=... ODX_AB_B=..."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

var = os.environ.get("ODX_AB_VAR", "ODX_H2_STAGING")
arms = (os.environ.get("ODX_AB_A", "reg"), os.environ.get("ODX_AB_B", "dma"))
be = odx.get_backend()
n, M, D = int(os.environ.get("ODX_N", 1000000)), int(os.environ.get("ODX_M", 10000)), int(os.environ.get("ODX_D", 1024))
X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
Z = X[:M].clone()
F, Zf = be.features(X), be.features(Z)
w = torch.randn(n, dtype=torch.float64, device="cuda")
buf = torch.empty(be.knm_bytes(n, M), dtype=torch.uint8, device="cuda")
al = torch.randn(M, dtype=torch.float64, device="cuda")
out = torch.empty(n, 1, device="cuda")
ops = {"build": lambda: be.knm_rhs(F, Zf, 15.0, w, out=buf), "score": lambda: be.mmv(F, Zf, 15.0, al, None, out=out)}
times = {(o, a): [] for o in ops for a in arms}
for rnd in range(int(os.environ.get("ODX_AB_ROUNDS", 6))):
    for a in arms:
        os.environ[var] = a
        for o, fn in ops.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[(o, a)].append(e0.elapsed_time(e1) / 3)
flop = 2.0 * n * M * D
for o in ops:
    ta, tb = (statistics.median(times[(o, a)]) for a in arms)
    print("%s n=%d M=%d D=%d: %s=%s %.2f ms (%.0f TF)   %s=%s %.2f ms (%.0f TF)   B/A time %.3f" % (
        o, n, M, D, var, arms[0], ta, flop / ta / 1e9, var, arms[1], tb, flop / tb / 1e9, tb / ta))
