#!/usr/bin/env python3
"""Latency of the test-time heads for one image: 300 RoIs x 30 classes (M centres each) + 30 box regressors."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.heads import OnlineBoxPredictor  # noqa: E402


def main():
    be = odx.get_backend()
    R, D, C = 300, 2048, 30
    for M in (1000, 2000):
        g = torch.Generator().manual_seed(0)
        clfs = []
        for c in range(C):
            m = odx.InCoreFalkon(kernel=odx.GaussianKernel(20.0), penalty=1e-4, M=M)
            m.ny_points_ = (torch.randn(M, D, generator=g) * (20.0 / D ** 0.5)).cuda()
            m.alpha_ = torch.randn(M, 1, generator=g, dtype=torch.float64).cuda()
            clfs.append(m)
        regs = [{"mu": torch.zeros(4), "T": torch.eye(4), "T_inv": torch.eye(4),
                 "Beta": {str(k): {"weights": torch.randn(D + 1, generator=g) * 0.01} for k in range(4)}} for _ in range(C)]
        stats = {"mean": torch.zeros(D).cuda(), "std": torch.ones(D).cuda(), "mean_norm": torch.tensor(20.0).cuda()}
        head = OnlineBoxPredictor(clfs, regs, stats)
        F = (torch.randn(R, D, generator=g) * (20.0 / D ** 0.5)).cuda()
        for _ in range(3):
            head(F)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            s, d = head(F)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        flops = 2.0 * R * C * M * D
        print("heads R=%d C=%d M=%d D=%d: %.3f ms per image (scoring %.1f GFLOP -> %.1f TFLOP/s incl. everything)"
              % (R, C, M, D, dt * 1e3, flops / 1e9, flops / dt / 1e12))


if __name__ == "__main__":
    main()
