#!/usr/bin/env python3
"""Wall time of small FALKON fits (the minibootstrap regime of the reference: n ~ 1e4 rows, M ~ 1e3 centres)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402


def main():
    be = odx.get_backend()
    for n, D, M in ((4000, 2048, 1000), (12000, 2048, 2000), (22000, 1024, 2000)):
        X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
        y = torch.where(torch.rand(n, device="cuda") < 0.1, 1.0, -1.0).double()
        F = be.features(X)
        Zf = be.rows(F, torch.randperm(n)[:M])
        opt = odx.SolverOptions(check_pivots=False)
        for _ in range(2):
            odx.falkon_fit(be, F, y, Zf, 15.0, 1e-5, 20, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            odx.falkon_fit(be, F, y, Zf, 15.0, 1e-5, 20, opt)
        t_host = (time.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        print("n=%d D=%d M=%d: %.2f ms per fit (host enqueue %.2f ms)" % (n, D, M, t_all * 1e3, t_host * 1e3))


if __name__ == "__main__":
    main()
