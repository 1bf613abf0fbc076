#!/usr/bin/env python3
"""The CG's triangular products alone (development aid): odx_trmv_f64 with a lower and an upper M x M factor, M = 1e4 (400 MB per
product): microseconds per launch and TB/s, and the product against numpy at M = 700."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import odx  # noqa: E402
from odx import hip  # noqa: E402
from odx.backend import _p  # noqa: E402

be = odx.get_backend()
for M in (700, 10000):
    ld = (M + 1) // 2 * 2
    g = torch.Generator(device="cuda").manual_seed(M)
    A = torch.randn((M, ld), generator=g, device="cuda", dtype=torch.float64)
    x = torch.randn(M, generator=g, device="cuda", dtype=torch.float64)
    y = torch.empty(M, device="cuda", dtype=torch.float64)
    for uplo, name in ((0, "lower"), (1, "upper")):
        def run():
            hip.check(be.lib.odx_trmv_f64(_p(A), ld, M, uplo, _p(x), 1.0, 0.0, None, _p(y), be._stream()), "odx_trmv_f64")
        run()
        torch.cuda.synchronize()
        if M == 700:
            a = A.cpu().numpy()[:, :M]
            want = (np.triu(a) if uplo else np.tril(a)) @ x.cpu().numpy()
            print(name, "max rel err", float(np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()))
            continue
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200)
        print("M = %d %s: %.1f us per product, %.2f TB/s" % (M, name, best * 1e6, M * (M + 1) / 2 * 8 / best / 1e12))
