#!/usr/bin/env python3
"""HIP-event times of the hot kernels alone (idle GPU) at the headline shard shape: K_nM build with the fused right-hand
side, one- and two-vector pass, fused scoring.  ODX_N rows (default 1000000), ODX_M centres (default 10000), ODX_D."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return t[0], t[len(t) // 2]


be = odx.get_backend()
n, M, D = int(os.environ.get("ODX_N", 1000000)), int(os.environ.get("ODX_M", 10000)), int(os.environ.get("ODX_D", 1024))
X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
Z = X[:M].clone()
F, Zf = be.features(X), be.features(Z)
w = torch.randn(n, dtype=torch.float64, device="cuda")
fmt = be.knm_format(n, M)
buf = torch.empty(be.knm_bytes(n, M), dtype=torch.uint8, device="cuda")
state = {}


def build():
    state["K"] = be.knm_rhs(F, Zf, 15.0, w, out=buf)[0]


flop = 2.0 * n * M * D
lo, med = timed(build)
print("build[%s] n=%d M=%d D=%d: min %.2f ms, median %.2f ms  (%.0f algorithmic TFLOP/s)" % (fmt, n, M, D, lo, med, flop / med / 1e9))
K = state["K"]
v, v2 = torch.randn(M, dtype=torch.float64, device="cuda"), torch.randn(M, dtype=torch.float64, device="cuda")
lo, med = timed(lambda: be.ktk(K, v=v), reps=20)
print("pass: min %.3f ms, median %.3f ms  (%.0f GB/s)" % (lo, med, be.knm_bytes(n, M) / med / 1e6))
if be.can_ktk2(K):
    lo, med = timed(lambda: be.ktk2(K, v, v2), reps=10)
    print("two-vector pass: min %.3f ms, median %.3f ms" % (lo, med))
al = torch.randn(M, dtype=torch.float64, device="cuda")
out = torch.empty(n, 1, device="cuda")
lo, med = timed(lambda: be.mmv(F, Zf, 15.0, al, None, out=out))
print("score: min %.2f ms, median %.2f ms  (%.0f algorithmic TFLOP/s)" % (lo, med, flop / med / 1e9))
