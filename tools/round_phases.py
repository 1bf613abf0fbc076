#!/usr/bin/env python3
"""GPU time of the pieces of one class-batched Minibootstrap round (30 classes, M = 2000, D = 2048, n = 4000 rows per
class): the batched preconditioner chain, the K_nM builds and the CG loops on k streams.  Development aid."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.solver import SolverOptions, falkon_fit  # noqa: E402

be = odx.get_backend()
C, M, D, n = 30, 2000, 2048, 4000
g = torch.Generator(device="cuda").manual_seed(0)
Xs = [torch.randn((n, D), device="cuda", generator=g) * (20.0 / D ** 0.5) for _ in range(C)]
Fs = [be.features(x) for x in Xs]
Zfs = [be.rows(f, torch.arange(0, n, n // M)[:M]) for f in Fs]
ys = [be.vec(torch.where(torch.arange(n) % 5 == 0, 1.0, -1.0)) for _ in range(C)]
opt = SolverOptions(check_pivots=False)


def wall(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, t_host * 1e3


print("precond_batched(30): %.2f ms (host %.2f)" % wall(lambda: be.precond_batched(Zfs, 15.0, 1e-4, 1e-5)))
print("precond x 30 sequential: %.2f ms (host %.2f)" % wall(lambda: [be.precond(z, 15.0, 1e-4, 1e-5) for z in Zfs]))
Ps = be.precond_batched(Zfs, 15.0, 1e-4, 1e-5)
print("knm_rhs x 30: %.2f ms (host %.2f)" % wall(lambda: [be.knm_rhs(Fs[i], Zfs[i], 15.0, ys[i] / n) for i in range(C)]))
for k in (1, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(k)]

    def fits():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i in range(C):
            with torch.cuda.stream(streams[i % k]):
                falkon_fit(be, Fs[i], ys[i], Zfs[i], 15.0, 1e-4, 20, opt, precond=Ps[i])
        for s in streams:
            cur.wait_stream(s)
    print("30 fits (K_nM build + CG, given the factors) on %d streams: %.2f ms (host %.2f)" % ((k,) + wall(fits)))
al = torch.randn(M, dtype=torch.float64, device="cuda")
print("mmv x 30 (2000 rows each): %.2f ms (host %.2f)" % wall(lambda: [be.mmv(be.features(Xs[i][:2000]), Zfs[i], 15.0, al) for i in range(C)]))

from odx.falkon import GaussianKernel, InCoreFalkon, fit_batch  # noqa: E402
from odx.wrappers import CenterSelector  # noqa: E402


def batch_fit():
    ests = [InCoreFalkon(kernel=GaussianKernel(15.0), penalty=1e-4, M=M, maxiter=20,
                         center_selection=CenterSelector(torch.arange(0, n, n // M)[:M])) for _ in range(C)]
    for e in ests:
        e.options.check = False
    fit_batch(ests, Xs, [y.float() for y in ys])


print("fit_batch(30) = batched factors + 30 K_nM builds + one lock-step CG: %.2f ms (host %.2f)" % wall(batch_fit))
