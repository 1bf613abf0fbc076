#!/usr/bin/env python3
"""What a class-batched preconditioner chain (B classes, M centres) costs the kernel stream it runs beside: the chain alone,
then beside a run of CG passes, beside a run of K_nM builds and beside a run of scoring launches — extra time of that run
against its idle-GPU time, and how long the chain took.  ODX_N (1000000), ODX_M (10000), ODX_B (6)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
n, M, D, B = int(os.environ.get("ODX_N", 1000000)), int(os.environ.get("ODX_M", 10000)), 1024, int(os.environ.get("ODX_B", 6))
X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
Z = X[:M].clone()
F, Zf = be.features(X), be.features(Z)
w = torch.randn(n, dtype=torch.float64, device="cuda")
buf = torch.empty(be.knm_bytes(n, M), dtype=torch.uint8, device="cuda")
buf2 = torch.empty(be.knm_bytes(n, M), dtype=torch.uint8, device="cuda")
K = be.knm_rhs(F, Zf, 15.0, w, out=buf)[0]
v = torch.randn(M, dtype=torch.float64, device="cuda")
al = torch.randn(M, dtype=torch.float64, device="cuda")
out1 = torch.empty(n, 1, device="cuda")
Zc = torch.randn(B * M, D, device="cuda") * (20.0 / D ** 0.5)
Zfs = [be.features(Zc[b * M:(b + 1) * M]) for b in range(B)]
pout = torch.empty((B, 4, M, (M + 1) // 2 * 2), dtype=torch.float64, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def chain():
    be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=pout)


def run(work, reps, with_chain):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    if with_chain:
        side.wait_event(start)
        with torch.cuda.stream(side):
            e[2].record()
            chain()
            e[3].record()
    e[0].record()
    for _ in range(reps):
        work()
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]), (e[2].elapsed_time(e[3]) if with_chain else 0.0)


with torch.cuda.stream(side):
    chain()
torch.cuda.synchronize()
_, t_chain = run(lambda: None, 0, True)
print("chain of %d classes at M=%d alone: %.1f ms" % (B, M, t_chain))
for name, work, reps in (("passes", lambda: be.ktk(K, v=v), 120), ("builds", lambda: be.knm_rhs(F, Zf, 15.0, w, out=buf2), 8),
                         ("scoring", lambda: be.mmv(F, Zf, 15.0, al, None, out=out1), 8)):
    run(work, 2, False)
    t0, _ = run(work, reps, False)
    t1, tc = run(work, reps, True)
    print("%d %s: %.1f ms alone, %.1f ms beside the chain (+%.1f ms = %.2f of the chain's own %.1f ms); the chain took %.1f ms"
          % (reps, name, t0, t1, t1 - t0, (t1 - t0) / t_chain, t_chain, tc))
