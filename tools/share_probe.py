#!/usr/bin/env python3
"""What sharing the chip between the HBM-bound CG passes and the class-batched preconditioner chains costs each side:
passes on 256 - r CUs (odx_set_pass_reserved_cus) alone and beside a batched chain of 6 classes at M = 1e4 on a side
stream.  Development aid (DESIGN §7 "Next")."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.backend import Knm  # noqa: E402

be = odx.get_backend()
n, M, D, G = int(os.environ.get("ODX_N", 1000000)), 10000, 1024, 6
ld = (M + 3) // 4 * 4
K = Knm()
K.K, K.n, K.M, K.ld = torch.rand((n, ld), device="cuda"), n, M, ld
v = torch.randn(M, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
Zfs = [be.features(torch.randn((M, D), device="cuda", generator=g) * (20.0 / D ** 0.5)) for _ in range(G)]
out = torch.empty((G, 4, M, M), dtype=torch.float64, device="cuda")
side = torch.cuda.Stream()


def passes(k):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        be.ktk(K, v=v)
    b.record()
    return a, b


be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out)
torch.cuda.synchronize()
t_chain = (time.perf_counter() - t0) * 1e3
print("batched chain of %d classes alone: %.1f ms (%.1f ms per class, %.1f TF f64)" % (G, t_chain, t_chain / G, G * 5 / 3 * M ** 3 / t_chain / 1e9))
for r in (0, 32, 64, 96):
    be.reserve_cus_during_passes(r)
    be.ktk(K, v=v)
    torch.cuda.synchronize()
    a, b = passes(10)
    torch.cuda.synchronize()
    alone = a.elapsed_time(b) / 10
    # beside the chain: keep passing until the chain is done
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ev0.record()
        be.precond_batched(Zfs, 15.0, 1e-5, 1e-5, out=out, ws_key="probe")
        ev1.record()
    k = max(4, int(1.3 * t_chain / alone))
    a, b = passes(k)
    torch.cuda.synchronize()
    both = a.elapsed_time(b) / k
    chain = ev0.elapsed_time(ev1)
    print("reserve %3d CUs: pass alone %.2f ms (%.0f GB/s) | beside the chain: pass %.2f ms (%.0f GB/s) over %d passes, chain %.1f ms (alone %.1f)"
          % (r, alone, n * M * 4 / alone / 1e6, both, n * M * 4 / both / 1e6, k, chain, t_chain))
be.reserve_cus_during_passes(0)
