#!/usr/bin/env python3
"""Rate of the compact-format CG pass (odx_knm_fwd_bwd_q) alone, per thread / chunk / row-block configuration.
Usage: python tools/passq_bench.py [n M]   (ODX_PASSQ_CFG="nt ch r" is set per run by this script)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.backend import Knm  # noqa: E402

be = odx.get_backend()
n, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500000, 10000)
g = torch.Generator(device="cuda").manual_seed(0)


def block(fmt):
    K = Knm()
    K.n, K.M, K.fmt = n, M, fmt
    if fmt == "f32":
        K.ld = (M + 3) // 4 * 4
        K.K = torch.rand((n, K.ld), device="cuda", generator=g)
        return K, n * K.ld * 4
    K.ld = (M + 7) // 8 * 8
    K.K = torch.randint(-32768, 32767, (n, K.ld), dtype=torch.int16, device="cuda", generator=g)
    if fmt == "u24":
        K.lo = torch.randint(0, 255, (n, K.ld), dtype=torch.uint8, device="cuda", generator=g)
    return K, n * K.ld * (3 if fmt == "u24" else 2)


def rate(K, nbytes, reps=8):
    v = torch.randn(M, dtype=torch.float64, device="cuda", generator=g)
    o = torch.empty(M, dtype=torch.float64, device="cuda")
    for _ in range(3):                      # (the first launches of a configuration run slower)
        be.ktk(K, v=v, out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        be.ktk(K, v=v, out=o)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, nbytes / ms / 1e6


K, nb = block("f32")
print("f32  default           : %.3f ms  %.0f GB/s" % rate(K, nb))
del K
for fmt in ("u24", "bf16"):
    K, nb = block(fmt)
    os.environ.pop("ODX_PASSQ_CFG", None)
    print("%-4s default           : %.3f ms  %.0f GB/s" % ((fmt,) + rate(K, nb)))
    for cfg in sys.argv[3:] or ["512 5 4", "512 5 6", "1024 3 4", "1024 3 6"]:
        os.environ["ODX_PASSQ_CFG"] = cfg
        try:
            print("%-4s cfg %-14s: %.3f ms  %.0f GB/s" % ((fmt, cfg) + rate(K, nb)))
        except Exception as e:  # noqa: BLE001
            print("%-4s cfg %-14s: %s" % (fmt, cfg, e))
    os.environ.pop("ODX_PASSQ_CFG", None)
    del K
