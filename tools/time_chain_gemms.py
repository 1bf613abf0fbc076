#!/usr/bin/env python3
"""ms per launch of the conv5 head's four layer shapes at 8 images per call (tools/prof_chain_gemms.py's shapes), HIP events, 20 launches."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
R, H, W = 2400, 7, 7
m = R * H * W
g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(s, generator=g).cuda()


def timed(name, flop, run):
    for _ in range(3):
        run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print("%s: %.3f ms = %.0f TF algorithmic" % (name, ms, flop / ms / 1e9), flush=True)


A, Wt = be.packed(rnd(m, 2048).relu_()), rnd(512, 2048) / 45
wp, bounds = be.packed(Wt), be.weight_bounds(Wt)
timed("conv1 (K 2048 -> 512, packed out)", 2.0 * m * 2048 * 512, lambda: be.chain_gemm(A, wp, relu=True, bounds=bounds, f32_out=False, zero_row=True))
W1 = rnd(512, 64) / 8
y = be.chain_gemm(be.packed(rnd(m, 64)), be.packed(W1), relu=True, bounds=be.weight_bounds(W1), f32_out=False, zero_row=True)
Wt = rnd(512, 9 * 512) / 68
wp2, bounds2 = be.packed(Wt), be.weight_bounds(Wt)
timed("conv2 (3 x 3 on 512 channels, taps in the loads)", 2.0 * m * 4608 * 512, lambda: be.chain_conv3x3(y, R, H, W, wp2, relu=True, bounds=bounds2, f32_out=False))
del A
A3, Wt, res = be.packed(rnd(m, 512).relu_()), rnd(2048, 512) / 22, rnd(m, 2048).relu_()
wp3, bounds3, rm = be.packed(Wt), be.weight_bounds(Wt), be.packed(res[:4096]).meta
rm[1] = res.abs().max()
timed("conv3 (K 512 -> 2048 + identity, f32 + packed out)", 2.0 * m * 512 * 2048,
      lambda: be.chain_gemm(A3, wp3, residual=res, residual_meta=rm, relu=True, bounds=bounds3, f32_out=True))
timed("conv3, last block (f32 out only)", 2.0 * m * 512 * 2048, lambda: be.gemm_h2(A3, wp3, residual=res, relu=True))
A4, Wt = be.packed(rnd(m, 1024).relu_()), rnd(2048, 1024) / 32
wp4 = be.packed(Wt)
timed("down (K 1024 -> 2048, f32 out)", 2.0 * m * 1024 * 2048, lambda: be.gemm_h2(A4, wp4, with_max=True))
