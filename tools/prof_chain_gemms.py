#!/usr/bin/env python3
"""What to run under `rocprofv3 --kernel-trace --pmc ... -- python3 tools/prof_chain_gemms.py SHAPE`: ten launches of ONE layer shape
of the conv5 head's chain at 8 images per call (2400 RoIs x 7 x 7 = 117 600 rows), so that a counter pass reads one shape per
kernel name.  SHAPE: conv1 (K 2048 -> 512, packed output only), conv2 (3 x 3 on 512 channels, taps gathered in the operand loads),
conv3 (K 512 -> 2048 + identity, f32 and packed output), down (K 1024 -> 2048, f32 output)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
shape = sys.argv[1] if len(sys.argv) > 1 else "conv2"
R, H, W = 2400, 7, 7
m = R * H * W
g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(s, generator=g).cuda()


if shape == "conv1":
    A, Wt = be.packed(rnd(m, 2048).relu_()), rnd(512, 2048) / 45
    wp, bounds = be.packed(Wt), be.weight_bounds(Wt)
    run = lambda: be.chain_gemm(A, wp, relu=True, bounds=bounds, f32_out=False, zero_row=True)   # noqa: E731
elif shape == "conv2":
    W1 = rnd(512, 64) / 8
    y = be.chain_gemm(be.packed(rnd(m, 64)), be.packed(W1), relu=True, bounds=be.weight_bounds(W1), f32_out=False, zero_row=True)
    Wt = rnd(512, 9 * 512) / 68
    wp, bounds = be.packed(Wt), be.weight_bounds(Wt)
    run = lambda: be.chain_conv3x3(y, R, H, W, wp, relu=True, bounds=bounds, f32_out=False)      # noqa: E731
elif shape == "conv3":
    A, Wt, res = be.packed(rnd(m, 512).relu_()), rnd(2048, 512) / 22, rnd(m, 2048).relu_()
    wp, bounds, rm = be.packed(Wt), be.weight_bounds(Wt), be.packed(res[:4096]).meta
    rm[1] = res.abs().max()
    run = lambda: be.chain_gemm(A, wp, residual=res, residual_meta=rm, relu=True, bounds=bounds, f32_out=True)   # noqa: E731
else:
    A, Wt = be.packed(rnd(m, 1024).relu_()), rnd(2048, 1024) / 32
    wp = be.packed(Wt)
    run = lambda: be.gemm_h2(A, wp, with_max=True)                                               # noqa: E731
for _ in range(10):
    out = run()
torch.cuda.synchronize()
