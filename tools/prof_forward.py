#!/usr/bin/env python3
"""The feature forward a few times — R-50-C4 (trunk, RPN proposals, RoIAlign rows, conv5 head as split-f16 GEMMs) and
R-50-FPN (pyramid, five-level RPN, multi-level RoIAlign, fc6 / fc7) on one synthetic 600 x 800 image, f32 — for
`rocprofv3 --kernel-trace [--stats | --pmc ...] -- python tools/prof_forward.py [c4|fpn|both]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel  # noqa: E402
from odx.fpn import OnlineDetectionModelFPN  # noqa: E402

be = odx.get_backend()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.randn((1, 3, 600, 800), device="cuda", generator=g)
mk = torch.zeros(2, dtype=torch.float64, device="cuda")


def marker():
    """A kernel that appears nowhere else in the forward (odx_axpby_f64): tools/summarize_profile.py cuts the trace at it,
    so that the tables show ONE steady-state image and not the convolution library's solver search on the first one."""
    torch.cuda.synchronize()
    be.axpby(0.0, mk[:1], 1.0, mk[1:])
    torch.cuda.synchronize()


with torch.no_grad():
    for name, make in (("c4", lambda: OnlineDetectionModel(post_nms_top_n=300)), ("fpn", OnlineDetectionModelFPN)):
        if which not in (name, "both"):
            continue
        m = make().cuda().eval()
        for _ in range(reps):
            m(img)
        marker()
        m(img)
        marker()
        del m
