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

odx.get_backend()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.randn((1, 3, 600, 800), device="cuda", generator=g)
with torch.no_grad():
    if which in ("c4", "both"):
        m = OnlineDetectionModel(post_nms_top_n=300).cuda().eval()
        for _ in range(reps):
            m(img)
        torch.cuda.synchronize()
    if which in ("fpn", "both"):
        m = OnlineDetectionModelFPN().cuda().eval()
        for _ in range(reps):
            m(img)
        torch.cuda.synchronize()
