#!/usr/bin/env python3
"""ms per image of the R-50-FPN trunk + pyramid (OnlineDetectionModelFPN.c4) at 1 / 4 / 8 images per call, f32, on the convolution
library (ODX_TRUNK=conv) and as row GEMMs on the library's tile cores; and of the whole forward per image.  Development aid."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.fpn import OnlineDetectionModelFPN  # noqa: E402

odx.get_backend()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)


def best(fn, reps=3):
    out = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = dt if out is None else min(out, dt)
    return out


for route in ("conv", "rows"):
    os.environ["ODX_TRUNK"] = route
    model = OnlineDetectionModelFPN().to(dev).eval()
    model.rows_min_positions = 0
    for B in (1, 4, 8):
        x = torch.randn((B, 3, 600, 800), generator=g).to(dev)
        with torch.no_grad():
            for _ in range(3):
                model.c4(x)
            t = best(lambda: [model.c4(x) for _ in range(4)])
        print("[%s] trunk + pyramid, %d image(s) per call: %.2f ms per image" % (route, B, t / (4 * B) * 1e3), flush=True)
    x = torch.randn((1, 3, 600, 800), generator=g).to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(x)
        t = best(lambda: [model(x) for _ in range(8)])
    print("[%s] whole forward, one image: %.2f ms" % (route, t / 8 * 1e3), flush=True)
