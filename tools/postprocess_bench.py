#!/usr/bin/env python3
"""Per-image cost of the detection post-processing (decode, clip, threshold, per-class NMS, top-k:
OnlineDetectionPostProcessor.py:12-79) for 300 proposals x 30 classes on the GPU box.  Development aid."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.postprocess import postprocess_detections  # noqa: E402

odx.get_backend()
g = torch.Generator().manual_seed(0)
R, C = 300, 30
xy = torch.rand((R, 2), generator=g) * torch.tensor([600.0, 400.0])
wh = 40 + torch.rand((R, 2), generator=g) * 200
props = torch.cat((xy, xy + wh), dim=1).cuda()
deltas = (torch.randn((R, 4 * (C + 1)), generator=g) * 0.1).cuda()
for name, scores in (("every score above the threshold (-2)", (torch.rand((R, C + 1), generator=g) * 2 - 1).cuda()),
                     ("trained-model-like scores (2 % above -0.9... threshold 0)", (torch.randn((R, C + 1), generator=g) * 0.4 - 1.0).cuda())):
    thr = -2.0 if "every" in name else 0.0
    for _ in range(3):
        res = postprocess_detections(scores, deltas, props, (800, 600), thr, 0.3, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        res = postprocess_detections(scores, deltas, props, (800, 600), thr, 0.3, 100)
    torch.cuda.synchronize()
    print("%s: %.2f ms per image, %d detections kept" % (name, (time.perf_counter() - t0) / 20 * 1e3, 0 if res is None else len(res["scores"])))
