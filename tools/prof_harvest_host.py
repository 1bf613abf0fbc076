#!/usr/bin/env python3
"""Host profile (cProfile) of the harvest loop, warm: python tools/prof_harvest_host.py [trunk_batch] (development aid)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor  # noqa: E402

odx.get_backend()
tb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
g = torch.Generator().manual_seed(3)
n, C = 32, 30
imgs = torch.randn((n, 3, 600, 800), generator=g).to(dev)
model = OnlineDetectionModel().to(dev).eval()
samples = []
for i in range(n):
    G = 1 + i % 3
    xy = torch.rand((G, 2), generator=g) * torch.tensor([500.0, 300.0])
    wh = 80 + torch.rand((G, 2), generator=g) * 200
    boxes = torch.cat((xy, xy + wh), dim=1)
    masks = torch.zeros((G, 600, 800), dtype=torch.uint8)
    for j in range(G):
        x1, y1, x2, y2 = [int(v) for v in boxes[j]]
        masks[j, y1 + 10:y2 - 10, x1 + 10:x2 - 10] = 1
    samples.append((imgs[i:i + 1], boxes.to(dev), [1 + (i + j) % C for j in range(G)], masks.to(dev)))
ex = OnlineFeatureExtractor(model, C, parts=("rpn", "detector", "mask"), pipeline=True, trunk_batch=tb)
torch.manual_seed(0)
ex.train(samples[:2 * tb])
ex.train(samples)
ts = []
for _ in range(7):
    torch.manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ex.train(samples)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / n * 1e3)
ts.sort()
print("loop: %.2f ms per image (best of 7 passes; median %.2f, worst %.2f)" % (ts[0], ts[3], ts[-1]))
if os.environ.get("PROFILE", "0") != "1":
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
ex.train(samples)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(38)
