#!/usr/bin/env python3
"""cProfile of the feature-harvest loop (forward + RPN / detector / mask harvesting) on the GPU box; development aid."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor  # noqa: E402

odx.get_backend()
dev = torch.device("cuda")
C = 30
model = OnlineDetectionModel().to(dev).eval()
g = torch.Generator().manual_seed(3)
samples = []
for i in range(24):
    img = torch.randn((1, 3, 600, 800), generator=g)
    G = 1 + i % 3
    xy = torch.rand((G, 2), generator=g) * torch.tensor([500.0, 300.0])
    wh = 80 + torch.rand((G, 2), generator=g) * 200
    boxes = torch.cat((xy, xy + wh), dim=1)
    labels = [1 + (i + j) % C for j in range(G)]
    masks = torch.zeros((G, 600, 800), dtype=torch.uint8)
    for j in range(G):
        x1, y1, x2, y2 = [int(v) for v in boxes[j]]
        masks[j, y1 + 10:y2 - 10, x1 + 10:x2 - 10] = 1
    samples.append((img.to(dev), boxes.to(dev), labels, masks.to(dev)))
parts = tuple(sys.argv[1].split("+")) if len(sys.argv) > 1 else ("rpn", "detector", "mask")
ex = OnlineFeatureExtractor(model, C, parts=parts)
torch.manual_seed(0)
ex.train(samples[:2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
ex.train(samples)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:8000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
