#!/usr/bin/env python3
"""Per-image cost of the feature-harvest loop (rows A11-A13 + f1 of SURVEY §8) on the GPU box: synthetic 600 x 800 images
with a few ground-truth boxes, random weights, 30 classes — the forward alone against forward + harvesting of the
detector rows, the on-line RPN rows and the mask pixel rows.  Development aid.

    python tools/harvest_bench.py [--images 24]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=24)
    args = ap.parse_args()
    odx.get_backend()
    dev = torch.device("cuda")
    C = 30
    model = OnlineDetectionModel().to(dev).eval()
    g = torch.Generator().manual_seed(3)
    samples = []
    for i in range(args.images):
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

    def run(parts, pipeline=True, tb=2):
        ex = OnlineFeatureExtractor(model, C, parts=parts, pipeline=pipeline, trunk_batch=tb)
        torch.manual_seed(0)
        ex.train(samples[:3])                 # warm-up
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ex.train(samples)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / len(samples) * 1e3)
        return min(ts), sorted(ts)[1]

    with torch.no_grad():
        for _ in range(3):                    # MIOpen picks its algorithms on the first calls at a shape
            for s in samples[:3]:
                model(s[0], s[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in samples:
            model(s[0], s[1])
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / len(samples) * 1e3
    print("forward alone: %.2f ms per image" % fwd)
    for parts in (("detector",), ("rpn",), ("rpn", "detector"), ("rpn", "detector", "mask")):
        print("forward + harvest %s: %.2f ms per image (median of 3: %.2f); one image per trunk call: %.2f; without the forward / harvest pipeline: %.2f" % (
            ("+".join(parts),) + run(parts) + (run(parts, True, 1)[0], run(parts, False)[0],)))


if __name__ == "__main__":
    main()
