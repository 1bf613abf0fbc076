#!/usr/bin/env python3
"""ms per image of the R-50-C4 forward at 1 / 2 / 4 / 8 images per call (extract.forward_batch) and of the harvest loop at the
same group sizes; f32 and bf16.  Development aid (GPU box).

    python tools/forward_batch_bench.py [--images 24]
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
from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor, forward_batch  # noqa: E402


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=24)
    ap.add_argument("--phases", action="store_true", help="device time of trunk / proposals / head per group size")
    ap.add_argument("--harvest-only", action="store_true")
    ap.add_argument("--groups", default="1,2,4,8")
    ap.add_argument("--pipe", default="1,0", help="harvest loop variants to time: 1 = forward of the next group under the harvest, 0 = plain loop")
    ap.add_argument("--parts", default="rpn,detector,mask")
    args = ap.parse_args()
    odx.get_backend()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn((args.images, 3, 600, 800), generator=g).to(dev)
    for name, dt in (() if args.harvest_only else (("f32", None), ("bf16", torch.bfloat16))):
        model = OnlineDetectionModel(compute_dtype=dt).to(dev).eval()
        for B in (1, 2, 4, 8):
            def run():
                for i in range(0, args.images, B):
                    if B == 1:
                        model(imgs[i:i + 1])
                    else:
                        forward_batch(model, imgs[i:i + B])
            with torch.no_grad():
                run()
                run()
                t = best(run)
                line = "[%s] %d image(s) per forward: %.2f ms per image" % (name, B, t / args.images * 1e3)
                if args.phases:
                    x = imgs[:B]
                    c4 = model.c4(x)
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    torch.cuda.synchronize()
                    ev[0].record()
                    for _ in range(5):
                        c4 = model.c4(x)
                    ev[1].record()
                    for _ in range(5):
                        props = model.proposals_batch(c4, (800, 600))
                    ev[2].record()
                    boxes = torch.cat([p[0] for p in props])
                    bidx = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor([len(p[0]) for p in props], device=dev))
                    for _ in range(5):
                        model.roi_head_maps(c4, boxes, batch_idx=bidx).mean(dim=(2, 3))
                    ev[3].record()
                    torch.cuda.synchronize()
                    line += " | alone, per image: trunk %.2f, proposals %.2f, RoIAlign + head %.2f ms" % tuple(
                        ev[k].elapsed_time(ev[k + 1]) / 5 / B for k in range(3))
            print(line, flush=True)
            if B > 1:
                model._group_graphs.enabled = True          # (opt-in path, see OnlineDetectionModel.__init__)

                def run_g():
                    for i in range(0, args.images, B):
                        model.forward_group(imgs[i:i + B], [None] * B)
                with torch.no_grad():
                    run_g()
                    run_g()
                    tg = best(run_g)
                    t0 = time.perf_counter()
                    run_g()
                    th = time.perf_counter() - t0          # (includes the one synchronisation per group)
                print("[%s] %d image(s) per forward, ONE HIP graph per group: %.2f ms per image (graphs kept: %d)" % (
                    name, B, tg / args.images * 1e3, len(model._group_graphs.graphs)), flush=True)
    # the harvest loop
    C = 30
    model = OnlineDetectionModel().to(dev).eval()
    samples = []
    for i in range(args.images):
        G = 1 + i % 3
        xy = torch.rand((G, 2), generator=g) * torch.tensor([500.0, 300.0])
        wh = 80 + torch.rand((G, 2), generator=g) * 200
        boxes = torch.cat((xy, xy + wh), dim=1)
        labels = [1 + (i + j) % C for j in range(G)]
        masks = torch.zeros((G, 600, 800), dtype=torch.uint8)
        for j in range(G):
            x1, y1, x2, y2 = [int(v) for v in boxes[j]]
            masks[j, y1 + 10:y2 - 10, x1 + 10:x2 - 10] = 1
        samples.append((imgs[i:i + 1], boxes.to(dev), labels, masks.to(dev)))
    for tb in [int(v) for v in args.groups.split(",")]:
        for pipe in [v == "1" for v in args.pipe.split(",")]:
            ex = OnlineFeatureExtractor(model, C, parts=tuple(args.parts.split(",")), pipeline=pipe, trunk_batch=tb)
            torch.manual_seed(0)
            ex.train(samples[:8])
            t = best(lambda: ex.train(samples))
            print("harvest %s, %d image(s) per forward, %s: %.2f ms per image" % (
                args.parts, tb, "pipelined" if pipe else "plain loop", t / args.images * 1e3), flush=True)


if __name__ == "__main__":
    main()
