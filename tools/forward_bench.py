#!/usr/bin/env python3
"""Feature-forward timings on the GPU box (rows A11-A13 of SURVEY §8; BASELINE config 2's first half): one synthetic
600 x 800 image, random weights, 300 RoIs — R-50-C4 trunk, RPN head + proposals (HIP NMS), RoIAlign (HIP), conv5 head,
and the whole `OnlineDetectionModel.forward`, in f32 and under bf16 autocast.  Development aid; numbers quoted in
DESIGN.md come from here.

    python tools/forward_bench.py [--images 8] [--rois 300]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rois", type=int, default=300)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--graph", action="store_true", help="also time the trunk replayed from a captured HIP graph")
    ap.add_argument("--channels-last", action="store_true", help="model and image in NHWC memory format")
    args = ap.parse_args()
    be = odx.get_backend()
    dev = torch.device("cuda")
    model = OnlineDetectionModel(post_nms_top_n=args.rois).to(dev).eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randn((1, 3, args.height, args.width), device=dev, generator=g)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)

    for name, ctx in (("f32", torch.autocast("cuda", enabled=False)), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
        with torch.no_grad(), ctx:
            c4 = model.backbone(img)
            boxes, _ = model.proposals(c4.float(), (args.width, args.height))
            R = boxes.shape[0]
            t_trunk = timeit(lambda: model.backbone(img))
            t_prop = timeit(lambda: model.proposals(c4.float(), (args.width, args.height)))
            rois = torch.cat((torch.zeros((R, 1), device=dev), boxes), dim=1)
            c4f = c4.float().contiguous()
            t_roi = timeit(lambda: be.roi_align(c4f, rois, 1.0 / 16, (14, 14), 0))
            crops = be.roi_align(c4f, rois, 1.0 / 16, (14, 14), 0)
            t_head = timeit(lambda: model.head(crops).mean(dim=(2, 3)))
            t_all = timeit(lambda: model(img))
            if args.graph:
                static = img.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        model.backbone(static)
                torch.cuda.current_stream().wait_stream(side)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    out_static = model.backbone(static)
                ref = model.backbone(img)
                gr.replay()
                torch.cuda.synchronize()
                print("  trunk as a captured graph: %.2f ms (max abs diff vs eager %.3g)" % (timeit(lambda: gr.replay()), float((out_static.float() - ref.float()).abs().max())))
        C = c4.shape[1]
        out_bytes = R * C * 14 * 14 * 4
        print("[%s] %dx%d image, %d RoIs: trunk %.2f ms | RPN head + proposals (top-k, HIP NMS) %.2f ms | RoIAlign %.3f ms "
              "(%.0f GB/s of output written; %d x %d x 14 x 14 f32) | conv5 head + pool %.2f ms | whole forward %.2f ms = %.1f images/s"
              % (name, args.height, args.width, R, t_trunk, t_prop, t_roi, out_bytes / t_roi / 1e6, R, C, t_head, t_all, 1e3 / t_all))

    # NMS alone at the RPN's pre-NMS size
    for Rn in (2000, 6000):
        b = torch.rand((Rn, 4), device=dev, generator=g) * 500
        b[:, 2:] = b[:, :2] + 20 + torch.rand((Rn, 2), device=dev, generator=g) * 200
        s = torch.rand(Rn, device=dev, generator=g)
        t = timeit(lambda: be.nms(b, s, 0.7))
        print("nms R=%d thr 0.7: %.3f ms" % (Rn, t))


if __name__ == "__main__":
    main()
