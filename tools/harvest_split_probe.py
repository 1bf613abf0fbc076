#!/usr/bin/env python3
"""Where a harvested image's time goes (development aid, GPU box): the forward of the image groups alone (host time to queue
it / wall time), then the three harvesters alone on the forward's cached results, one at a time and together — wall time per
image and host synchronisations are what bound the pipelined loop (both threads share the interpreter lock).

    python tools/harvest_split_probe.py [--images 24] [--group 4]
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
from odx.extract import OnlineDetectionModel, forward_batch, grid_anchors  # noqa: E402
from odx.harvest import DetectorHarvester, MaskHarvester, RPNHarvester, project_masks_on_boxes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=24)
    ap.add_argument("--group", type=int, default=4)
    args = ap.parse_args()
    odx.get_backend()
    dev = torch.device("cuda")
    C, n, B = 30, args.images, args.group
    model = OnlineDetectionModel().to(dev).eval()
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn((n, 3, 600, 800), generator=g).to(dev)
    gts, labels, masks = [], [], []
    for i in range(n):
        G = 1 + i % 3
        xy = torch.rand((G, 2), generator=g) * torch.tensor([500.0, 300.0])
        wh = 80 + torch.rand((G, 2), generator=g) * 200
        bx = torch.cat((xy, xy + wh), dim=1)
        mk = torch.zeros((G, 600, 800), dtype=torch.uint8)
        for j in range(G):
            x1, y1, x2, y2 = [int(v) for v in bx[j]]
            mk[j, y1 + 10:y2 - 10, x1 + 10:x2 - 10] = 1
        gts.append(bx.to(dev)), labels.append([1 + (i + j) % C for j in range(G)]), masks.append(mk.to(dev))

    def forward_all():
        items = []
        with torch.no_grad():
            for i in range(0, n, B):
                per, c4s, maps, offs = forward_batch(model, imgs[i:i + B], gts[i:i + B])
                ts = model.rpn_activation(c4s)
                anchors = grid_anchors(c4s.shape[2], c4s.shape[3], model.stride, model.cells.to(dev))
                for j in range(len(per)):
                    G = len(labels[i + j])
                    act = model.mask_activation(maps[offs[j]:offs[j] + G])
                    items.append({"boxes": per[j][0], "feats": per[j][1], "t": ts[j], "anchors": anchors, "act": act,
                                  "mg": project_masks_on_boxes(masks[i + j], gts[i + j], act.shape[2]), "i": i + j})
        return items

    for _ in range(2):
        items = forward_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    items = forward_all()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("forward (+ RPN activation, mask activation, mask projection), %d per call: host %.2f ms per image to queue, %.2f ms wall"
          % (B, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))

    def harvest(parts):
        hv_det = DetectorHarvester(model.feat_dim, C, num_images=n, device=dev, iterations=10, batch_size=2000) if "det" in parts else None
        hv_rpn = RPNHarvester(model.backbone.out_channels, 15, num_images=n, device=dev, iterations=10, batch_size=2000) if "rpn" in parts else None
        hv_mask = MaskHarvester(model.mask_dim, C, device=dev) if "mask" in parts else None
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in items:
            i = it["i"]
            if hv_rpn is not None:
                hv_rpn.add_image(it["t"], it["anchors"], (800, 600), gts[i])
            if hv_det is not None:
                hv_det.add_image(it["feats"], it["boxes"], gts[i], labels[i], [800, 600])
            if hv_mask is not None:
                hv_mask.add_image(it["act"], it["mg"], labels[i])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3

    for parts in (("rpn",), ("det",), ("mask",), ("rpn", "det", "mask")):
        harvest(parts)
        h, w = min(harvest(parts) for _ in range(3))
        print("harvest %s on cached forwards: %.2f ms per image on the host, %.2f ms wall" % ("+".join(parts), h, w))


if __name__ == "__main__":
    main()
