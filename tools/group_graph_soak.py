#!/usr/bin/env python3
"""Soak of the graphed group forward (ODX_GROUP_GRAPH): N rounds of forward_group on changing images, every result checked against
the launch-by-launch forward_batch of the same images, with other GPU work (a second stream's products, host reads) in between;
then whole harvest passes with the graph on against passes with it off.  Development aid (GPU box): python tools/group_graph_soak.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor, forward_batch  # noqa: E402

odx.get_backend()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
model = OnlineDetectionModel(seed=6).to(dev).eval()
model.rpn_logits.weight.data.normal_(0, 0.3)
model.rpn_deltas.weight.data.normal_(0, 0.05)
model._group_graphs.enabled = True
pool = torch.randn((24, 3, 600, 800), generator=g).to(dev)
side = torch.cuda.Stream()
side2 = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
bad = 0
t0 = time.time()
with torch.no_grad():
    for it in range(rounds):
        idx = torch.randint(0, 24, (4,), generator=g).tolist()
        x = pool[idx]
        G = 1 + it % 3
        gts = [torch.tensor([[20.0 + 7 * k, 30.0, 220.0 + 5 * k, 260.0]] * G) for k in range(4)]
        if it % 3 == 1:                                     # (replays called from changing streams, another graph replayed in between:
            with torch.cuda.stream(side2):                  # the sequence round 5 could not explain)
                side2.wait_stream(torch.cuda.current_stream())
                res = model.forward_group(x, gts)
            torch.cuda.current_stream().wait_stream(side2)
        else:
            res = model.forward_group(x, gts)
        if it % 4 == 2:
            model.c4(pool[it % 24:it % 24 + 1])
        with torch.cuda.stream(side):                       # other work beside / between the replays
            for _ in range(3):
                junk @ junk
        _ = float(junk[0, 0])                               # a host read in between
        model._group_graphs.enabled = False
        per, _, _, _ = forward_batch(model, x, gts)
        model._group_graphs.enabled = True
        for b in range(4):
            ok = res[b]["boxes"].shape == per[b][0].shape and bool(torch.equal(res[b]["boxes"], per[b][0]))
            ok = ok and float((res[b]["feats"] - per[b][1]).abs().max()) <= 1e-5 * float(per[b][1].abs().max())
            if not ok:
                bad += 1
                print("round %d image %d differs (boxes %s / %s)" % (it, b, tuple(res[b]["boxes"].shape), tuple(per[b][0].shape)), flush=True)
        if it % 10 == 9:
            print("round %d: %d mismatching images so far, graphs kept %d, %.0f s" % (it + 1, bad, len(model._group_graphs.graphs), time.time() - t0), flush=True)
print("forward soak: %d rounds, %d mismatching images" % (rounds, bad), flush=True)

# whole harvest passes
C = 30
samples = []
for i in range(16):
    Gn = 1 + i % 3
    xy = torch.rand((Gn, 2), generator=g) * torch.tensor([500.0, 300.0])
    wh = 80 + torch.rand((Gn, 2), generator=g) * 200
    boxes = torch.cat((xy, xy + wh), dim=1)
    masks = torch.zeros((Gn, 600, 800), dtype=torch.uint8)
    for j in range(Gn):
        x1, y1, x2, y2 = [int(v) for v in boxes[j]]
        masks[j, y1 + 10:y2 - 10, x1 + 10:x2 - 10] = 1
    samples.append((pool[i:i + 1], boxes.to(dev), [1 + (i + j) % C for j in range(Gn)], masks.to(dev)))


def harvest(graph):
    model._group_graphs.enabled = graph
    torch.manual_seed(0)
    out = OnlineFeatureExtractor(model, C, parts=("rpn", "detector", "mask"), trunk_batch=4).train(samples)
    return out


ref = harvest(False)
bad_h = 0
for it in range(int(os.environ.get('SOAK_PASSES', '6'))):
    got = harvest(True)
    for part in ("detector", "rpn"):
        a, b = got[part], ref[part]
        for xa, xb in zip(a[1], b[1]):                       # positives per class
            if xa.shape != xb.shape or float((xa - xb).abs().max() if xa.numel() else 0.0) > 1e-4 * max(1.0, float(xb.abs().max()) if xb.numel() else 1.0):
                bad_h += 1
print("harvest soak: graphed passes against the launch-by-launch pass, %d differing positive blocks" % bad_h, flush=True)
