#!/usr/bin/env python3
"""Is the conv5 head on the wide tile core the same function of its input from call to call (development aid)?  The head on fixed
rows, launch by launch, with other allocations in between; then the same from a captured graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import GraphedCall, OnlineDetectionModel  # noqa: E402


def rel(a, b):
    return float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-30)


def between(dev):
    x = torch.randn((60000, 2048), device=dev)
    y = x.index_select(0, torch.randint(0, 60000, (70000,), device=dev))
    return float(y.sum())


def main():
    be = odx.get_backend()
    dev = torch.device("cuda")
    m = OnlineDetectionModel().to(dev).eval()
    R = int(os.environ.get("ROIS", "1216"))
    rows = torch.randn((R * 49, 1024), generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        def head(rows):
            return m.head.forward_rows(rows, R, 7, 7).float().mean(dim=(2, 3))
        ref = head(rows).clone()
        print("max |feature| %.4g" % float(ref.abs().max()))
        for i in range(3):
            between(dev)
            print("eager call %d vs the first: rel %.3g" % (i + 2, rel(head(rows), ref)), flush=True)
        # layer by layer: which product differs between two eager calls
        blk = m.head.layer4[0]
        from odx.extract import _bottleneck_rows_h2
        a = _bottleneck_rows_h2(be, blk, rows, R, 7, 7).clone()
        between(dev)
        b = _bottleneck_rows_h2(be, blk, rows, R, 7, 7)
        print("first bottleneck, two eager calls: rel %.3g" % rel(b, a), flush=True)
        xp = be.packed(rows)
        w1 = blk._folded[("conv1/h2", torch.float32)]
        y1 = be.gemm_h2(xp, w1[0], bias=w1[1], relu=True).clone()
        between(dev)
        y1b = be.gemm_h2(be.packed(rows), w1[0], bias=w1[1], relu=True)
        print("conv1 GEMM (m = %d, n = %d, K = 1024), two eager calls: rel %.3g; vs f64 torch: %.3g" % (
            y1.shape[0], y1.shape[1], rel(y1b, y1),
            rel(y1, torch.relu(rows.double() @ blk._fold("conv1", blk.conv1, blk.bn1, rows)[0].reshape(y1.shape[1], -1).double().t()
                               + blk._fold("conv1", blk.conv1, blk.bn1, rows)[1].double()))), flush=True)
        t = be.packed_taps3x3(y1, R, 7, 7)
        w2 = blk._folded[("conv2/h2", torch.float32)]
        y2 = be.gemm_h2(t, w2[0], bias=w2[1], relu=True).clone()
        between(dev)
        y2b = be.gemm_h2(be.packed_taps3x3(y1, R, 7, 7), w2[0], bias=w2[1], relu=True)
        print("conv2 GEMM over the 9-tap operand, two eager calls: rel %.3g" % rel(y2b, y2), flush=True)
        gc = GraphedCall(head)
        gc(rows)
        g1 = gc(rows).clone()
        print("graph, first replay vs eager: rel %.3g" % rel(g1, ref))
        for i in range(3):
            between(dev)
            print("graph replay %d vs eager: rel %.3g" % (i + 2, rel(gc(rows), ref)), flush=True)


if __name__ == "__main__":
    main()
