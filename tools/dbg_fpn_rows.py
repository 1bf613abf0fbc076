import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch, odx
from odx.fpn import OnlineDetectionModelFPN
odx.get_backend()
torch.manual_seed(0)
m = OnlineDetectionModelFPN(width=16, fpn_channels=32, mlp_dim=64).cuda().eval()
x = torch.randn(2, 3, 160, 224).cuda()
with torch.no_grad():
    cs = m.backbone(x)
    rs = m.backbone.forward_rows(x)
    for k, (c, (r, (B, H, W))) in enumerate(zip(cs, rs)):
        got = r.X.view(B, H, W, -1).permute(0, 3, 1, 2)
        print("C%d" % (k + 2), tuple(c.shape), tuple(got.shape), float((got - c).abs().max()) / float(c.abs().max()))
    ps = m.fpn(cs)
    pr = m.fpn.forward_rows(rs)
    for k, (a, b) in enumerate(zip(pr, ps)):
        print("P%d" % (k + 2), tuple(a.shape), tuple(b.shape), float((a - b).abs().max()) / float(b.abs().max()))
    pr2 = m.fpn.forward_rows([(type(r)(c.permute(0, 2, 3, 1).reshape(-1, c.shape[1]).contiguous(), r.n, r.D, odx.get_backend().packed(c.permute(0, 2, 3, 1).reshape(-1, c.shape[1]).contiguous()).P,
                                odx.get_backend().packed(c.permute(0, 2, 3, 1).reshape(-1, c.shape[1]).contiguous()).meta), d) for c, (r, d) in zip(cs, rs)])
    for k, (a, b) in enumerate(zip(pr2, ps)):
        print("P%d from the library's C maps" % (k + 2), float((a - b).abs().max()) / float(b.abs().max()))
