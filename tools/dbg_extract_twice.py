import os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "online-detection_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ODX_ROWS_MIN_POSITIONS", "0")
import torch, odx
from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor
import test_extract as T
odx.get_backend()
C = 3
model = OnlineDetectionModel(width=16, post_nms_top_n=40, pre_nms_top_n=400).cuda().eval()
samples = T._samples(8, 192, 256, C, seed=3)
if os.environ.get("DBG_NOGRAPH") == "1":
    model._trunk_graphs.enabled = False
res = []
import threading
from odx.extract import GraphedCall
_orig = GraphedCall.__call__
_n = [0]
def _checked(self, x, *more, **kw):
    out = _orig(self, x, *more, **kw)
    if self is model._trunk_graphs:
        torch.cuda.current_stream().synchronize()
        ref = self.fn(x, *more)
        torch.cuda.current_stream().synchronize()
        d = float((out - ref).abs().max()) / float(ref.abs().max())
        _n[0] += 1
        print("   call", _n[0], threading.current_thread().name, "stream", hex(torch.cuda.current_stream().cuda_stream), "graphs", len(self.graphs),
              "rel diff graph vs eager %.3g" % d, "out max %.3g" % float(out.abs().max()), flush=True)
    return out
if os.environ.get("DBG_CHECK") == "1":
    GraphedCall.__call__ = _checked
for it in range(4):
    torch.manual_seed(1)
    if os.environ.get("TB", "4") == "0":
        from odx.extract import DetectorFeatureExtractor
        neg, pos, COXY = DetectorFeatureExtractor(model, num_classes=C, iterations=2, batch_size=40).train(samples)
    else:
        ex = OnlineFeatureExtractor(model, C, parts=("detector",), trunk_batch=int(os.environ.get("TB", "4")))
        out = ex.train(samples)
        neg, pos, COXY = out["detector"] if isinstance(out, dict) else out
    res.append([p.clone() for p in pos])
    with torch.no_grad():
        x0 = samples[0][0].cuda()
        c4g, c4e = model.c4(x0), model._c4_eager(x0)
        bx = torch.tensor([[20., 30., 120., 150.], [5., 5., 60., 90.]]).cuda()
        fe_ = model.roi_features(c4e, bx)
        if it == 0:
            fe0 = fe_.clone()
        print(it, "graph vs eager c4:", float((c4g - c4e).abs().max()) / float(c4e.abs().max()), "head now vs pass 0:", float((fe_ - fe0).abs().max()) / float(fe0.abs().max()),
              "graphs", len(model._trunk_graphs.graphs))
    if it:
        print(it, [[round(float(x), 4) for x in ((a - b).abs().max(dim=1)[0] / b.abs().max())] for a, b in zip(res[it], res[0])])
        print(it, [tuple(p.shape) for p in pos], [float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(res[it], res[0])])
