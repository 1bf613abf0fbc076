#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE'S OWN CODE
(/root/reference, read-only) in the build container.  Container-only: nothing here runs on
the GPU box or in the test suite; only the .npz / .json data files it writes are committed.

How the reference is executed without its un-installable dependencies:
  * source text is read from /root/reference, device literals 'cuda' are retargeted to 'cpu'
    IN MEMORY and the module is exec'd under its own name (nothing is copied to disk);
  * torch.eig (removed from current PyTorch) is shimmed with torch.linalg.eig;
  * `maskrcnn_benchmark.structures.bounding_box.BoxList` is a 10-line stand-in placed in
    sys.modules (the reference only uses .bbox / .size / add_field on this path);
  * `falkon` is a recording stand-in: it pins the CALL CONTRACT of FALKONWrapper (constructor
    keywords, fit/predict, the centre-selector protocol), not falkon's arithmetic.

    python tests/golden/make_golden.py
"""
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load_ref(relpath, name, retarget=True, extra=None):
    src = open(os.path.join(REF, relpath)).read()
    if retarget:
        src = src.replace("'cuda'", "'cpu'").replace('"cuda"', '"cpu"')
    mod = types.ModuleType(name)
    mod.__file__ = os.path.join(REF, relpath)
    if extra:
        mod.__dict__.update(extra)
    sys.modules[name] = mod
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    return mod


def eig_shim(S, eigenvectors=False):
    w, V = torch.linalg.eig(S)
    return torch.stack([w.real, w.imag], dim=1), V.real


class BoxList:
    def __init__(self, bbox, image_size, mode="xyxy"):
        self.bbox = torch.as_tensor(bbox)
        self.size = image_size
        self.mode = mode
        self.extra_fields = {}

    def add_field(self, k, v):
        self.extra_fields[k] = v

    def get_field(self, k):
        return self.extra_fields[k]


def install_boxlist():
    for n in ("maskrcnn_benchmark", "maskrcnn_benchmark.structures", "maskrcnn_benchmark.structures.bounding_box"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["maskrcnn_benchmark.structures.bounding_box"].BoxList = BoxList


def write_cfg(path, classes, lam_rls, extra_rpn=False):
    import yaml
    cfg = {"NUM_CLASSES": len(classes),
           "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                        "CLASSIFIER": {"lambda": 0.001, "sigma": 10, "M": 40, "kernel_type": "gauss"}},
           "REGION_REFINER": {"opts": {"lambda": lam_rls}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(classes)}}
    if extra_rpn:
        cfg["RPN"] = {k: v for k, v in cfg.items()}
    yaml.safe_dump(cfg, open(path, "w"))
    return cfg


# --------------------------------------------------------------------------- A7 / A8: RLS
def make_rls():
    torch.eig = eig_shim
    sys.path.insert(0, os.path.join(REF, "src/modules/region-refiner"))
    trainer = load_ref("src/modules/region-refiner/region_refiner_trainer/train_region_refiner.py", "ref_train_rr")
    predictor = load_ref("src/modules/region-refiner/region_predictor/predict_regions.py", "ref_predict_rr")
    classes = ["_background_", "a", "b", "c", "d"]
    cfg = {"CHOSEN_CLASSES": {i: c for i, c in enumerate(classes)}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
    g = torch.Generator().manual_seed(11)
    n, D = 420, 24
    X = torch.randn(n, D, generator=g) * 2.0 + 0.5
    C = torch.randint(1, 5, (n, 1), generator=g).float()
    C[C == 3] = 2  # class 3 has no rows -> all-None model
    Y = torch.randn(n, 4, generator=g) * torch.tensor([0.1, 0.2, 0.3, 0.15]) + torch.tensor([0.01, -0.02, 0.1, 0.0])
    Y[:, 1] += 0.5 * Y[:, 0]  # correlated targets so the whitening matters
    out = {"X": X.numpy(), "C": C.numpy(), "Y": Y.numpy(), "lambda": np.float64(10.0), "classes": np.array(classes)}
    for tag, is_rpn in (("det", False), ("rpn", True)):
        COXY = {"C": C.clone() if not is_rpn else C.clone() - 1, "O": None, "X": X.clone(), "Y": Y.clone()}
        with redirect_stdout(io.StringIO()):
            models = trainer.RegionRefinerTrainer(cfg, 10.0, is_rpn)(COXY)
        out[tag + "_num_models"] = np.int64(len(models))
        for i, m in enumerate(models):
            out["%s_%d_none" % (tag, i)] = np.bool_(m["Beta"] is None)
            if m["Beta"] is None:
                continue
            out["%s_%d_mu" % (tag, i)] = m["mu"].numpy()
            out["%s_%d_T" % (tag, i)] = m["T"].numpy()
            out["%s_%d_T_inv" % (tag, i)] = m["T_inv"].numpy()
            out["%s_%d_W" % (tag, i)] = np.stack([m["Beta"][str(k)]["weights"].numpy() for k in range(4)])
            out["%s_%d_losses" % (tag, i)] = np.stack([m["Beta"][str(k)]["losses"].numpy() for k in range(4)])
        if not is_rpn:
            det_models = models
    # apply (A8): RegionPredictor on two "images"; models with Beta None cannot be applied by the
    # reference (it indexes ['Beta']['0']) -> use a 3-class cfg whose models all exist
    classes3 = ["_background_", "a", "b"]
    cfg3 = {"CHOSEN_CLASSES": {i: c for i, c in enumerate(classes3)}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
    boxes, feats = [], []
    for im in range(2):
        R = 7 + im
        xy = torch.rand(R, 2, generator=g) * 200
        wh = torch.rand(R, 2, generator=g) * 150 + 5
        bb = torch.cat([xy, xy + wh], dim=1)
        boxes.append(BoxList(bb.clone(), (320, 240)))
        f = torch.randn(R + 2, D, generator=g) * 2.0 + 0.5
        gt = np.zeros(R + 2)
        gt[[0, 3]] = 1  # ground-truth rows are excluded by the predictor
        feats.append({"feat": f.numpy(), "gt": gt})
        out["apply_boxes_%d" % im] = bb.numpy()
        out["apply_feat_%d" % im] = f.numpy()
        out["apply_gt_%d" % im] = gt
    with redirect_stdout(io.StringIO()):
        res = predictor.RegionPredictor(cfg3, det_models[:2])(boxes, feats)
    for im in range(2):
        out["apply_out_%d" % im] = res[im].bbox.numpy()
    # decode_boxes_detector (py_od_utils.py:247-274)
    utils = load_ref("src/py_od_utils.py", "ref_py_od_utils")
    bl = BoxList(torch.tensor(out["apply_boxes_0"]), (320, 240))
    pred = torch.randn(7, 12, generator=g) * 0.2
    out["decode_in"] = pred.numpy()
    out["decode_out"] = utils.decode_boxes_detector(bl, pred).numpy()
    np.savez_compressed(os.path.join(OUT, "rls_golden.npz"), **out)
    print("rls_golden.npz:", len(out), "arrays")
    return utils


# --------------------------------------------------------------------------- A6: minibootstrap
class RidgeClassifier:
    """Deterministic stand-in classifier (f64 ridge regression on the raw features).  Both the
    reference's OnlineRegionClassifier and the build's are driven with it, so the fixture pins the
    hard/easy-negative state machine, not FALKON."""

    def __init__(self, lam=1.0):
        self.lam = lam
        self.calls = []

    def train(self, X, y, sigma=None, lam=None):
        X64 = X.double()
        A = torch.cat([X64, torch.ones(len(X64), 1, dtype=torch.float64)], 1)
        w = torch.linalg.solve(A.T @ A + self.lam * torch.eye(A.shape[1], dtype=torch.float64), A.T @ y.double())
        self.calls.append(("train", int((y == 1).sum()), int((y == -1).sum())))
        return {"w": w}

    def predict(self, model, X, y=None):
        A = torch.cat([X.double(), torch.ones(len(X), 1, dtype=torch.float64)], 1)
        self.calls.append(("predict", len(X)))
        return (A @ model["w"]).float().view(-1, 1)


def make_bootstrap():
    install_boxlist()
    sys.path.insert(0, os.path.join(REF, "src"))
    sys.path.insert(0, os.path.join(REF, "src/modules/region-classifier"))
    sys.path.insert(0, os.path.join(REF, "src/modules"))
    classes = ["_background_", "a", "b", "c"]
    cfg_path = os.path.join(OUT, "cfg_bootstrap.yaml")
    write_cfg(cfg_path, classes, 10.0, extra_rpn=True)
    g = torch.Generator().manual_seed(5)
    D, C, ITER = 12, 3, 4
    mus = torch.randn(C, D, generator=g) * 1.5
    positives, negatives = [], []
    for c in range(C):
        npos = [30, 0, 18][c]  # class 1 has no positives -> model None
        positives.append(mus[c] + 0.5 * torch.randn(npos, D, generator=g) if npos else torch.empty((0, D)))
        negatives.append([mus[(c + 1 + (j % 2)) % C] * (0.3 + 0.2 * j) + 0.9 * torch.randn(40, D, generator=g)
                          for j in range(ITER)])
    stats = {"mean": torch.randn(D, generator=g) * 0.1, "std": torch.ones(D), "mean_norm": torch.tensor(4.0)}
    out = {"D": np.int64(D), "C": np.int64(C), "ITER": np.int64(ITER), "mean": stats["mean"].numpy(),
           "mean_norm": np.float64(4.0)}
    for c in range(C):
        out["pos_%d" % c] = positives[c].numpy()
        for j in range(ITER):
            out["neg_%d_%d" % (c, j)] = negatives[c][j].numpy()
    for variant, modname, retarget in (("cpu", "OnlineRegionClassifier", False), ("incore", "OnlineRegionClassifier_incore", True)):
        mod = load_ref("src/modules/region-classifier/%s.py" % modname, "ref_" + modname, retarget=retarget)
        clf = RidgeClassifier()
        pos = [p.clone() for p in positives]
        neg = [[b.clone() for b in nb] for nb in negatives]
        with redirect_stdout(io.StringIO()):
            orc = mod.OnlineRegionClassifier(clf, pos, neg, stats, cfg_path=cfg_path)
            models = orc.trainRegionClassifier()
        out[variant + "_calls"] = np.array(json.dumps(clf.calls))
        for c, m in enumerate(models):
            out["%s_model_%d_none" % (variant, c)] = np.bool_(m is None)
            if m is not None:
                out["%s_model_%d_w" % (variant, c)] = m["w"].numpy()
        out[variant + "_pos0_normalized"] = pos[0].numpy()
        if variant == "incore":
            # stand-alone scoring (testRegionClassifier, :182-219) with models that all exist
            clf2 = RidgeClassifier()
            full_models = [models[0], models[0], models[2]]
            tb = []
            for im in range(2):
                R = 6
                feat = torch.randn(R, D, generator=g).numpy()
                gt = np.zeros(R); gt[1] = 1
                bx = (torch.rand(R, 4, generator=g) * 100).numpy()
                tb.append({"boxes": bx, "feat": feat, "gt": gt, "img_size": (320, 240)})
                out["test_feat_%d" % im], out["test_gt_%d" % im], out["test_boxes_%d" % im] = feat, gt, bx
            with redirect_stdout(io.StringIO()):
                orc2 = mod.OnlineRegionClassifier(clf2, pos, neg, stats, cfg_path=cfg_path)
                preds = orc2.testRegionClassifier(full_models, tb)
            for im in range(2):
                out["test_scores_%d" % im] = preds[im].get_field("scores").numpy()
    np.savez_compressed(os.path.join(OUT, "bootstrap_golden.npz"), **out)
    print("bootstrap_golden.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- A1 / A2 / A10 helpers
def make_helpers(utils):
    out = {}
    g = torch.Generator().manual_seed(21)
    D, C = 16, 3
    positives = [torch.randn(25 + 5 * c, D, generator=g) + c for c in range(C)]
    positives[1] = torch.empty((0, D))
    negatives = [[torch.randn(30, D, generator=g) - c for _ in range(3)] for c in range(C)]
    for c in range(C):
        out["pos_%d" % c] = positives[c].numpy()
        for j in range(3):
            out["neg_%d_%d" % (c, j)] = negatives[c][j].numpy()
    torch.manual_seed(1234)
    with redirect_stdout(io.StringIO()):
        st = utils.computeFeatStatistics_torch(positives, negatives, num_samples=90, features_dim=D, cpu_tensor=True,
                                               pos_fraction=0.8)
    out["stats_mean"], out["stats_std"], out["stats_mean_norm"] = st["mean"].numpy(), st["std"].numpy(), st["mean_norm"].numpy()
    COXY = {"C": torch.randint(0, 4, (40, 1), generator=g).float(), "O": None, "X": torch.randn(40, D, generator=g),
            "Y": torch.randn(40, 4, generator=g)}
    out["coxy_C"], out["coxy_X"] = COXY["C"].numpy(), COXY["X"].numpy()
    norm = utils.normalize_COXY({k: (v.clone() if torch.is_tensor(v) else v) for k, v in COXY.items()}, st, cpu=True)
    out["coxy_X_normalized"] = norm["X"].numpy()
    pos_from = utils.load_positives_from_COXY({"C": COXY["C"][:, 0].clone(), "X": COXY["X"].clone()})
    out["pos_from_coxy_n"] = np.int64(len(pos_from))
    for i, p in enumerate(pos_from):
        out["pos_from_coxy_%d" % i] = p.numpy()
    torch.manual_seed(99)
    sh = utils.shuffle_negatives([[b.clone() for b in nb] for nb in negatives], batch_size=40, num_batches=3)
    for c in range(C):
        for j in range(3):
            out["shuf_%d_%d" % (c, j)] = sh[c][j].numpy()
    out["zscores"] = utils.zScores(positives[0].numpy(), st["mean"], st["mean_norm"]).numpy()
    # MyCenterSelector (A2)
    sel = load_ref("src/modules/region-classifier/MyCenterSelector.py", "MyCenterSelector", retarget=False)
    Xs = torch.randn(20, D, generator=g)
    Ys = torch.randn(20, 1, generator=g)
    idx = [3, 3, 7, 19, 0]
    out["sel_X"], out["sel_Y"], out["sel_idx"] = Xs.numpy(), Ys.numpy(), np.array(idx)
    out["sel_out_X"] = sel.MyCenterSelector(idx).select(Xs, None).numpy()
    xo, yo = sel.MyCenterSelector(idx).select(Xs, Ys)
    out["sel_out_X2"], out["sel_out_Y2"] = xo.numpy(), yo.numpy()
    np.savez_compressed(os.path.join(OUT, "helpers_golden.npz"), **out)
    print("helpers_golden.npz:", len(out), "arrays")
    return sel


# --------------------------------------------------------------------------- FALKONWrapper contract
def make_wrapper_contract():
    record = {"ctor": [], "fit": [], "predict": [], "select": []}

    class FakeKernel:
        def __init__(self, sigma):
            self.sigma = sigma

    class FakeOptions:
        def __init__(self, **kw):
            self.kw = kw

    class FakeFalkon:
        def __init__(self, **kw):
            record["ctor"].append({k: (v.kw if isinstance(v, FakeOptions) else
                                       ("GaussianKernel(%g)" % v.sigma if isinstance(v, FakeKernel) else
                                        (type(v).__name__ if k == "center_selection" else v))) for k, v in kw.items()})
            self.cs = kw["center_selection"]
            self.M = kw["M"]

        def fit(self, X, y):
            Z = self.cs.select(X, None)
            record["select"].append({"returns_tensor": bool(torch.is_tensor(Z)), "shape": list(Z.shape)})
            record["fit"].append({"X": list(X.shape), "y": list(y.shape), "y_dtype": str(y.dtype)})
            self.ny_points_ = Z
            self.alpha_ = torch.zeros(Z.shape[0], 1)

        def predict(self, X):
            record["predict"].append({"X": list(X.shape)})
            return torch.zeros(X.shape[0], 1)

    fk = types.ModuleType("falkon")
    fk.Falkon = FakeFalkon
    fk.InCoreFalkon = FakeFalkon
    fk.kernels = types.SimpleNamespace(GaussianKernel=FakeKernel)
    fo = types.ModuleType("falkon.options")
    fo.FalkonOptions = FakeOptions
    fo.__all__ = ["FalkonOptions"]
    sys.modules["falkon"], sys.modules["falkon.options"] = fk, fo
    cfg_path = os.path.join(OUT, "cfg_bootstrap.yaml")
    res = {}
    for variant, fname in (("cpu", "FALKONWrapper_with_centers_selection"), ("incore", "FALKONWrapper_with_centers_selection_incore")):
        for k in record:
            record[k] = []
        mod = load_ref("src/modules/region-classifier/%s.py" % fname, "ref_" + fname, retarget=False)
        w = mod.FALKONWrapper(cfg_path=cfg_path)
        g = torch.Generator().manual_seed(3)
        X = torch.randn(100, 8, generator=g)
        y = torch.cat([torch.ones(30), -torch.ones(70)])
        torch.manual_seed(77)
        idx = w.compute_indices_selection(y)
        torch.manual_seed(77)
        with redirect_stdout(io.StringIO()):
            model = w.train(X, y, sigma=7.0, lam=0.01)
            p = w.predict(model, X[:5])
        # few samples: everything is taken, in positives-then-negatives order
        y_small = torch.cat([torch.ones(5), -torch.ones(9)])
        idx_small = w.compute_indices_selection(y_small)
        y_one = torch.tensor([1.0])
        res[variant] = {"attrs": {"sigma": w.sigma, "lam": w.lam, "nyst_centers": w.nyst_centers,
                                  "maxiter": getattr(w, "maxiter", None)},
                        "indices_seed77": idx, "indices_small": idx_small,
                        "indices_single_is_int": isinstance(w.compute_indices_selection(y_one), int),
                        "record": json.loads(json.dumps(record)), "predict_shape": list(p.shape)}
    json.dump(res, open(os.path.join(OUT, "wrapper_contract.json"), "w"), indent=1, sort_keys=True)
    print("wrapper_contract.json written")


# --------------------------------------------------------------------------- A8 / A9: test-time heads
def _strip_imports(src):
    keep = []
    for line in src.splitlines():
        st = line.strip()
        if st.startswith(("from mrcnn_modified", "from maskrcnn_benchmark", "from .", "from falkon", "import falkon")):
            continue
        keep.append(line)
    return "\n".join(keep)


class _Registry(dict):
    def register(self, name):
        def deco(cls):
            self[name] = cls
            return cls
        return deco


class FakeKernel:
    """f64 Gaussian mmv: the heads only need *a* kernel object with falkon's mmv signature."""

    def __init__(self, sigma):
        self.sigma = sigma

    def mmv(self, X1, X2, v, out=None):
        d2 = torch.cdist(X1.double(), X2.double()) ** 2
        r = (torch.exp(-d2 / (2 * self.sigma ** 2)) @ v.double()).float()
        if out is not None:
            out.copy_(r)
            return out
        return r


class FakeModel:
    def __init__(self, ny, alpha, kernel):
        self.ny_points_, self.alpha_, self.M, self.kernel = ny, alpha, ny.shape[0], kernel

    def predict(self, X):
        return self.kernel.mmv(X, self.ny_points_, self.alpha_)


def make_heads():
    from torch import nn
    reg = types.SimpleNamespace(ROI_BOX_PREDICTOR=_Registry(), RPN_HEADS=_Registry())
    base = "src/modules/feature-extractor/mrcnn_modified/modeling/"
    ns_box = {"registry": reg, "__name__": "ref_roi_box_predictors"}
    src = _strip_imports(open(os.path.join(REF, base, "roi_heads/box_head/roi_box_predictors.py")).read())
    exec(compile(src.replace("'cuda'", "'cpu'"), "roi_box_predictors.py", "exec"), ns_box)
    ns_rpn = {"registry": reg, "__name__": "ref_rpn"}
    src = _strip_imports(open(os.path.join(REF, base, "rpn/rpn.py")).read())
    exec(compile(src.replace("'cuda'", "'cpu'"), "rpn.py", "exec"), ns_rpn)
    BoxPred, RPNHead = reg.ROI_BOX_PREDICTOR["OnlineDetectionBOXPredictor"], reg.RPN_HEADS["OnlineRPNHead"]

    g = torch.Generator().manual_seed(41)
    D, out = 16, {}
    kern = FakeKernel(6.0)

    def models(Ms):
        ms = []
        for m in Ms:
            ms.append(None if m == 0 else FakeModel(torch.randn(m, D, generator=g) * 3, torch.randn(m, 1, generator=g), kern))
        return ms

    def regressors(n, none_at):
        arr = np.empty((0))
        for j in range(n):
            if j in none_at:
                arr = np.append(arr, {"mu": None, "T": None, "T_inv": None, "Beta": None})
                continue
            A = torch.randn(4, 4, generator=g) * 0.3
            arr = np.append(arr, {"mu": torch.randn(4, generator=g) * 0.1, "T": torch.eye(4), "T_inv": A @ A.t() + torch.eye(4),
                                  "Beta": {str(k): {"weights": torch.randn(D + 1, generator=g) * 0.2, "losses": None} for k in range(4)}})
        return arr

    def dump_models(tag, ms, rs):
        out[tag + "_Ms"] = np.array([0 if m is None else m.M for m in ms])
        for i, m in enumerate(ms):
            if m is not None:
                out["%s_ny_%d" % (tag, i)], out["%s_alpha_%d" % (tag, i)] = m.ny_points_.numpy(), m.alpha_.numpy()
        out[tag + "_reg_none"] = np.array([r["Beta"] is None for r in rs])
        for j, r in enumerate(rs):
            if r["Beta"] is not None:
                out["%s_reg_W_%d" % (tag, j)] = np.stack([r["Beta"][str(k)]["weights"].numpy() for k in range(4)])
                out["%s_reg_Tinv_%d" % (tag, j)], out["%s_reg_mu_%d" % (tag, j)] = r["T_inv"].numpy(), r["mu"].numpy()

    stats = {"mean": torch.randn(D, generator=g) * 0.2, "mean_norm": torch.tensor(5.0)}
    out["stats_mean"], out["stats_mean_norm"], out["sigma"] = stats["mean"].numpy(), np.float64(5.0), np.float64(6.0)
    # ---- detector head
    cls_models, regs = models([7, 0, 12, 5]), regressors(4, {1})
    dump_models("det", cls_models, regs)
    x = torch.randn(9, D, 1, 1, generator=g) * 3
    out["det_x"] = x.view(9, D).numpy()
    for par in (True, False):
        for norm_reg in (False, True):
            h = BoxPred.__new__(BoxPred)
            nn.Module.__init__(h)
            h.avgpool = nn.AdaptiveAvgPool2d(1)
            h.parallel_inference, h.normalize_features_regressors, h.feat_size = par, norm_reg, D
            h.classifiers, h.regressors, h.stats = cls_models, regs, stats
            with torch.no_grad():
                sc, bb = h.forward(x)
            tag = "det_par%d_norm%d" % (par, norm_reg)
            out[tag + "_scores"], out[tag + "_bbox"] = sc.numpy(), bb.numpy()
    # ---- RPN head
    A_, H, W = 6, 4, 5
    rpn_models, rpn_regs = models([8, 8, 0, 6, 8, 8]), regressors(A_, {4})
    dump_models("rpn", rpn_models, rpn_regs)
    t_in = torch.randn(1, D, H, W, generator=g)
    for par in (True, False):
        h = RPNHead.__new__(RPNHead)
        nn.Module.__init__(h)
        h.conv = nn.Conv2d(D, D, 3, 1, 1)
        torch.manual_seed(8)
        nn.init.normal_(h.conv.weight, std=0.2)
        nn.init.constant_(h.conv.bias, 0.1)
        h.parallel_inference, h.feat_size, h.height, h.width, h.num_clss, h.area = par, None, None, None, A_, None
        h.classifiers, h.regressors, h.stats = rpn_models, rpn_regs, stats
        with torch.no_grad():
            out["rpn_act"] = torch.relu(h.conv(t_in)).numpy()
            logits, bbox = h.forward([t_in])
        out["rpn_par%d_logits" % par], out["rpn_par%d_bbox" % par] = logits[0].numpy(), bbox[0].numpy()
    # ---- mask head (roi_mask_predictors.py:37-99)
    reg.ROI_MASK_PREDICTOR = _Registry()
    ns_mask = {"registry": reg, "Conv2d": nn.Conv2d, "ConvTranspose2d": nn.ConvTranspose2d, "__name__": "ref_roi_mask_predictors"}
    src = _strip_imports(open(os.path.join(REF, base, "roi_heads/mask_head/roi_mask_predictors.py")).read())
    exec(compile(src.replace("'cuda'", "'cpu'"), "roi_mask_predictors.py", "exec"), ns_mask)
    MaskPred = reg.ROI_MASK_PREDICTOR["MaskRCNNC4Predictor"]
    mask_models = models([9, 0, 7])
    dump_models("mask", mask_models, regressors(0, set()))
    xin = torch.randn(3, 24, 3, 3, generator=g)
    for par in (True, False):
        h = MaskPred.__new__(MaskPred)
        nn.Module.__init__(h)
        torch.manual_seed(5)
        h.conv5_mask = nn.ConvTranspose2d(24, D, 2, 2, 0)
        h.parallel_inference, h.feat_size = par, D
        h.classifiers, h.stats = mask_models, stats
        with torch.no_grad():
            out["mask_act"] = torch.relu(h.conv5_mask(xin)).numpy()
            out["mask_par%d_out" % par] = h.forward(xin).numpy()
    np.savez_compressed(os.path.join(OUT, "heads_golden.npz"), **out)
    print("heads_golden.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- A11: detector harvesting
class ResizableBoxList(BoxList):
    def resize(self, size):
        rw, rh = size[0] / self.size[0], size[1] / self.size[1]
        b = self.bbox.clone().float()
        b[:, 0::2] *= rw
        b[:, 1::2] *= rh
        return ResizableBoxList(b, size, self.mode)


def make_harvest():
    from torch import nn
    base = "src/modules/feature-extractor/mrcnn_modified/"
    ev = load_ref(base + "utils/evaluations.py", "ref_evaluations")
    src = _strip_imports(open(os.path.join(REF, base, "modeling/roi_heads/box_head/box_head_getProposals.py")).read())
    ns = {"compute_overlap_torch": ev.compute_overlap_torch, "__name__": "ref_box_head_getProposals"}
    exec(compile(src.replace("'cuda'", "'cpu'"), "box_head_getProposals.py", "exec"), ns)
    Head = ns["ROIBoxHead"]
    D, C, ITER, BS, NIMG = 10, 3, 3, 12, 6
    out = {"D": np.int64(D), "C": np.int64(C), "ITER": np.int64(ITER), "BS": np.int64(BS), "NIMG": np.int64(NIMG)}
    g = torch.Generator().manual_seed(77)
    images = []
    for im in range(NIMG):
        G = [2, 1, 0, 2, 1, 3][im]
        labels = [[1, 3], [2], [], [3, 3], [1], [2, 1, 3]][im]
        gt = torch.rand(G, 2, generator=g) * 150
        gt = torch.cat([gt, gt + 30 + torch.rand(G, 2, generator=g) * 80], 1)
        R = 25
        jit = []
        for k in range(R):                        # proposals: jittered copies of the gts + random boxes
            if G and k % 2 == 0:
                j = gt[k % G] + (torch.rand(4, generator=g) - 0.5) * 24
            else:
                a = torch.rand(2, generator=g) * 200
                j = torch.cat([a, a + 10 + torch.rand(2, generator=g) * 90])
            jit.append(j)
        prop = torch.stack(jit)
        prop[3] = torch.tensor([-5.0, -3.0, 400.0, 300.0])  # sticks out of the 320 x 240 image
        allp = torch.cat([gt, prop])
        x = torch.randn(G + R, D, generator=g)
        images.append((x, allp, gt, labels))
        out["x_%d" % im], out["prop_%d" % im], out["gt_%d" % im], out["labels_%d" % im] = x.numpy(), allp.numpy(), gt.numpy(), np.array(labels, dtype=np.int64)
    for shuffle in (False, True):
        h = Head.__new__(Head)
        nn.Module.__init__(h)
        dcfg = types.SimpleNamespace(NUM_CLASSES=C, ITERATIONS=ITER, BATCH_SIZE=BS, EXTRACT_ONLY_GT_POSITIVES=True,
                                     SHUFFLE_NEGATIVES=shuffle, NEG_IOU_THRESH=0.3, FEATURES_DEVICE="cpu")
        h.cfg = types.SimpleNamespace(MINIBOOTSTRAP=types.SimpleNamespace(DETECTOR=dcfg), DEMO=types.SimpleNamespace(INCREMENTAL_TRAIN=False),
                                      REGRESSORS=types.SimpleNamespace(MIN_OVERLAP=0.6), NUM_IMAGES=NIMG)
        h.training_device = "cpu"
        h.save_features = False
        h.avgpool = nn.AdaptiveAvgPool2d(1)
        h.feature_extractor = types.SimpleNamespace(out_channels=D)
        h.initialize_online_detection_params()
        torch.manual_seed(123)
        for (x, allp, gt, labels) in images:
            h.feature_extractor = lambda features, proposals, _x=x: _x.view(_x.shape[0], D, 1, 1)
            h.feature_extractor.out_channels = D
            gl = torch.tensor(labels, dtype=torch.uint8).view(-1, 1) if labels else None
            h.forward_train(None, [ResizableBoxList(allp.clone(), (320, 240))], gt_bbox=ResizableBoxList(gt.clone(), (320, 240)),
                            gt_label=gl, img_size=[320, 240], gt_labels_list=labels)
        tag = "shuf" if shuffle else "fill"
        out[tag + "_C"] = torch.cat(h.C).numpy()
        out[tag + "_X"], out[tag + "_Y"] = torch.cat(h.X).numpy(), torch.cat(h.Y).numpy()
        for c in range(C):
            out["%s_pos_%d" % (tag, c)] = torch.cat(h.positives[c]).numpy()
            if shuffle:
                out["%s_neg_%d" % (tag, c)] = torch.cat(h.negatives[c]).numpy()
            else:
                for b in range(ITER):
                    out["%s_neg_%d_%d" % (tag, c, b)] = h.negatives[c][b].numpy()
        if not shuffle:
            out["fill_still_to_complete"] = np.array(h.still_to_complete, dtype=np.int64)
            x, allp, gt, labels = images[0]
            h.feature_extractor = lambda features, proposals, _x=x: _x.view(_x.shape[0], D, 1, 1)
            h.forward_test(None, [ResizableBoxList(allp.clone(), (320, 240))], gt_bbox=ResizableBoxList(gt.clone(), (320, 240)),
                           gt_label=torch.tensor(labels, dtype=torch.uint8).view(-1, 1), img_size=[320, 240], gt_labels_list=labels)
            tb = h.test_boxes[0]
            out["test_boxes"], out["test_feat"], out["test_gt"] = tb["boxes"], tb["feat"], tb["gt"]
    # IoU helper on its own
    out["iou_gt"], out["iou_prop"] = images[0][2][0].numpy(), images[0][1].numpy()
    out["iou_out"] = ev.compute_overlap_torch(images[0][2][0], images[0][1]).numpy()
    np.savez_compressed(os.path.join(OUT, "harvest_golden.npz"), **out)
    print("harvest_golden.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- A12: RPN harvesting
class FieldBoxList:
    """Stand-in for maskrcnn_benchmark's BoxList with what RPNModule.forward touches: fields that
    follow indexing, copy_with_fields, resize, size."""

    def __init__(self, bbox, image_size, mode="xyxy"):
        self.bbox, self.size, self.mode, self.extra_fields = bbox, image_size, mode, {}

    def add_field(self, k, v):
        self.extra_fields[k] = v

    def get_field(self, k):
        return self.extra_fields[k]

    def fields(self):
        return list(self.extra_fields)

    def copy_with_fields(self, fields):
        b = FieldBoxList(self.bbox, self.size, self.mode)
        for f in fields:
            b.add_field(f, self.get_field(f))
        return b

    def resize(self, size):
        rw, rh = size[0] / self.size[0], size[1] / self.size[1]
        b = self.bbox.clone().float()
        b[:, 0::2] *= rw
        b[:, 1::2] *= rh
        return FieldBoxList(b, size, self.mode)

    def __getitem__(self, item):
        b = FieldBoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            b.add_field(k, v[item])
        return b

    def __len__(self):
        return self.bbox.shape[0]


def ref_boxlist_iou(b1, b2):
    """maskrcnn_benchmark.structures.boxlist_ops.boxlist_iou (TO_REMOVE = 1), restated."""
    a1 = (b1.bbox[:, 2] - b1.bbox[:, 0] + 1) * (b1.bbox[:, 3] - b1.bbox[:, 1] + 1)
    a2 = (b2.bbox[:, 2] - b2.bbox[:, 0] + 1) * (b2.bbox[:, 3] - b2.bbox[:, 1] + 1)
    lt = torch.max(b1.bbox[:, None, :2], b2.bbox[:, :2])
    rb = torch.min(b1.bbox[:, None, 2:], b2.bbox[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def ref_cat_boxlist(bl):
    out = FieldBoxList(torch.cat([b.bbox for b in bl]), bl[0].size, bl[0].mode)
    for f in bl[0].fields():
        out.add_field(f, torch.cat([b.get_field(f) for b in bl]))
    return out


def make_rpn_harvest():
    from torch import nn
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), os.pardir, "online-detection_amd"))
    from odx.extract import cell_anchors, grid_anchors
    base = "src/modules/feature-extractor/mrcnn_modified/"
    src = _strip_imports(open(os.path.join(REF, base, "modeling/rpn/rpn_getProposals.py")).read())
    reg = types.SimpleNamespace(RPN_HEADS=_Registry())
    ns = {"registry": reg, "boxlist_iou": ref_boxlist_iou, "cat_boxlist": ref_cat_boxlist, "BoxList": FieldBoxList,
          "__name__": "ref_rpn_getProposals"}
    exec(compile(src.replace("'cuda'", "'cpu'"), "rpn_getProposals.py", "exec"), ns)
    RPN = ns["RPNModule"]
    D, A, H, W, ITER, BS, NIMG = 6, 15, 9, 12, 2, 10, 4
    img = (W * 16, H * 16)
    anchors_all = grid_anchors(H, W, 16, cell_anchors(16))
    vis = (anchors_all[:, 0] >= 0) & (anchors_all[:, 1] >= 0) & (anchors_all[:, 2] < img[0]) & (anchors_all[:, 3] < img[1])
    out = {"D": np.int64(D), "A": np.int64(A), "H": np.int64(H), "W": np.int64(W), "ITER": np.int64(ITER), "BS": np.int64(BS),
           "NIMG": np.int64(NIMG), "anchors": anchors_all.numpy()}
    g = torch.Generator().manual_seed(91)
    images = []
    for im in range(NIMG):
        G = [1, 2, 3, 2][im]
        xy = torch.rand(G, 2, generator=g) * torch.tensor([img[0] * 0.5, img[1] * 0.5])
        gt = torch.cat([xy, xy + 24 + torch.rand(G, 2, generator=g) * torch.tensor([img[0] * 0.4, img[1] * 0.4])], 1)
        gt[0] = torch.tensor([[31.0, 31.0, 96.0, 96.0], [20.0, 40.0, 111.0, 85.0], [8.0, 8.0, 71.0, 135.0], [40.0, 30.0, 167.0, 93.0]][im])
        t = torch.randn(D, H, W, generator=g)
        images.append((t, gt))
        out["t_%d" % im], out["gt_%d" % im] = t.numpy(), gt.numpy()
    for shuffle in (False, True):
        m = RPN.__new__(RPN)
        nn.Module.__init__(m)
        rcfg = types.SimpleNamespace(NUM_CLASSES=A, ITERATIONS=ITER, BATCH_SIZE=BS, NEG_IOU_THRESH=0.3, POS_IOU_THRESH=0.7,
                                     SHUFFLE_NEGATIVES=shuffle, FEATURES_DEVICE="cpu")
        m.cfg = types.SimpleNamespace(MINIBOOTSTRAP=types.SimpleNamespace(RPN=rcfg), DEMO=types.SimpleNamespace(INCREMENTAL_TRAIN=False),
                                      NUM_IMAGES=NIMG)
        m.save_features = False
        m.prev_classifiers = m.prev_feature_ids = None

        def anchor_generator(images_, features_):
            b = FieldBoxList(anchors_all.clone(), img)
            b.add_field("visibility", vis.clone())
            return [[b]]
        m.anchor_generator = anchor_generator
        m.initialize_online_rpn_params()
        torch.manual_seed(321)
        with redirect_stdout(io.StringIO()):
            for (t, gt) in images:
                m.head = lambda feats, _t=t: [_t.unsqueeze(0)]
                m.forward(None, None, gt_bbox=FieldBoxList(gt.clone(), img), img_size=None)
        tag = "shuf" if shuffle else "fill"
        out[tag + "_C"], out[tag + "_X"], out[tag + "_Y"] = torch.cat(m.C).numpy(), torch.cat(m.X).numpy(), torch.cat(m.Y).numpy()
        for c in range(A):
            out["%s_pos_%d" % (tag, c)] = torch.cat(m.positives[c]).numpy()
            if shuffle:
                out["%s_neg_%d" % (tag, c)] = torch.cat(m.negatives[c]).numpy()
            else:
                for b in range(ITER):
                    out["%s_neg_%d_%d" % (tag, c, b)] = m.negatives[c][b].numpy()
        if not shuffle:
            out["fill_still_to_complete"] = np.array(m.still_to_complete, dtype=np.int64)
            out["fill_anchors_ids"] = np.array(m.anchors_ids, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "rpn_harvest_golden.npz"), **out)
    print("rpn_harvest_golden.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- A13: mask harvesting
def make_mask_harvest():
    from torch import nn
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), os.pardir, "online-detection_amd"))
    from odx.harvest import project_masks_on_boxes as my_project
    base = "src/modules/feature-extractor/mrcnn_modified/"
    src = _strip_imports(open(os.path.join(REF, base, "modeling/roi_heads/mask_head/mask_head_getProposals.py")).read())
    ns = {"BoxList": FieldBoxList, "__name__": "ref_mask_head_getProposals"}
    exec(compile(src.replace("'cuda'", "'cpu'"), "mask_head_getProposals.py", "exec"), ns)
    Head = ns["ROIMaskHead"]

    class OneMask:                      # what project_masks_on_boxes needs from a SegmentationMask item
        def __init__(self, m):
            self.m = m

        def crop(self, box):
            self.box = box
            return self

        def resize(self, size):
            self.M = size[0]
            return self

        def get_mask_tensor(self):
            return my_project(self.m[None], self.box[None], self.M)[0]

    class Masks:
        def __init__(self, masks, size):
            self.masks, self.size = masks, size

        def __iter__(self):
            return iter([OneMask(m) for m in self.masks])

    D, C, Cin, S = 8, 3, 12, 6
    g = torch.Generator().manual_seed(55)
    out = {"D": np.int64(D), "C": np.int64(C), "S": np.int64(S)}
    h = Head.__new__(Head)
    nn.Module.__init__(h)
    h.cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(ROI_MASK_HEAD={"FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor",
                                                                               "SHARE_BOX_FEATURE_EXTRACTOR": True}),
                                  MINIBOOTSTRAP=types.SimpleNamespace(DETECTOR=types.SimpleNamespace(NUM_CLASSES=C)),
                                  SEGMENTATION=types.SimpleNamespace(BATCH_SIZE=60, SAMPLING_FACTOR=0.3, FEATURES_DEVICE="cpu"))
    h.cfg.MODEL.ROI_MASK_HEAD = type("D", (dict,), {"SHARE_BOX_FEATURE_EXTRACTOR": True})({"FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor"})
    h.save_features, h.training_device = False, "cpu"
    torch.manual_seed(9)
    conv = nn.ConvTranspose2d(Cin, D, 2, 2, 0)
    h.predictor = types.SimpleNamespace(conv5_mask=conv, mask_fcn_logits=types.SimpleNamespace(in_channels=D))
    h.initialize_online_segmentation_params()
    torch.manual_seed(77)
    for im in range(3):
        G = [2, 1, 2][im]
        labels = [[1, 3], [2], [3, 1]][im]
        Himg, Wimg = 60, 80
        masks = torch.zeros(G, Himg, Wimg, dtype=torch.uint8)
        boxes = torch.zeros(G, 4)
        for k in range(G):
            x1, y1 = int(torch.randint(0, 30, (1,), generator=g)), int(torch.randint(0, 20, (1,), generator=g))
            w, hh = int(torch.randint(12, 40, (1,), generator=g)), int(torch.randint(10, 30, (1,), generator=g))
            boxes[k] = torch.tensor([x1, y1, x1 + w, y1 + hh], dtype=torch.float32)
            yy, xx = torch.meshgrid(torch.arange(Himg), torch.arange(Wimg), indexing="ij")
            masks[k] = ((((xx - (x1 + w / 2)) / (w / 2)) ** 2 + ((yy - (y1 + hh / 2)) / (hh / 2)) ** 2) <= 1).to(torch.uint8)
        feats = torch.randn(G + 4, Cin, S // 2, S // 2, generator=g)        # gt rows first, then other RoIs
        gtb = FieldBoxList(boxes.clone(), (Wimg, Himg))
        gtb.convert = lambda mode, _b=gtb: _b
        gtb.add_field("masks", Masks(masks, (Wimg, Himg)))
        with torch.no_grad():
            h.forward(feats, None, labels, gtb)
            out["act_%d" % im] = torch.relu(conv(feats[:G])).numpy()
        out["masks_%d" % im], out["boxes_%d" % im], out["labels_%d" % im] = masks.numpy(), boxes.numpy(), np.array(labels)
    for c in range(C):
        out["pos_%d" % c] = torch.cat(h.positives[c]).detach().numpy()
        out["neg_%d" % c] = torch.cat(h.negatives[c]).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "mask_harvest_golden.npz"), **out)
    print("mask_harvest_golden.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- f3: post-processing + VOC evaluation
def ref_nms_keep(boxes, scores, thresh):
    """maskrcnn_benchmark layers.nms (absent dependency, restated): greedy, +1 areas, suppress on iou > thresh."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].double()
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    dead = torch.zeros(len(b), dtype=torch.bool)
    keep = []
    for i in range(len(b)):
        if dead[i]:
            continue
        keep.append(i)
        w = (torch.min(b[i, 2], b[i + 1:, 2]) - torch.max(b[i, 0], b[i + 1:, 0]) + 1).clamp(min=0)
        h = (torch.min(b[i, 3], b[i + 1:, 3]) - torch.max(b[i, 1], b[i + 1:, 1]) + 1).clamp(min=0)
        iou = w * h / (area[i] + area[i + 1:] - w * h)
        dead[i + 1:] |= iou > thresh
    return order[torch.tensor(keep, dtype=torch.int64)]


def ref_boxlist_nms(boxlist, thresh, max_proposals=-1, score_field="scores"):
    return boxlist[ref_nms_keep(boxlist.bbox, boxlist.get_field(score_field), thresh)]


class _ClipBoxList(FieldBoxList):
    def clip_to_image(self, remove_empty=True):
        self.bbox[:, 0].clamp_(min=0, max=self.size[0] - 1)
        self.bbox[:, 1].clamp_(min=0, max=self.size[1] - 1)
        self.bbox[:, 2].clamp_(min=0, max=self.size[0] - 1)
        self.bbox[:, 3].clamp_(min=0, max=self.size[1] - 1)
        return self

    def resize(self, size):
        r = FieldBoxList.resize(self, size)
        return _ClipBoxList(r.bbox, r.size, r.mode)

    def __getitem__(self, item):
        b = _ClipBoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            b.add_field(k, v[item])
        return b


class _PostProcessorBase:
    """box_head.inference.PostProcessor: only what OnlineDetectionPostProcessor inherits."""

    def __init__(self, score_thresh, nms, detections_per_img):
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img

    def prepare_boxlist(self, boxes, scores, image_shape):
        b = _ClipBoxList(boxes.reshape(-1, 4), image_shape)
        b.add_field("scores", scores.reshape(-1))
        return b


def make_postprocess():
    utils = load_ref("src/py_od_utils.py", "py_od_utils")
    src = _strip_imports(open(os.path.join(REF, "src/modules/accuracy-evaluator/OnlineDetectionPostProcessor.py")).read())
    src = src.replace("from py_od_utils import decode_boxes_detector", "")

    def cat(bl):
        out = _ClipBoxList(torch.cat([b.bbox for b in bl]), bl[0].size, bl[0].mode)
        for f in bl[0].fields():
            out.add_field(f, torch.cat([b.get_field(f) for b in bl]))
        return out
    ns = {"PostProcessor": _PostProcessorBase, "BoxList": _ClipBoxList, "boxlist_nms": ref_boxlist_nms, "cat_boxlist": cat,
          "decode_boxes_detector": utils.decode_boxes_detector, "__name__": "ref_postproc"}
    exec(compile(src.replace("'cuda'", "'cpu'"), "OnlineDetectionPostProcessor.py", "exec"), ns)
    PP = ns["OnlineDetectionPostProcessor"]
    out = {}
    g = torch.Generator().manual_seed(4242)
    cases = [dict(R=60, C=4, thr=-2.0, nms=0.3, dpi=100, psz=(320, 240), isz=(320, 240)),
             dict(R=300, C=6, thr=-2.0, nms=0.3, dpi=100, psz=(640, 480), isz=(320, 240)),
             dict(R=80, C=3, thr=0.2, nms=0.5, dpi=12, psz=(200, 160), isz=(400, 320)),
             dict(R=25, C=3, thr=5.0, nms=0.3, dpi=100, psz=(200, 160), isz=(200, 160))]
    out["n_cases"] = np.int64(len(cases))
    for ci, c in enumerate(cases):
        R, C = c["R"], c["C"] + 1
        ctr = torch.rand(6, 2, generator=g) * torch.tensor([c["psz"][0] * 0.7, c["psz"][1] * 0.7])
        pick = torch.randint(0, 6, (R,), generator=g)
        xy = (ctr[pick] + torch.randn(R, 2, generator=g) * 6).clamp(min=0)
        wh = 30 + torch.rand(R, 2, generator=g) * 60
        props = torch.cat([xy, torch.min(xy + wh, torch.tensor([c["psz"][0] - 1.0, c["psz"][1] - 1.0]))], 1)
        scores = torch.randn(R, C, generator=g)
        deltas = torch.randn(R, 4 * C, generator=g) * 0.15
        pp = PP(c["thr"], c["nms"], c["dpi"])
        res = pp.forward((scores.clone(), deltas.clone()), [_ClipBoxList(props.clone(), c["psz"])], C, c["isz"])
        out["c%d_meta" % ci] = np.array([R, C, c["dpi"], c["psz"][0], c["psz"][1], c["isz"][0], c["isz"][1]], dtype=np.int64)
        out["c%d_thr" % ci] = np.array([c["thr"], c["nms"]], dtype=np.float64)
        out["c%d_props" % ci], out["c%d_scores" % ci], out["c%d_deltas" % ci] = props.numpy(), scores.numpy(), deltas.numpy()
        out["c%d_boxes" % ci] = res.bbox.numpy()
        out["c%d_out_scores" % ci] = res.get_field("scores").numpy()
        out["c%d_labels" % ci] = res.get_field("labels").numpy()
    np.savez_compressed(os.path.join(OUT, "postprocess_golden.npz"), **out)
    print("postprocess_golden.npz:", len(out), "arrays", [out["c%d_labels" % i].shape[0] for i in range(len(cases))])


def make_eval():
    base = "src/modules/feature-extractor/mrcnn_modified/"
    src = open(os.path.join(REF, base, "data/datasets/evaluation/icubworld/icw_eval.py")).read()
    keep = []
    for line in src.splitlines():
        if line.strip().startswith(("from maskrcnn_benchmark", "from mrcnn_modified", "from py_od_utils", "import cv2")):
            continue
        keep.append(line)
    ns = {"BoxList": FieldBoxList, "boxlist_iou": lambda a, b: ref_boxlist_iou(_t(a), _t(b)), "__name__": "ref_icw_eval"}

    def _t(b):
        b.bbox = torch.as_tensor(b.bbox)
        return b
    with redirect_stdout(io.StringIO()):
        exec(compile("\n".join(keep).replace("'cuda'", "'cpu'"), "icw_eval.py", "exec"), ns)
    rng = np.random.RandomState(77)
    out = {}
    NIMG, NCLS = 12, 5
    preds, gts = [], []
    for im in range(NIMG):
        G = rng.randint(0, 4)
        gxy = rng.rand(G, 2) * 200
        gb = np.concatenate([gxy, gxy + 30 + rng.rand(G, 2) * 80], 1).astype(np.float32)
        gl = rng.randint(1, NCLS, size=G)
        gd = rng.rand(G) < 0.2
        P = rng.randint(0, 9)
        rows = []
        for _ in range(P):
            if G and rng.rand() < 0.7:
                k = rng.randint(G)
                rows.append((gb[k] + rng.randn(4) * rng.choice([2.0, 12.0]), gl[k] if rng.rand() < 0.8 else rng.randint(1, NCLS)))
            else:
                xy = rng.rand(2) * 200
                rows.append((np.concatenate([xy, xy + 30 + rng.rand(2) * 80]), rng.randint(1, NCLS + 1)))
        pb = np.array([r[0] for r in rows], dtype=np.float32).reshape(-1, 4)
        pl = np.array([r[1] for r in rows], dtype=np.int64)
        ps = rng.randn(P).astype(np.float32)
        if P > 3:
            ps[1] = ps[0]          # tie
        for k, v in (("pb", pb), ("pl", pl), ("ps", ps), ("gb", gb), ("gl", gl.astype(np.int64)), ("gd", gd)):
            out["%s_%d" % (k, im)] = v
        p = FieldBoxList(torch.from_numpy(pb), (320, 240))
        p.add_field("labels", torch.from_numpy(pl))
        p.add_field("scores", torch.from_numpy(ps))
        q = FieldBoxList(torch.from_numpy(gb), (320, 240))
        q.add_field("labels", torch.from_numpy(gl.astype(np.int64)))
        q.add_field("difficult", torch.from_numpy(gd))
        preds.append(p)
        gts.append(q)
    out["NIMG"] = np.int64(NIMG)
    with np.errstate(all="ignore"):
        for thr in (0.5, 0.7):
            for m07 in (True, False):
                r = ns["eval_detection_icw"](preds, gts, iou_thresh=thr, use_07_metric=m07)
                tag = "iou%02d_%s" % (int(thr * 100), "voc07" if m07 else "area")
                out[tag + "_ap"], out[tag + "_map"] = np.asarray(r["ap"], dtype=np.float64), np.float64(r["map"])
        prec, rec = ns["calc_detection_icw_prec_rec"](gts, preds, 0.5)
    for l in range(len(prec)):
        out["prec_%d" % l] = np.array([]) if prec[l] is None else np.asarray(prec[l], dtype=np.float64)
        out["rec_%d" % l] = np.array([]) if rec[l] is None else np.asarray(rec[l], dtype=np.float64)
        out["has_%d" % l] = np.array([prec[l] is not None, rec[l] is not None])
    out["n_cls"] = np.int64(len(prec))
    np.savez_compressed(os.path.join(OUT, "eval_golden.npz"), **out)
    print("eval_golden.npz:", len(out), "arrays; mAP@0.5 voc07 =", out["iou50_voc07_map"])


# --------------------------------------------------------------------------- f2: on-disk feature caches
def _flush_like_reference(model_part, result_dir, sub, classes, feat_size, with_positives=True):
    """The end-of-harvest flush (extract_features_detector.py:193-248 / extract_features_RPN.py:171-198), restated:
    it lives inside train() methods that build datasets and models and cannot be executed here."""
    d = os.path.join(result_dir, sub)
    for clss in classes:
        for batch in range(len(model_part.negatives[clss])):
            if model_part.negatives[clss][batch].size()[0] > 0:
                torch.save(model_part.negatives[clss][batch], os.path.join(d, "negatives_cl_{}_batch_{}".format(clss, batch)))
        if with_positives:
            if model_part.positives[clss][0].size()[0] == 0 and len(model_part.positives[clss]) == 1:
                torch.save(torch.empty((0, feat_size)), os.path.join(d, "positives_cl_{}_batch_{}".format(clss, 0)))
            else:
                for batch in range(len(model_part.positives[clss])):
                    if model_part.positives[clss][batch].size()[0] > 0:
                        torch.save(model_part.positives[clss][batch], os.path.join(d, "positives_cl_{}_batch_{}".format(clss, batch)))
    for i in range(len(model_part.X)):
        if model_part.X[i].size()[0] > 0:
            torch.save(model_part.X[i], os.path.join(d, "reg_x_batch_{}".format(i)))
            torch.save(model_part.C[i], os.path.join(d, "reg_c_batch_{}".format(i)))
            torch.save(model_part.Y[i], os.path.join(d, "reg_y_batch_{}".format(i)))


def _collect(result_dir, sub, tag, out):
    d = os.path.join(result_dir, sub)
    names = sorted(os.listdir(d))
    for n in names:
        out["%s/%s" % (tag, n)] = torch.load(os.path.join(d, n)).numpy()
    out["%s/__files__" % tag] = np.array(names)


def make_feature_cache():
    """Runs the reference harvesters with save_features=True on the inputs of harvest_golden / rpn_harvest_golden
    (same seeds) and records every file they write."""
    import tempfile
    from torch import nn
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), os.pardir, "online-detection_amd"))
    from odx.extract import cell_anchors, grid_anchors
    base = "src/modules/feature-extractor/mrcnn_modified/"
    out = {}
    # ---- detector
    ev = load_ref(base + "utils/evaluations.py", "ref_evaluations")
    src = _strip_imports(open(os.path.join(REF, base, "modeling/roi_heads/box_head/box_head_getProposals.py")).read())
    ns = {"compute_overlap_torch": ev.compute_overlap_torch, "__name__": "ref_box_head_getProposals"}
    exec(compile(src.replace("'cuda'", "'cpu'"), "box_head_getProposals.py", "exec"), ns)
    Head = ns["ROIBoxHead"]
    hg = np.load(os.path.join(OUT, "harvest_golden.npz"))
    D, C, ITER, BS, NIMG = (int(hg[k]) for k in ("D", "C", "ITER", "BS", "NIMG"))
    for shuffle in (False, True):
        with tempfile.TemporaryDirectory() as tmp:
            os.mkdir(os.path.join(tmp, "features_detector"))
            h = Head.__new__(Head)
            nn.Module.__init__(h)
            dcfg = types.SimpleNamespace(NUM_CLASSES=C, ITERATIONS=ITER, BATCH_SIZE=BS, EXTRACT_ONLY_GT_POSITIVES=True,
                                         SHUFFLE_NEGATIVES=shuffle, NEG_IOU_THRESH=0.3, FEATURES_DEVICE="cpu")
            h.cfg = types.SimpleNamespace(MINIBOOTSTRAP=types.SimpleNamespace(DETECTOR=dcfg), DEMO=types.SimpleNamespace(INCREMENTAL_TRAIN=False),
                                          REGRESSORS=types.SimpleNamespace(MIN_OVERLAP=0.6), NUM_IMAGES=NIMG)
            h.training_device = "cpu"
            h.save_features = True
            h.avgpool = nn.AdaptiveAvgPool2d(1)
            h.feature_extractor = types.SimpleNamespace(out_channels=D)
            h.initialize_online_detection_params()
            torch.manual_seed(123)
            for im in range(NIMG):
                x, allp, gt = (torch.from_numpy(hg["%s_%d" % (k, im)]) for k in ("x", "prop", "gt"))
                labels = hg["labels_%d" % im].tolist()
                h.feature_extractor = lambda features, proposals, _x=x: _x.view(_x.shape[0], D, 1, 1)
                h.feature_extractor.out_channels = D
                gl = torch.tensor(labels, dtype=torch.uint8).view(-1, 1) if labels else None
                h.forward_train(None, [ResizableBoxList(allp.clone(), (320, 240))], gt_bbox=ResizableBoxList(gt.clone(), (320, 240)),
                                gt_label=gl, img_size=[320, 240], gt_labels_list=labels, result_dir=tmp)
            _flush_like_reference(h, tmp, "features_detector", range(C), D)
            _collect(tmp, "features_detector", "det_shuf" if shuffle else "det_fill", out)
    # ---- RPN
    src = _strip_imports(open(os.path.join(REF, base, "modeling/rpn/rpn_getProposals.py")).read())
    reg = types.SimpleNamespace(RPN_HEADS=_Registry())
    ns = {"registry": reg, "boxlist_iou": ref_boxlist_iou, "cat_boxlist": ref_cat_boxlist, "BoxList": FieldBoxList,
          "__name__": "ref_rpn_getProposals"}
    exec(compile(src.replace("'cuda'", "'cpu'"), "rpn_getProposals.py", "exec"), ns)
    RPN = ns["RPNModule"]
    rg = np.load(os.path.join(OUT, "rpn_harvest_golden.npz"))
    D, A, H, W, ITER, BS, NIMG = (int(rg[k]) for k in ("D", "A", "H", "W", "ITER", "BS", "NIMG"))
    img = (W * 16, H * 16)
    anchors_all = grid_anchors(H, W, 16, cell_anchors(16))
    vis = (anchors_all[:, 0] >= 0) & (anchors_all[:, 1] >= 0) & (anchors_all[:, 2] < img[0]) & (anchors_all[:, 3] < img[1])
    for shuffle in (False, True):
        with tempfile.TemporaryDirectory() as tmp:
            os.mkdir(os.path.join(tmp, "features_RPN"))
            m = RPN.__new__(RPN)
            nn.Module.__init__(m)
            rcfg = types.SimpleNamespace(NUM_CLASSES=A, ITERATIONS=ITER, BATCH_SIZE=BS, NEG_IOU_THRESH=0.3, POS_IOU_THRESH=0.7,
                                         SHUFFLE_NEGATIVES=shuffle, FEATURES_DEVICE="cpu")
            m.cfg = types.SimpleNamespace(MINIBOOTSTRAP=types.SimpleNamespace(RPN=rcfg), DEMO=types.SimpleNamespace(INCREMENTAL_TRAIN=False),
                                          NUM_IMAGES=NIMG)
            m.save_features = True
            m.prev_classifiers = m.prev_feature_ids = None

            def anchor_generator(images_, features_):
                b = FieldBoxList(anchors_all.clone(), img)
                b.add_field("visibility", vis.clone())
                return [[b]]
            m.anchor_generator = anchor_generator
            m.initialize_online_rpn_params()
            torch.manual_seed(321)
            with redirect_stdout(io.StringIO()):
                for im in range(NIMG):
                    t, gt = torch.from_numpy(rg["t_%d" % im]), torch.from_numpy(rg["gt_%d" % im])
                    m.head = lambda feats, _t=t: [_t.unsqueeze(0)]
                    m.forward(None, None, gt_bbox=FieldBoxList(gt.clone(), img), img_size=None, result_dir=tmp)
            _flush_like_reference(m, tmp, "features_RPN", m.anchors_ids, D)
            _collect(tmp, "features_RPN", "rpn_shuf" if shuffle else "rpn_fill", out)
    np.savez_compressed(os.path.join(OUT, "feature_cache_golden.npz"), **out)
    print("feature_cache_golden.npz:", len(out), "entries;", {k: len(v) for k, v in out.items() if k.endswith("__files__")})


# --------------------------------------------------------------------------- f3: mask pasting + segmentation AP
def make_masks():
    import torch.nn.functional as Fn
    base = "src/modules/feature-extractor/mrcnn_modified/"
    src = open(os.path.join(REF, base, "modeling/roi_heads/mask_head/inference.py")).read()
    src = "\n".join(l for l in src.splitlines() if not l.strip().startswith("from maskrcnn_benchmark"))

    class _BL(FieldBoxList):
        def convert(self, mode):
            return self

        def has_field(self, k):
            return k in self.extra_fields
    ns = {"interpolate": Fn.interpolate, "BoxList": _BL, "__name__": "ref_mask_inference"}
    exec(compile(src, "mask_inference.py", "exec"), ns)
    out = {}
    g = torch.Generator().manual_seed(2024)
    S, H, W, R = 14, 60, 80, 9
    logits = torch.randn(R, 4, S, S, generator=g) * 2
    labels = torch.randint(1, 4, (R,), generator=g)
    boxes = torch.rand(R, 2, generator=g) * torch.tensor([50.0, 30.0])
    boxes = torch.cat([boxes, boxes + 6 + torch.rand(R, 2, generator=g) * 35], 1)
    boxes[0] = torch.tensor([-6.3, -4.2, 20.7, 18.1])          # sticks out top-left
    boxes[1] = torch.tensor([60.0, 40.0, 95.5, 75.2])          # sticks out bottom-right
    boxes[2] = torch.tensor([30.2, 20.9, 30.9, 21.3])          # sub-pixel box
    boxes[3] = torch.tensor([10.0, 10.0, 40.0, 12.0])          # thin
    pp = ns["MaskPostProcessor"](ns["Masker"](threshold=0.5, padding=1))
    bl = _BL(boxes.clone(), (W, H))
    bl.add_field("labels", labels)
    res = pp(logits.clone(), [bl])[0]
    out["logits"], out["labels"], out["boxes"] = logits.numpy(), labels.numpy(), boxes.numpy()
    out["HW"] = np.array([H, W], dtype=np.int64)
    out["pasted"] = res.get_field("mask").numpy().squeeze(1)
    prob = logits.sigmoid()[torch.arange(R), labels]
    out["prob"] = prob.numpy()
    # segmentation AP on pasted masks (icw_eval.eval_segmentation_ycbv), masks given unpasted as the reference does
    esrc = open(os.path.join(REF, base, "data/datasets/evaluation/icubworld/icw_eval.py")).read()
    esrc = "\n".join(l for l in esrc.splitlines() if not l.strip().startswith(("from maskrcnn_benchmark", "from mrcnn_modified", "from py_od_utils", "import cv2")))
    utils = load_ref("src/py_od_utils.py", "py_od_utils")
    ens = {"BoxList": _BL, "boxlist_iou": None, "Masker": ns["Masker"], "mask_iou": utils.mask_iou, "__name__": "ref_icw_eval_seg"}
    with redirect_stdout(io.StringIO()):
        exec(compile(esrc.replace("'cuda'", "'cpu'"), "icw_eval.py", "exec"), ens)

    class _GTMasks:
        def __init__(self, m):
            self.m = m

        def get_mask_tensor(self):
            return self.m
    rng = np.random.RandomState(5)
    preds, gts = [], []
    NIMG = 14
    out["seg_NIMG"] = np.int64(NIMG)
    for im in range(NIMG):
        G = rng.randint(1, 4)
        gm = np.zeros((G, H, W), dtype=np.float32)
        gl = rng.randint(1, 4, size=G)
        gboxes = []
        for k in range(G):
            x1, y1 = rng.randint(0, W - 30), rng.randint(0, H - 25)
            x2, y2 = x1 + rng.randint(12, 29), y1 + rng.randint(10, 24)
            gm[k, y1:y2, x1:x2] = 1.0
            gboxes.append([x1, y1, x2 - 1, y2 - 1])
        P = rng.randint(1, 5)
        pm = torch.rand(P, 1, S, S, generator=g)
        pbx, hits = [], []
        for k in range(P):
            if rng.rand() < 0.75:
                j = rng.randint(G)
                b = np.array(gboxes[j], dtype=np.float32) + rng.randn(4) * 1.5
                pm[k, 0] = 0.5 + 0.5 * torch.rand(S, S, generator=g)          # mostly "inside"
                hit = j
            else:
                x1, y1 = rng.randint(0, W - 30), rng.randint(0, H - 25)
                b = np.array([x1, y1, x1 + rng.randint(8, 29), y1 + rng.randint(8, 24)], dtype=np.float32)
                hit = rng.randint(G)
            pbx.append(b)
            hits.append(hit)
        pbx = np.array(pbx, dtype=np.float32)
        pl = np.array([gl[hits[k]] if rng.rand() < 0.85 else rng.randint(1, 4) for k in range(P)], dtype=np.int64)
        ps = rng.rand(P).astype(np.float32)
        p = _BL(torch.from_numpy(pbx), (W, H))
        p.add_field("labels", torch.from_numpy(pl))
        p.add_field("scores", torch.from_numpy(ps))
        p.add_field("mask", pm)
        q = _BL(torch.tensor(gboxes, dtype=torch.float32), (W, H))
        q.add_field("labels", torch.from_numpy(gl.astype(np.int64)))
        q.add_field("difficult", torch.zeros(G, dtype=torch.uint8))
        q.add_field("masks", _GTMasks(torch.from_numpy(gm)))
        preds.append(p)
        gts.append(q)
        for k, v in (("seg_pm", pm.numpy().squeeze(1)), ("seg_pb", pbx), ("seg_pl", pl), ("seg_ps", ps), ("seg_gm", gm), ("seg_gl", gl.astype(np.int64))):
            out["%s_%d" % (k, im)] = v
    with np.errstate(all="ignore"):
        for m07 in (True, False):
            r = ens["eval_segmentation_ycbv"](preds, gts, iou_thresh=0.5, use_07_metric=m07)
            out["seg_ap_%s" % ("voc07" if m07 else "area")] = np.asarray(r["ap"], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "masks_golden.npz"), **out)
    print("masks_golden.npz:", len(out), "arrays; pasted pixels", int(out["pasted"].sum()), "seg AP", out["seg_ap_voc07"])


def make_featstats():
    """computeFeatStatistics (py_od_utils.py:8-56), the numpy-RNG variant with the stats cache file: run from a scratch
    copy of the module namespace whose __file__ points into a temporary directory, so the cache file is written there."""
    import tempfile
    utils = load_ref("src/py_od_utils.py", "ref_py_od_utils_stats")
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "Data", "feat_cache", "folder"))
    os.makedirs(os.path.join(tmp, "Data", "feat_cache_RPN", "folder"))
    utils.__dict__["__file__"] = os.path.join(tmp, "src", "py_od_utils.py")
    os.makedirs(os.path.join(tmp, "src"))
    out = {}
    g = torch.Generator().manual_seed(31)
    D, C = 12, 4
    positives = [torch.randn(20 + 3 * c, D, generator=g) + c for c in range(C)]
    positives[3] = torch.empty((0, D))
    negatives = [[torch.randn(25, D, generator=g) - c for _ in range(3)] for c in range(C)]
    negatives[1][2] = torch.empty((0, D))
    for c in range(C):
        out["pos_%d" % c] = positives[c].numpy()
        for j in range(3):
            out["neg_%d_%d" % (c, j)] = negatives[c][j].numpy()
    for tag, is_rpn in (("det", False), ("rpn", True)):
        np.random.seed(77)
        with redirect_stdout(io.StringIO()):
            mean, std, mean_norm = utils.computeFeatStatistics(positives, negatives, "folder", is_rpn, num_samples=120)
        out[tag + "_mean"], out[tag + "_std"], out[tag + "_mean_norm"] = mean.numpy(), std.numpy(), mean_norm.numpy()
        path = os.path.join(tmp, "Data", "feat_cache_RPN" if is_rpn else "feat_cache", "folder", "rpn_stats" if is_rpn else "stats")
        saved = torch.load(path)
        out[tag + "_saved_mean"] = saved["mean"].numpy()
        # second call: the cache file answers, no draw is made
        state = np.random.get_state()[1].copy()
        with redirect_stdout(io.StringIO()):
            m2, _, _ = utils.computeFeatStatistics(positives, negatives, "folder", is_rpn, num_samples=120)
        assert np.array_equal(state, np.random.get_state()[1]) and np.array_equal(m2.numpy(), mean.numpy())
    np.savez_compressed(os.path.join(OUT, "featstats_golden.npz"), **out)
    print("featstats_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    if "--only-masks" in sys.argv:
        make_masks()
        sys.exit(0)
    if "--only-feature-cache" in sys.argv:
        make_feature_cache()
        sys.exit(0)
    if "--only-featstats" in sys.argv:
        make_featstats()
        sys.exit(0)
    if "--only-postprocess" in sys.argv:
        make_postprocess()
        make_eval()
        sys.exit(0)
    if "--only-mask-harvest" in sys.argv:
        make_mask_harvest()
        sys.exit(0)
    if "--only-rpn-harvest" in sys.argv:
        make_rpn_harvest()
        sys.exit(0)
    if "--only-harvest" in sys.argv:
        make_harvest()
        sys.exit(0)
    if "--only-heads" in sys.argv:
        make_heads()
        sys.exit(0)
    utils = make_rls()
    make_bootstrap()
    make_helpers(utils)
    make_wrapper_contract()
    make_heads()
    make_harvest()
    make_rpn_harvest()
    make_mask_harvest()
    make_postprocess()
    make_eval()
    make_feature_cache()
    make_masks()
    make_featstats()
