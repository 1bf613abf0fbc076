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


if __name__ == "__main__":
    utils = make_rls()
    make_bootstrap()
    make_helpers(utils)
    make_wrapper_contract()
