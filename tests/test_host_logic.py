"""Host logic of the product (solver, estimator objects, drop-in modules) on CPU, driven through
the numpy-oracle backend (tests/oracle_backend.py) and checked against the oracle and against
the golden vectors the reference itself produced (tests/golden/*.npz, *.json)."""
import copy
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import odx
from oracle import falkon_ref as fr
from tests import dropin
from tests.oracle_backend import OracleBackend
from tests.synth import blob_problem, centres

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# Reference-pinned checks that only need torch ops / the product backend also run on HIP tensors (`-m gpu`).
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


@pytest.fixture(autouse=True)
def oracle_backend(request):
    """CPU arms drive the host logic through the numpy-oracle backend; the cuda arms of the device-parametrised
    tests run on the product's HIP backend (libodx.so)."""
    params = getattr(getattr(request.node, "callspec", None), "params", {})
    if params.get("device") == "cuda":
        odx.set_backend(None)
        assert odx.get_backend().name == "hip-gfx950"
    else:
        odx.set_backend(OracleBackend(np.float64))
    yield
    odx.set_backend(None)


def quiet(fn, *a, **kw):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


# ------------------------------------------------------------------ solver / estimator objects
@pytest.mark.parametrize("sigma,lam", [(5.0, 1e-3), (10.0, 1e-5)])
def test_solver_equals_oracle(sigma, lam):
    X, y, rng = blob_problem(1500, 48, seed=31)
    idx = centres(y, 150, rng)
    be = odx.get_backend()
    F = be.features(torch.from_numpy(X))
    alpha = odx.falkon_fit(be, F, be.vec(y), be.rows(F, idx), sigma, lam, 20).numpy()
    ref, _ = fr.falkon_fit(X.astype(np.float64), y, idx, sigma, lam, maxiter=20, dtype=np.float64, pc_eps=1e-5,
                           cg_epsilon=1e-7)
    assert np.linalg.norm(alpha - ref[:, 0]) / np.linalg.norm(ref[:, 0]) < 1e-6


def test_folded_full_residual_equals_the_recomputation():
    """The periodic residual recomputation folded into the neighbouring step's two-vector pass (solver.py: W x_new =
    W x_old + a W p) gives the iterate of the plain sequence (its own pass for R = B - W x) up to f64 rounding — at
    iteration counts that end before, on and after a recomputation step."""
    X, y, rng = blob_problem(1500, 48, seed=32)
    idx = centres(y, 150, rng)
    for maxiter in (9, 10, 11, 20, 25):
        out = {}
        for fold in (True, False):
            be = OracleBackend(np.float64)
            be.fold = fold
            calls = {"ktk": 0, "ktk2": 0}
            k1, k2 = be.ktk, be.ktk2
            be.ktk = lambda *a, _k=k1, **kw: (calls.__setitem__("ktk", calls["ktk"] + 1), _k(*a, **kw))[1]
            be.ktk2 = lambda *a, _k=k2, **kw: (calls.__setitem__("ktk2", calls["ktk2"] + 1), _k(*a, **kw))[1]
            F = be.features(torch.from_numpy(X))
            out[fold] = (odx.falkon_fit(be, F, be.vec(y), be.rows(F, idx), 10.0, 1e-5, maxiter).numpy(), dict(calls))
        (a, ca), (b, cb) = out[True], out[False]
        # f64 rounding of the two evaluation orders, amplified by the system's conditioning: ~1e-11 .. 1e-10 measured
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-8, maxiter
        nfull = sum(1 for it in range(maxiter - 1) if (it + 1) % 10 == 0)
        assert ca["ktk2"] == nfull and cb["ktk2"] == 0


def test_estimator_surface():
    X, y, rng = blob_problem(600, 24, seed=8)
    idx = centres(y, 60, rng)
    from odx.wrappers import CenterSelector
    m = odx.InCoreFalkon(kernel=odx.GaussianKernel(sigma=6.0), penalty=1e-3, M=len(idx), maxiter=20,
                         center_selection=CenterSelector(idx), options=odx.FalkonOptions(keops_active="no"))
    Xt, yt = torch.from_numpy(X), torch.from_numpy(y)
    assert m.fit(Xt, yt) is m
    assert m.M == 60 and tuple(m.ny_points_.shape) == (60, 24) and tuple(m.alpha_.shape) == (60, 1)
    p = m.predict(Xt[:7])
    assert tuple(p.shape) == (7, 1) and p.dtype == torch.float32
    ref, Z = fr.falkon_fit(X.astype(np.float64), y, idx, 6.0, 1e-3, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
    assert np.abs(p.numpy() - fr.falkon_predict(X[:7].astype(np.float64), Z, ref, 6.0)).max() < 1e-5
    # what the callers do with a model: truthiness, deep copy, re-assignable tensors, pickle round trip
    assert bool(m)
    m2 = copy.deepcopy(m)
    m2.alpha_ = m2.alpha_.to("cpu")
    m2.ny_points_ = m2.ny_points_.to("cpu")
    buf = io.BytesIO()
    torch.save([m2, None], buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)[0]
    assert torch.equal(m3.alpha_, m.alpha_) and m3.kernel.sigma == 6.0 and m3.M == 60
    # the centres' kernel operands stay with the model while ny_points_ is the same, unmodified tensor — and only then
    assert "_zf" in m.__dict__ and "_zf" not in m2.__dict__ and "_zf" not in m3.__dict__
    assert torch.equal(m3.predict(Xt[:7]), p) and torch.equal(m.predict(Xt[:7]), p) and m3._zf[3] is m3._centres()
    m3.ny_points_[0] += 1.0                                            # in place: the version counter moves
    p_moved = m3.predict(Xt[:7])
    assert not torch.equal(p_moved, p)
    m3.ny_points_ = m.ny_points_.clone()                               # re-assigned (falkon_models_to_cuda does this)
    assert torch.equal(m3.predict(Xt[:7]), p)
    # kernel.mmv with a block-structured alpha_parallel (roi_box_predictors.py:140-160)
    alpha_par = torch.zeros(120, 2, dtype=torch.float64)
    alpha_par[:60, 0] = m.alpha_[:, 0]
    alpha_par[60:, 1] = 2 * m.alpha_[:, 0]
    s = m.kernel.mmv(Xt[:7], torch.cat([m.ny_points_, m.ny_points_]), alpha_par)
    assert np.allclose(s[:, 0].numpy(), p[:, 0].numpy(), atol=1e-6) and np.allclose(s[:, 1].numpy(), 2 * p[:, 0].numpy(), atol=1e-5)


def test_train_batch_on_a_backend_without_batched_kernels_equals_train():
    """FALKONWrapper.train_batch (the class-batched Minibootstrap's entry point) on a backend that has no batched
    preconditioner (the numpy-oracle backend of this suite): every class is fitted on its own, with the same result as
    `train` given the same index draws."""
    mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    w = mod.FALKONWrapper(cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
    Xs, ys = [], []
    for seed in (1, 2, 3):
        X, y, _ = blob_problem(260 + 20 * seed, 16, seed=seed)
        Xs.append(torch.from_numpy(X)), ys.append(torch.from_numpy(y))
    torch.manual_seed(9)
    batch = quiet(w.train_batch, Xs, ys, sigma=7.0, lam=0.01)
    torch.manual_seed(9)
    single = [quiet(w.train, X, y, sigma=7.0, lam=0.01) for X, y in zip(Xs, ys)]
    assert len(batch) == 3
    for a, b in zip(batch, single):
        assert a.M == b.M and torch.equal(a.ny_points_, b.ny_points_) and torch.equal(a.alpha_, b.alpha_)
        assert tuple(a.predict(Xs[0][:5]).shape) == (5, 1)


def test_two_phase_train_batch_equals_train_batch():
    """train_batch_begin / add (group by group) / finish: the same models as one train_batch call over all classes with the
    same index draws — here on the backend without batched kernels (everything through the general path in finish)."""
    mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    w = mod.FALKONWrapper(cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
    Xs, ys = [], []
    for seed in (1, 2, 3, 4):
        X, y, _ = blob_problem(240 + 20 * seed, 16, seed=seed)
        Xs.append(torch.from_numpy(X)), ys.append(torch.from_numpy(y))
    torch.manual_seed(9)
    whole = quiet(w.train_batch, Xs, ys, sigma=7.0, lam=0.01)
    torch.manual_seed(9)
    seen = []

    def run():
        rnd = w.train_batch_begin(sigma=7.0, lam=0.01, expect_total=4)
        rnd.add(Xs[:3], ys[:3], index_rng=lambda pos, fn: (seen.append(pos), fn())[1])
        rnd.add(Xs[3:], ys[3:], index_rng=lambda pos, fn: (seen.append(pos), fn())[1])
        return rnd.finish()
    parts = quiet(run)
    assert seen == [0, 1, 2, 3] and len(parts) == 4
    for a, b in zip(parts, whole):
        assert a.M == b.M and torch.equal(a.ny_points_, b.ny_points_) and torch.equal(a.alpha_, b.alpha_)


def test_multi_output_fit_is_rejected_loudly():
    m = odx.InCoreFalkon(kernel=odx.GaussianKernel(5.0), penalty=1e-3, M=10)
    with pytest.raises(ValueError):
        m.fit(torch.randn(50, 4), torch.randn(50, 2))


def test_block_ranges():
    from odx.falkon import block_ranges
    v = torch.zeros(10, 4)
    v[2:5, 0] = 1
    v[5:10, 2] = -1
    v[0, 3] = 3
    assert block_ranges(v).tolist() == [[2, 5], [0, 0], [5, 10], [0, 1]]


# ------------------------------------------------------------------ FALKONWrapper call contract
@pytest.mark.parametrize("variant,modname", [("incore", "FALKONWrapper_with_centers_selection_incore"),
                                              ("cpu", "FALKONWrapper_with_centers_selection")])
def test_falkon_wrapper_contract_matches_reference(variant, modname):
    contract = json.load(open(os.path.join(GOLD, "wrapper_contract.json")))[variant]
    mod = dropin.load(modname)
    rec = {"ctor": [], "fit": [], "select": []}

    class Recorder:
        def __init__(self, **kw):
            from odx.falkon import FalkonOptions, GaussianKernel
            rec["ctor"].append({k: (dict(v.extra) if isinstance(v, FalkonOptions) else
                                    ("GaussianKernel(%g)" % v.sigma if isinstance(v, GaussianKernel) else
                                     ("MyCenterSelector" if k == "center_selection" else v))) for k, v in kw.items()})
            self.cs = kw["center_selection"]

        def fit(self, X, y):
            Z = self.cs.select(X, None)
            rec["select"].append({"returns_tensor": bool(torch.is_tensor(Z)), "shape": list(Z.shape)})
            rec["fit"].append({"X": list(X.shape), "y": list(y.shape), "y_dtype": str(y.dtype)})

        def predict(self, X):
            return torch.zeros(X.shape[0], 1)

    w = mod.FALKONWrapper(cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
    w.estimator_incore = w.estimator_cpu = Recorder
    assert {"sigma": w.sigma, "lam": w.lam, "nyst_centers": w.nyst_centers, "maxiter": getattr(w, "maxiter", None)} == contract["attrs"]
    g = torch.Generator().manual_seed(3)
    X = torch.randn(100, 8, generator=g)
    y = torch.cat([torch.ones(30), -torch.ones(70)])
    torch.manual_seed(77)
    assert w.compute_indices_selection(y) == contract["indices_seed77"]
    torch.manual_seed(77)
    model = w.train(X, y, sigma=7.0, lam=0.01)
    assert list(w.predict(model, X[:5]).shape) == contract["predict_shape"]
    assert rec["ctor"] == contract["record"]["ctor"]
    assert rec["fit"] == contract["record"]["fit"]
    assert rec["select"] == contract["record"]["select"]
    assert w.compute_indices_selection(torch.cat([torch.ones(5), -torch.ones(9)])) == contract["indices_small"]
    assert isinstance(w.compute_indices_selection(torch.tensor([1.0])), int) == contract["indices_single_is_int"]


# ------------------------------------------------------------------ minibootstrap state machine
class RidgeClassifier:
    """Same deterministic stand-in the golden generator plugged into the REFERENCE's loop."""

    def __init__(self, lam=1.0):
        self.lam = lam
        self.calls = []

    def train(self, X, y, sigma=None, lam=None):
        A = torch.cat([X.double(), torch.ones(len(X), 1, dtype=torch.float64)], 1)
        w = torch.linalg.solve(A.T @ A + self.lam * torch.eye(A.shape[1], dtype=torch.float64), A.T @ y.double())
        self.calls.append(["train", int((y == 1).sum()), int((y == -1).sum())])
        return {"w": w}

    def predict(self, model, X, y=None):
        A = torch.cat([X.double(), torch.ones(len(X), 1, dtype=torch.float64)], 1)
        self.calls.append(["predict", len(X)])
        return (A @ model["w"]).float().view(-1, 1)


@pytest.mark.parametrize("variant,modname", [("cpu", "OnlineRegionClassifier"), ("incore", "OnlineRegionClassifier_incore")])
def test_minibootstrap_matches_reference_trace(variant, modname):
    G = np.load(os.path.join(GOLD, "bootstrap_golden.npz"))
    mod = dropin.load(modname)
    C, ITER = int(G["C"]), int(G["ITER"])
    pos = [torch.from_numpy(G["pos_%d" % c]) for c in range(C)]
    neg = [[torch.from_numpy(G["neg_%d_%d" % (c, j)]) for j in range(ITER)] for c in range(C)]
    stats = {"mean": torch.from_numpy(G["mean"]), "std": torch.ones(int(G["D"])), "mean_norm": torch.tensor(float(G["mean_norm"]))}
    clf = RidgeClassifier()
    orc = mod.OnlineRegionClassifier(clf, pos, neg, stats, cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
    models = quiet(orc.trainRegionClassifier)
    assert clf.calls == json.loads(str(G[variant + "_calls"]))
    assert len(models) == C
    for c, m in enumerate(models):
        assert (m is None) == bool(G["%s_model_%d_none" % (variant, c)])
        if m is not None:
            assert np.allclose(m["w"].numpy(), G["%s_model_%d_w" % (variant, c)], rtol=1e-9, atol=1e-12)
    assert np.allclose(pos[0].numpy(), G[variant + "_pos0_normalized"])  # normalised in place, once
    assert orc.normalized
    if variant == "incore":
        tb = [{"boxes": G["test_boxes_%d" % im], "feat": G["test_feat_%d" % im], "gt": G["test_gt_%d" % im],
               "img_size": (320, 240)} for im in range(2)]
        orc2 = mod.OnlineRegionClassifier(RidgeClassifier(), pos, neg, stats, cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
        preds = quiet(orc2.testRegionClassifier, [models[0], models[0], models[2]], tb)
        for im in range(2):
            sc = preds[im].get_field("scores").numpy()
            assert sc.shape == G["test_scores_%d" % im].shape and np.allclose(sc, G["test_scores_%d" % im], atol=1e-6)
            assert np.all(sc[:, 0] == -1)  # background column of the stand-alone scorer


def test_minibootstrap_return_caches_and_options():
    G = np.load(os.path.join(GOLD, "bootstrap_golden.npz"))
    mod = dropin.load("OnlineRegionClassifier_incore")
    C, ITER = int(G["C"]), int(G["ITER"])
    pos = [torch.from_numpy(G["pos_%d" % c]) for c in range(C)]
    neg = [[torch.from_numpy(G["neg_%d_%d" % (c, j)]) for j in range(ITER)] for c in range(C)]
    stats = {"mean": torch.from_numpy(G["mean"]), "std": torch.ones(int(G["D"])), "mean_norm": torch.tensor(float(G["mean_norm"]))}
    orc = mod.OnlineRegionClassifier(RidgeClassifier(), pos, neg, stats, cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
    models, caches = quiet(orc.trainRegionClassifier, {"return_caches": True, "lam": 0.5, "sigma": 3})
    assert orc.lam == 0.5 and orc.sigma == 3 and len(caches) == C
    assert caches[1] == {} and models[1] is None and set(caches[0]) == {"pos", "neg"}


# ------------------------------------------------------------------ py_od_utils
@pytest.mark.parametrize("device", DEVICES)
def test_feat_statistics_with_cache_file_match_reference(device, tmp_path):
    """computeFeatStatistics (the numpy-RNG variant with the `stats` cache file, py_od_utils.py:8-56) against the
    reference's own run: same draws, same statistics, same file; the second call is answered by the file without a draw."""
    H = np.load(os.path.join(GOLD, "featstats_golden.npz"))
    from odx.utils import computeFeatStatistics
    assert dropin.load("py_od_utils").computeFeatStatistics is computeFeatStatistics
    C = 4
    positives = [torch.from_numpy(H["pos_%d" % c]).to(device) for c in range(C)]
    negatives = [[torch.from_numpy(H["neg_%d_%d" % (c, j)]).to(device) for j in range(3)] for c in range(C)]
    base = str(tmp_path / "src")
    os.makedirs(base)
    for tag, is_rpn in (("det", False), ("rpn", True)):
        os.makedirs(str(tmp_path / "Data" / ("feat_cache_RPN" if is_rpn else "feat_cache") / "folder"))
        np.random.seed(77)
        mean, std, mean_norm = quiet(computeFeatStatistics, positives, negatives, "folder", is_rpn, num_samples=120, basedir=base)
        assert np.array_equal(mean.numpy(), H[tag + "_mean"]) and np.array_equal(std.numpy(), H[tag + "_std"])
        assert np.array_equal(mean_norm.numpy(), H[tag + "_mean_norm"])
        saved = torch.load(str(tmp_path / "Data" / ("feat_cache_RPN" if is_rpn else "feat_cache") / "folder" / ("rpn_stats" if is_rpn else "stats")))
        assert np.array_equal(saved["mean"].numpy(), H[tag + "_saved_mean"])
        state = np.random.get_state()[1].copy()
        m2, _, _ = quiet(computeFeatStatistics, positives, negatives, "folder", is_rpn, num_samples=120, basedir=base)
        assert np.array_equal(state, np.random.get_state()[1]) and torch.equal(m2, mean)


@pytest.mark.parametrize("device", DEVICES)
def test_py_od_utils_match_reference(device):
    """A10 against vectors from the reference's own py_od_utils / MyCenterSelector; the cuda arm keeps every tensor
    (feature lists, COXY, statistics) on the MI355X as the drivers do."""
    H = np.load(os.path.join(GOLD, "helpers_golden.npz"))
    u = dropin.load("py_od_utils")
    C, D = 3, 16
    cpu = device == "cpu"
    positives = [torch.from_numpy(H["pos_%d" % c]).to(device) for c in range(C)]
    negatives = [[torch.from_numpy(H["neg_%d_%d" % (c, j)]).to(device) for j in range(3)] for c in range(C)]
    torch.manual_seed(1234)
    st = quiet(u.computeFeatStatistics_torch, positives, negatives, num_samples=90, features_dim=D, cpu_tensor=cpu,
               pos_fraction=0.8)
    if not cpu:
        assert all(v.is_cuda for v in st.values())
    assert np.allclose(st["mean"].cpu().numpy(), H["stats_mean"], atol=1e-6)
    assert np.allclose(st["std"].cpu().numpy(), H["stats_std"], atol=1e-6)
    assert np.allclose(st["mean_norm"].cpu().numpy(), H["stats_mean_norm"], atol=1e-6)
    COXY = {"C": torch.from_numpy(H["coxy_C"]).to(device), "O": None, "X": torch.from_numpy(H["coxy_X"]).to(device), "Y": None}
    stc = {k: v.to(device) for k, v in st.items()}
    Xn = u.normalize_COXY(dict(COXY), stc, cpu=cpu)["X"]
    assert Xn.device.type == device and np.allclose(Xn.cpu().numpy(), H["coxy_X_normalized"], atol=1e-6)
    pf = u.load_positives_from_COXY({"C": COXY["C"][:, 0].clone(), "X": COXY["X"].clone()})
    assert len(pf) == int(H["pos_from_coxy_n"])
    for i, p in enumerate(pf):
        assert p.device.type == device and np.array_equal(p.cpu().numpy(), H["pos_from_coxy_%d" % i])
    torch.manual_seed(99)
    sh = u.shuffle_negatives([[b.clone() for b in nb] for nb in negatives], batch_size=40, num_batches=3)
    for c in range(C):
        for j in range(3):
            assert sh[c][j].device.type == device and np.array_equal(sh[c][j].cpu().numpy(), H["shuf_%d_%d" % (c, j)])
    z = u.zScores(positives[0].cpu().numpy(), stc["mean"].cpu(), stc["mean_norm"].cpu())
    assert np.allclose(z.numpy(), H["zscores"], atol=1e-6)
    sel = dropin.load("MyCenterSelector")
    Xs, Ys, idx = torch.from_numpy(H["sel_X"]).to(device), torch.from_numpy(H["sel_Y"]).to(device), H["sel_idx"].tolist()
    assert np.array_equal(sel.MyCenterSelector(idx).select(Xs, None).cpu().numpy(), H["sel_out_X"])
    xo, yo = sel.MyCenterSelector(idx).select(Xs, Ys)
    assert xo.device.type == device
    assert np.array_equal(xo.cpu().numpy(), H["sel_out_X2"]) and np.array_equal(yo.cpu().numpy(), H["sel_out_Y2"])


@pytest.mark.parametrize("device", DEVICES)
def test_decode_boxes_detector_and_feature_cache_roundtrip(tmp_path, device):
    R = np.load(os.path.join(GOLD, "rls_golden.npz"))
    u = dropin.load("py_od_utils")
    from odx.boxlist import BoxList
    out = u.decode_boxes_detector(BoxList(torch.from_numpy(R["apply_boxes_0"]).to(device), (320, 240)),
                                  torch.from_numpy(R["decode_in"]).to(device))
    assert out.device.type == device and np.allclose(out.cpu().numpy(), R["decode_out"], atol=1e-4)
    # on-disk feature cache layout (py_od_utils.py:153-217)
    d = str(tmp_path)
    for c in range(2):
        for b in range(2):
            torch.save(torch.full((3, 4), float(10 * c + b), device=device), os.path.join(d, "positives_cl_%d_batch_%d" % (c, b)))
            torch.save(torch.full((2, 4), -float(10 * c + b), device=device), os.path.join(d, "negatives_cl_%d_batch_%d" % (c, b)))
    torch.save(torch.ones(5, 4, device=device), os.path.join(d, "reg_x_batch_0"))
    torch.save(torch.ones(5, 1, device=device), os.path.join(d, "reg_c_batch_0"))
    torch.save(torch.ones(5, 4, device=device), os.path.join(d, "reg_y_batch_0"))
    pos, neg = u.load_features_classifier(d, cpu_tensor=(device == "cpu"))
    assert pos[0].device.type == device
    assert [tuple(p.shape) for p in pos] == [(6, 4), (6, 4)] and [len(n) for n in neg] == [2, 2]
    assert float(neg[1][1][0, 0]) == -11.0
    pos_s, neg_s = u.load_features_classifier(d, is_segm=True, cpu_tensor=(device == "cpu"))
    assert tuple(neg_s[0].shape) == (4, 4)
    coxy = u.load_features_regressor(d)
    assert tuple(coxy["X"].shape) == (5, 4) and coxy["O"] is None


# ------------------------------------------------------------------ RLS modules
@pytest.mark.parametrize("tag,is_rpn", [("det", False), ("rpn", True)])
def test_region_refiner_matches_reference(tag, is_rpn, tmp_path):
    import yaml
    R = np.load(os.path.join(GOLD, "rls_golden.npz"))
    cfg = {"CHOSEN_CLASSES": {i: str(c) for i, c in enumerate(R["classes"])}, "REGION_REFINER": {"opts": {"lambda": float(R["lambda"])}}}
    if is_rpn:
        cfg = {"RPN": cfg}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    rr = dropin.load("region_refiner").RegionRefiner(path, is_rpn=is_rpn)
    C = torch.from_numpy(R["C"] if not is_rpn else R["C"] - 1)
    models = quiet(rr.trainRegionRefiner, {"C": C, "O": None, "X": torch.from_numpy(R["X"]), "Y": torch.from_numpy(R["Y"])},
                   output_dir=str(tmp_path))
    assert isinstance(models, np.ndarray) and len(models) == int(R[tag + "_num_models"])
    for i, m in enumerate(models):
        assert set(m) == {"mu", "T", "T_inv", "Beta"}
        if bool(R["%s_%d_none" % (tag, i)]):
            assert m["Beta"] is None and m["mu"] is None
            continue
        assert set(m["Beta"]) == {"0", "1", "2", "3"} and m["mu"].dtype == torch.float32
        for key in ("mu", "T", "T_inv"):
            assert np.allclose(m[key].cpu().numpy(), R["%s_%d_%s" % (tag, i, key)], atol=2e-6)
        W = np.stack([m["Beta"][str(k)]["weights"].cpu().numpy() for k in range(4)])
        assert np.abs(W - R["%s_%d_W" % (tag, i)]).max() < 2e-6
        L = np.stack([m["Beta"][str(k)]["losses"].cpu().numpy() for k in range(4)])
        assert np.abs(L - R["%s_%d_losses" % (tag, i)]).max() < 1e-5
    line = open(os.path.join(str(tmp_path), "result.txt")).read()
    assert line.startswith("RPN's Online Region Refiner training time" if is_rpn else "Detector's Online Region Refiner training time")
    # RegionRefinerTrainer.solve on its own (train_region_refiner.py:100-119): fed what the reference's train feeds it
    # (f64 rows with the bias column appended, whitened targets) it returns the golden weights / losses of that class
    from odx.rls import RegionRefinerTrainer, whiten_targets
    tr = RegionRefinerTrainer({"CHOSEN_CLASSES": cfg["RPN"]["CHOSEN_CLASSES"] if is_rpn else cfg["CHOSEN_CLASSES"]}, float(R["lambda"]), is_rpn)
    i = next(k for k in range(len(models)) if not bool(R["%s_%d_none" % (tag, k)]))
    rows = (C.reshape(-1) == (i if is_rpn else i + 1)).nonzero().reshape(-1)
    Xi = torch.cat((torch.from_numpy(R["X"])[rows].double(), torch.ones((len(rows), 1), dtype=torch.float64)), dim=1)
    mu, Yc, T, _ = whiten_targets(torch.from_numpy(R["Y"])[rows].double())
    Yw = Yc @ T
    beta = tr.solve(Xi, Yw, float(R["lambda"]))
    assert np.abs(np.stack([beta[str(k)]["weights"].cpu().numpy() for k in range(4)]) - R["%s_%d_W" % (tag, i)]).max() < 2e-6
    assert np.abs(np.stack([beta[str(k)]["losses"].cpu().numpy() for k in range(4)]) - R["%s_%d_losses" % (tag, i)]).max() < 1e-5
    # per-coordinate row subsets (`indices`), and a matrix that is not "features + ones" (dense f64 route): against numpy
    sub = [torch.arange(k, len(rows), 2) for k in range(4)]
    lam = float(R["lambda"])
    for Xm in (Xi, Xi * 1.000000123):
        got = tr.solve(Xm, Yw, lam, indices=sub)
        for k in range(4):
            A, b = Xm[sub[k]].numpy(), Yw[sub[k], k].numpy()
            w = np.linalg.solve(A.T @ A + lam * np.eye(A.shape[1]), A.T @ b)
            assert np.abs(got[str(k)]["weights"].cpu().numpy() - w).max() < 2e-6 * max(1.0, np.abs(w).max())
            assert np.abs(got[str(k)]["losses"].cpu().numpy() - 0.5 * (A @ w - b) ** 2).max() < 1e-5
    if not is_rpn:
        from odx.boxlist import BoxList
        import yaml as _y
        cfg3 = {"CHOSEN_CLASSES": {0: "_background_", 1: "a", 2: "b"}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
        p3 = str(tmp_path / "cfg3.yaml")
        _y.safe_dump(cfg3, open(p3, "w"))
        rr3 = dropin.load("region_refiner").RegionRefiner(p3)
        boxes = [BoxList(torch.from_numpy(R["apply_boxes_%d" % im]), (320, 240)) for im in range(2)]
        feats = [{"feat": R["apply_feat_%d" % im], "gt": R["apply_gt_%d" % im]} for im in range(2)]
        res = rr3.predict(boxes, feats, models=models[:2])
        for im in range(2):
            assert tuple(res[im].bbox.shape) == R["apply_out_%d" % im].shape
            assert np.abs(res[im].bbox.cpu().numpy() - R["apply_out_%d" % im]).max() < 2e-3


# ------------------------------------------------------------------ INTEGRATION.md route 2
@pytest.mark.skipif(not os.path.exists("/root/reference/src/modules/region-classifier"),
                    reason="reference checkout only exists in the build container")
def test_reference_wrapper_runs_unchanged_on_odx_falkon(monkeypatch):
    """sys.modules['falkon'] = odx.falkon: the REFERENCE's own FALKONWrapper (CPU variant) trains and
    predicts through odx.  (CPU test, never runs on the GPU box.)"""
    import sys
    import types
    import odx.falkon
    monkeypatch.setitem(sys.modules, "falkon", odx.falkon)
    opts = types.ModuleType("falkon.options")
    opts.FalkonOptions = odx.falkon.FalkonOptions
    opts.__all__ = ["FalkonOptions"]
    monkeypatch.setitem(sys.modules, "falkon.options", opts)
    ref_dir = "/root/reference/src/modules/region-classifier"
    src = open(os.path.join(ref_dir, "FALKONWrapper_with_centers_selection.py")).read()
    mod = types.ModuleType("ref_wrapper_under_test")
    mod.__file__ = os.path.join(ref_dir, "FALKONWrapper_with_centers_selection.py")
    monkeypatch.syspath_prepend(ref_dir)
    for name in ("ClassifierAbstract", "MyCenterSelector"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    for name in ("ClassifierAbstract", "MyCenterSelector"):
        sys.modules.pop(name, None)
    w = mod.FALKONWrapper(cfg_path=os.path.join(GOLD, "cfg_bootstrap.yaml"))
    X, y, rng = blob_problem(300, 16, seed=4)
    Xt, yt = torch.from_numpy(X), torch.from_numpy(y)
    torch.manual_seed(3)
    model = quiet(w.train, Xt, yt)
    torch.manual_seed(3)
    idx = fr.compute_indices_selection(y, 40, lambda high, size: torch.randint(high, (size,)).numpy())
    ref, Z = fr.falkon_fit(X.astype(np.float64), y, idx, 10.0, 1e-3, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
    assert np.linalg.norm(model.alpha_.numpy() - ref) / np.linalg.norm(ref) < 1e-6
    p = w.predict(model, Xt[:6])
    assert tuple(p.shape) == (6, 1)


# ------------------------------------------------------------------ test-time heads (A8 / A9)
class _GoldModel:
    def __init__(self, ny, alpha, sigma):
        self.ny_points_, self.alpha_, self.M = torch.from_numpy(ny), torch.from_numpy(alpha), ny.shape[0]
        self.kernel = odx.GaussianKernel(sigma)


def _gold_models(H, tag):
    cls = [None if m == 0 else _GoldModel(H["%s_ny_%d" % (tag, i)], H["%s_alpha_%d" % (tag, i)], float(H["sigma"]))
           for i, m in enumerate(H[tag + "_Ms"])]
    regs = []
    for j, none in enumerate(H[tag + "_reg_none"]):
        if none:
            regs.append({"mu": None, "T": None, "T_inv": None, "Beta": None})
        else:
            W = H["%s_reg_W_%d" % (tag, j)]
            regs.append({"mu": torch.from_numpy(H["%s_reg_mu_%d" % (tag, j)]), "T": torch.eye(4),
                         "T_inv": torch.from_numpy(H["%s_reg_Tinv_%d" % (tag, j)]),
                         "Beta": {str(k): {"weights": torch.from_numpy(W[k]), "losses": None} for k in range(4)}})
    stats = {"mean": torch.from_numpy(H["stats_mean"]), "mean_norm": torch.tensor(float(H["stats_mean_norm"]))}
    return cls, np.array(regs, dtype=object), stats


def check_heads_against_reference(atol=2e-4):
    """Shared by the CPU (oracle backend) and GPU suites: the product heads against the outputs
    of the reference's own FastRCNNPredictor / RPNHead code (tests/golden/heads_golden.npz)."""
    from odx.heads import OnlineBoxPredictor, OnlineRPNHead
    H = np.load(os.path.join(GOLD, "heads_golden.npz"))
    cls, regs, stats = _gold_models(H, "det")
    x = torch.from_numpy(H["det_x"])
    for par in (1, 0):
        for norm in (0, 1):
            head = OnlineBoxPredictor(cls, regs, stats, parallel_inference=bool(par), normalize_features_regressors=bool(norm))
            sc, bb = head(x)
            tag = "det_par%d_norm%d" % (par, norm)
            assert tuple(sc.shape) == H[tag + "_scores"].shape and tuple(bb.shape) == H[tag + "_bbox"].shape
            assert np.abs(sc.cpu().numpy() - H[tag + "_scores"]).max() < atol, tag
            assert np.abs(bb.cpu().numpy() - H[tag + "_bbox"]).max() < atol, tag
            sc2, _ = head(x.reshape(9, -1, 1, 1))          # 4-D input is average-pooled
            assert torch.allclose(sc2, sc)
    cls, regs, stats = _gold_models(H, "rpn")
    t = torch.from_numpy(H["rpn_act"])
    for par in (1, 0):
        lg, bb = OnlineRPNHead(cls, regs, stats, parallel_inference=bool(par))(t)
        assert tuple(lg.shape) == H["rpn_par%d_logits" % par].shape
        assert np.abs(lg.cpu().numpy() - H["rpn_par%d_logits" % par]).max() < atol
        assert np.abs(bb.cpu().numpy() - H["rpn_par%d_bbox" % par]).max() < atol
    from odx.heads import OnlineMaskPredictor
    cls, _, stats = _gold_models(H, "mask")
    for par in (1, 0):
        got = OnlineMaskPredictor(cls, stats, parallel_inference=bool(par))(torch.from_numpy(H["mask_act"]))
        assert tuple(got.shape) == H["mask_par%d_out" % par].shape
        assert np.abs(got.cpu().numpy() - H["mask_par%d_out" % par]).max() < atol
    cls, regs, stats = _gold_models(H, "rpn")
    # model hot-swap invalidates the cached concatenations
    head = OnlineRPNHead(cls, regs, stats)
    a, _ = head(t)
    head.set_models(cls[:3] + [None, None, None], regs, stats)
    b, _ = head(t)
    assert torch.allclose(a[:, :2], b[:, :2]) and torch.all(b[:, 3:] == -2)


def test_heads_match_reference():
    check_heads_against_reference()


def test_whitening_transforms_are_the_reference_formula():
    """rls.whitening_transforms (eigen-decomposition and the two 4 x 4 products on the host, one upload) against
    train_region_refiner.py:63-67 written out with torch: T = V diag(1 / sqrt(ev + 0.001)) V', T_inv = V diag(sqrt(ev + 0.001)) V',
    T T_inv = I, T' = T — for a batch of covariances including a rank-deficient one (a class with a single row: S = 0)."""
    from odx.rls import whitening_transforms
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((5, 40, 4)) * np.array([0.3, 0.2, 0.1, 0.05])
    S = np.einsum("bni,bnj->bij", Y, Y) / 40
    S[4] = 0.0
    T, Ti = whitening_transforms(torch.from_numpy(S))
    assert T.dtype == torch.float64 and tuple(T.shape) == (5, 4, 4)
    for b in range(5):
        ev, V = torch.linalg.eigh(torch.from_numpy(S[b]))
        root = torch.sqrt(ev + 0.001)
        assert torch.allclose(T[b], V @ torch.diag(1.0 / root) @ V.t(), atol=1e-12)
        assert torch.allclose(Ti[b], V @ torch.diag(root) @ V.t(), atol=1e-12)
        assert torch.allclose(T[b] @ Ti[b], torch.eye(4, dtype=torch.float64), atol=1e-10) and torch.allclose(T[b], T[b].t(), atol=1e-14)


@pytest.mark.parametrize("device", DEVICES)
def test_region_refiner_class_runs_with_stray_labels(device):
    """The trainer finds the classes' rows from ONE sort of the labels and the positions of the class boundaries in it
    (rls.py::_train_batched: one host read).  Labels that belong to no regressor — negative ones, the background's 0, labels past
    the last class — must neither be trained on nor shift a class's run; a class without rows gets the reference's empty
    entry (train_region_refiner.py:38-44).  Checked against the same call on the strays-free rows, class by class."""
    from odx.rls import RegionRefinerTrainer
    rng = np.random.default_rng(21)
    D, sizes = 24, {1: 37, 2: 0, 3: 16, 4: 5}
    labels = np.concatenate([np.full(n, c) for c, n in sizes.items()] + [np.full(9, -1), np.full(7, 0), np.full(4, 9), np.full(3, -5)])
    X = rng.standard_normal((len(labels), D)).astype(np.float32)
    Y = (rng.standard_normal((len(labels), 4)) * 0.3).astype(np.float32)
    perm = rng.permutation(len(labels))
    labels, X, Y = labels[perm], X[perm], Y[perm]
    cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(5)}, "REGION_REFINER": {"opts": {"lambda": 2.0}}}

    def train(keep):
        coxy = {"C": torch.from_numpy(labels[keep].astype(np.float32)).view(-1, 1).to(device), "O": None,
                "X": torch.from_numpy(X[keep]).to(device), "Y": torch.from_numpy(Y[keep]).to(device)}
        return list(quiet(RegionRefinerTrainer(cfg, 2.0, False), coxy))

    got = train(np.ones(len(labels), dtype=bool))
    want = train((labels >= 1) & (labels <= 4))
    assert len(got) == len(want) == 4
    for c, (a, b) in enumerate(zip(got, want), start=1):
        assert (a["Beta"] is None) == (b["Beta"] is None) == (sizes[c] == 0), c
        if a["Beta"] is None:
            continue
        for k in range(4):
            wa, wb = a["Beta"][str(k)]["weights"], b["Beta"][str(k)]["weights"]
            assert float((wa - wb).abs().max()) <= 1e-6 * max(1.0, float(wb.abs().max())), (c, k)
            assert a["Beta"][str(k)]["losses"].shape == (sizes[c],)
            assert torch.allclose(a["Beta"][str(k)]["losses"], b["Beta"][str(k)]["losses"], atol=1e-6), (c, k)


# ------------------------------------------------------------------ f4: adding a class to a running pipeline
@pytest.mark.parametrize("device", DEVICES)
def test_add_a_class_without_touching_the_others(device):
    """Train two classes, put them in the test-time head, then train a third with one more FALKON + RLS fit, append
    and update_model: the first two classes score and regress exactly as before, the new column is the new model's
    own stand-alone prediction (demo contract: predictor_online_segmentation.py:404-425, box_head_getProposals.py:90-99)."""
    from odx.extract import OnlineDetectionModel
    from odx.harvest import DetectorHarvester
    from odx.rls import RegionRefinerTrainer
    from odx.wrappers import CenterSelector
    D = 24
    rng = np.random.default_rng(11)
    mus = rng.standard_normal((3, D)) * 2

    def rows(c, n):
        return torch.from_numpy((mus[c] + rng.standard_normal((n, D))).astype(np.float32)).to(device)

    def fit_class(c):
        X = torch.cat([rows(c, 60), rows((c + 1) % 3, 90), rows((c + 2) % 3, 90)])
        y = torch.cat([torch.ones(60), -torch.ones(180)]).to(device)
        m = odx.InCoreFalkon(kernel=odx.GaussianKernel(6.0), penalty=1e-4, M=80, maxiter=20,
                             center_selection=CenterSelector(list(range(0, 240, 3))))
        m.fit(X, y)
        return m

    def fit_regressors(classes):
        Xr = torch.cat([rows(c, 40) for c in classes])
        Cr = torch.cat([torch.full((40, 1), float(k + 1)) for k in range(len(classes))]).to(device)
        Yr = torch.from_numpy(rng.standard_normal((len(Xr), 4)).astype(np.float32)).to(device) * 0.1
        cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(len(classes) + 1)}, "REGION_REFINER": {"opts": {"lambda": 1.0}}}
        return list(quiet(RegionRefinerTrainer(cfg, 1.0, False), {"C": Cr, "O": None, "X": Xr, "Y": Yr}))

    stats = {"mean": torch.zeros(D, device=device), "std": torch.ones(D, device=device), "mean_norm": torch.tensor(20.0, device=device)}
    clfs, regs = [fit_class(0), fit_class(1)], fit_regressors([0, 1])
    model = OnlineDetectionModel(width=8).to(device)
    model.update_model(models_detection={"classifiers": clfs, "regressors": regs, "stats": stats})
    F = torch.cat([rows(0, 5), rows(1, 5), rows(2, 5)])
    s2, d2 = model.online_box(F)
    assert tuple(s2.shape) == (15, 3) and tuple(d2.shape) == (15, 12)
    # a harvester grows by one class the same way
    hv = DetectorHarvester(D, 2, 2, 10, 4, device=device)
    hv.add_new_class()
    assert hv.num_classes == 3 and len(hv._neg) == 3 and hv.still_to_complete == [0, 1, 2]
    # one more FALKON + RLS fit, appended
    new_clf, new_reg = fit_class(2), fit_regressors([2])
    model.update_model(models_detection={"classifiers": clfs + [new_clf], "regressors": regs + new_reg, "stats": stats})
    s3, d3 = model.online_box(F)
    assert tuple(s3.shape) == (15, 4) and tuple(d3.shape) == (15, 16)
    assert torch.equal(s3[:, :3], s2) and torch.equal(d3[:, :12], d2)
    Fn = (F - stats["mean"]) * (20.0 / stats["mean_norm"])
    assert s3.device.type == device
    assert torch.allclose(s3[:, 3], new_clf.predict(Fn).squeeze(1).float(), atol=1e-5)
