"""The drop-in modules on the MI355X, through libodx.so, against (a) golden vectors produced by
the reference's own code and (b) the same host logic driven by the f64 oracle."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests import dropin  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quiet(fn, *a, **kw):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


@pytest.fixture(scope="module", autouse=True)
def hip_backend():
    import odx
    odx.set_backend(None)
    be = odx.get_backend()
    assert be.name == "hip-gfx950"
    yield be


def test_smoke_entry():
    import __graft_entry__ as g
    quiet(g.smoke)


@pytest.mark.parametrize("tag,is_rpn", [("det", False), ("rpn", True)])
def test_region_refiner_on_gpu_matches_reference(tag, is_rpn, tmp_path):
    import yaml
    R = np.load(os.path.join(GOLD, "rls_golden.npz"))
    cfg = {"CHOSEN_CLASSES": {i: str(c) for i, c in enumerate(R["classes"])}, "REGION_REFINER": {"opts": {"lambda": float(R["lambda"])}}}
    if is_rpn:
        cfg = {"RPN": cfg}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    rr = dropin.load("region_refiner").RegionRefiner(path, is_rpn=is_rpn)
    C = torch.from_numpy(R["C"] if not is_rpn else R["C"] - 1).cuda()
    models = quiet(rr.trainRegionRefiner, {"C": C, "O": None, "X": torch.from_numpy(R["X"]).cuda(), "Y": torch.from_numpy(R["Y"]).cuda()})
    assert len(models) == int(R[tag + "_num_models"])
    for i, m in enumerate(models):
        if bool(R["%s_%d_none" % (tag, i)]):
            assert m["Beta"] is None
            continue
        assert m["mu"].is_cuda and m["Beta"]["0"]["weights"].is_cuda
        for key in ("mu", "T", "T_inv"):
            assert np.allclose(m[key].cpu().numpy(), R["%s_%d_%s" % (tag, i, key)], atol=2e-6)
        W = np.stack([m["Beta"][str(k)]["weights"].cpu().numpy() for k in range(4)])
        assert np.abs(W - R["%s_%d_W" % (tag, i)]).max() < 2e-6
        L = np.stack([m["Beta"][str(k)]["losses"].cpu().numpy() for k in range(4)])
        assert np.abs(L - R["%s_%d_losses" % (tag, i)]).max() < 1e-5
    # RegionRefinerTrainer.solve on its own (train_region_refiner.py:100-119) on HIP tensors: the reference's inputs of one
    # class (f64 rows + bias column, whitened targets) -> that class's golden weights and losses
    from odx.rls import RegionRefinerTrainer, whiten_targets
    tr = RegionRefinerTrainer({"CHOSEN_CLASSES": {}}, float(R["lambda"]), is_rpn)
    i = next(k for k in range(len(models)) if not bool(R["%s_%d_none" % (tag, k)]))
    rows = (C.reshape(-1) == (i if is_rpn else i + 1)).nonzero().reshape(-1)
    Xi = torch.cat((torch.from_numpy(R["X"]).cuda()[rows].double(), torch.ones((len(rows), 1), dtype=torch.float64, device="cuda")), dim=1)
    mu, Yc, T, _ = whiten_targets(torch.from_numpy(R["Y"]).cuda()[rows].double())
    beta = tr.solve(Xi, Yc @ T, float(R["lambda"]))
    assert np.abs(np.stack([beta[str(k)]["weights"].cpu().numpy() for k in range(4)]) - R["%s_%d_W" % (tag, i)]).max() < 2e-6
    assert np.abs(np.stack([beta[str(k)]["losses"].cpu().numpy() for k in range(4)]) - R["%s_%d_losses" % (tag, i)]).max() < 1e-5
    if not is_rpn:
        from odx.boxlist import BoxList
        cfg3 = {"CHOSEN_CLASSES": {0: "_background_", 1: "a", 2: "b"}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
        p3 = str(tmp_path / "cfg3.yaml")
        yaml.safe_dump(cfg3, open(p3, "w"))
        rr3 = dropin.load("region_refiner").RegionRefiner(p3)
        boxes = [BoxList(torch.from_numpy(R["apply_boxes_%d" % im]), (320, 240)) for im in range(2)]
        feats = [{"feat": R["apply_feat_%d" % im], "gt": R["apply_gt_%d" % im]} for im in range(2)]
        res = rr3.predict(boxes, feats, models=models[:2])
        for im in range(2):
            assert np.abs(res[im].bbox.cpu().numpy() - R["apply_out_%d" % im]).max() < 2e-3


def test_rls_large_d_against_oracle(hip_backend):
    """D = 2048 (the detector's feature size): D + 1 = 2049 exercises ragged tiles everywhere."""
    from oracle import rls_ref
    rng = np.random.default_rng(0)
    n, D = 3000, 2048
    X = (rng.standard_normal((n, D)) * 0.5 + 0.2).astype(np.float32)
    Y = (rng.standard_normal((n, 4)) * 0.2).astype(np.float32)
    C = np.ones((n, 1), np.float32)
    from odx.rls import RegionRefinerTrainer
    cfg = {"CHOSEN_CLASSES": {0: "bg", 1: "a"}, "REGION_REFINER": {"opts": {"lambda": 1000.0}}}
    models = quiet(RegionRefinerTrainer(cfg, 1000.0, False), {"C": torch.from_numpy(C).cuda(), "O": None,
                                                               "X": torch.from_numpy(X).cuda(), "Y": torch.from_numpy(Y).cuda()})
    ref = rls_ref.train_class(X, Y, 1000.0)
    W = np.stack([models[0]["Beta"][str(k)]["weights"].cpu().numpy() for k in range(4)])
    assert np.abs(W - ref["W"]).max() < 1e-6 * max(1.0, np.abs(ref["W"]).max())


@pytest.mark.parametrize("D", [70, 72, 328])
def test_batched_rls_equals_the_class_by_class_loop(hip_backend, D):
    """RegionRefinerTrainer trains all classes with one launch chain (one Gram launch over the classes' row segments,
    batched Cholesky / inverses / triangular products); per class the result must be the class-by-class loop's (the
    reference's order of work) — ragged class sizes, a class without rows, more classes than one batch.  D = 70: the Grams
    through the transposed f64 copy and the NT GEMM (D % 8 != 0); 72 and 328: straight from the f32 rows
    (rls_gram_rows_kernel), one ragged tile / three tile rows with ragged last tiles on both sides."""
    from odx.rls import RegionRefinerTrainer
    rng = np.random.default_rng(5)
    C = 35                                                # 35 classes: two batches (32 + 3)
    sizes = [int(v) for v in rng.integers(1, 400, C)]
    sizes[3], sizes[20] = 0, 17
    X = torch.from_numpy(rng.standard_normal((sum(sizes), D)).astype(np.float32) * 0.5 + 0.1).cuda()
    Y = torch.from_numpy(rng.standard_normal((sum(sizes), 4)).astype(np.float32) * 0.2).cuda()
    Cl = torch.from_numpy(np.repeat(np.arange(1, C + 1), sizes).astype(np.float32)).cuda()
    perm = torch.from_numpy(rng.permutation(sum(sizes))).cuda()
    COXY = {"C": Cl[perm].view(-1, 1), "O": None, "X": X[perm], "Y": Y[perm]}
    cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
    tr = RegionRefinerTrainer(cfg, 10.0, False)
    tr.COXY = COXY
    got = quiet(tr._train_batched, hip_backend)
    ref = quiet(tr._train_sequential, hip_backend)
    assert len(got) == len(ref) == C
    for c, (a, b) in enumerate(zip(got, ref)):
        assert (a["Beta"] is None) == (b["Beta"] is None) == (sizes[c] == 0), c
        if a["Beta"] is None:
            continue
        for key in ("mu", "T", "T_inv"):
            assert torch.allclose(a[key], b[key], atol=1e-6), (c, key)
        for k in range(4):
            wa, wb = a["Beta"][str(k)]["weights"], b["Beta"][str(k)]["weights"]
            assert float((wa - wb).abs().max()) <= 1e-6 * max(1.0, float(wb.abs().max())), (c, k)
            assert torch.allclose(a["Beta"][str(k)]["losses"], b["Beta"][str(k)]["losses"], atol=1e-6), (c, k)


def test_rls_grams_from_rows_equal_the_transposed_copy_form(hip_backend, monkeypatch):
    """The two forms of the batched Gram step (the route of D % 8 != 0, forced here by the library's test hook
    rls_force_nt_gram: transposed f64 copy + NT GEMM; default: straight from the f32 rows) sum the same f64 products in different orders, and so do the two forms of the solves (block substitution with the
    Cholesky factor; its explicit inverse): the regressors agree to rounding, for class sizes that are not multiples of the
    16-row k-tile and feature counts that leave ragged tiles (D + 1 = 457: four 128-blocks, the last one 73 rows)."""
    from odx.rls import RegionRefinerTrainer
    rng = np.random.default_rng(11)
    D, C = 456, 5
    sizes = [1, 15, 16, 17, 1000]
    X = torch.from_numpy(rng.standard_normal((sum(sizes), D)).astype(np.float32) * 0.5 + 0.1).cuda()
    Y = torch.from_numpy(rng.standard_normal((sum(sizes), 4)).astype(np.float32) * 0.2).cuda()
    Cl = torch.from_numpy(np.repeat(np.arange(1, C + 1), sizes).astype(np.float32)).cuda()
    perm = torch.from_numpy(rng.permutation(sum(sizes))).cuda()
    COXY = {"C": Cl[perm].view(-1, 1), "O": None, "X": X[perm], "Y": Y[perm]}
    cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
    out = {}
    import odx
    try:
        for mode in ("nt", "rows", "rows+inverse"):
            odx.options.library_hook("rls_force_nt_gram", 1 if mode == "nt" else 0)
            # the solves: block substitution with the factor (default) / the explicit inverse and triangular products
            odx.options.library_hook("rls_force_inverse_solve", 1 if mode.endswith("inverse") else 0)
            tr = RegionRefinerTrainer(cfg, 10.0, False)
            tr.COXY = COXY
            out[mode] = quiet(tr._train_batched, hip_backend)
    finally:
        odx.options.library_hook("rls_force_nt_gram", 0)
        odx.options.library_hook("rls_force_inverse_solve", 0)
    for c in range(C):
        for k in range(4):
            wb = out["rows"][c]["Beta"][str(k)]["weights"]
            for other in ("nt", "rows+inverse"):
                wa = out[other][c]["Beta"][str(k)]["weights"]
                assert float((wa - wb).abs().max()) <= 1e-6 * max(1.0, float(wb.abs().max())), (other, c, k)


def test_rls_batched_predictions_equal_the_one_class_kernel(hip_backend):
    """odx_rls_predict_rows_batched_f64 (a wave takes four consecutive rows of ONE class and reads the class's weight rows once per
    column chunk for all of them) = odx_rls_predict_rows_f64 class by class to f64 rounding (the two kernels add a row's products
    in different orders: 1e-14 at D = 1024): class sizes that are not multiples of four, an empty class in the middle and at the
    end, a feature count with a ragged last float4 (D = 70)."""
    from odx.backend import Features
    rng = np.random.default_rng(17)
    for D in (70, 72, 1024):
        sizes = [5, 0, 16, 33, 1, 7, 0]
        n = sum(sizes)
        X = torch.from_numpy(rng.standard_normal((n + 11, D)).astype(np.float32)).cuda()
        F = hip_backend.features(X)
        ld = (D + 2) // 2 * 2
        W = torch.from_numpy(rng.standard_normal((len(sizes), 4, ld))).cuda()
        rows = torch.from_numpy(rng.permutation(n + 11)[:n]).cuda()
        starts = [int(v) for v in np.concatenate(([0], np.cumsum(sizes)[:-1]))]
        got = torch.empty((n, 4), dtype=torch.float64, device="cuda")
        hip_backend.rls_predict_rows_batched(F, rows, starts, W, got)
        for c, (s0, k) in enumerate(zip(starts, sizes)):
            if k:
                want = hip_backend.rls_predict_rows(F, rows[s0:s0 + k].contiguous(), W[c])
                assert float((got[s0:s0 + k] - want).abs().max()) <= 1e-13 * max(1.0, float(want.abs().max())), (D, c)


def test_rls_pad_index_equals_the_tensor_statements(hip_backend):
    """odx_rls_pad_index (the padded row-id array of a class batch and its inverse maps, one launch) against the tensor
    statements it replaced in rls.py: ragged classes, an empty one in the middle and at the end, segments padded to 16."""
    seg_len = [5, 0, 16, 33, 1, 0]
    seg_off, at = [], 0
    for n in seg_len:
        seg_off.append(at)
        at += (n + 15) // 16 * 16
    npad, total = at, sum(seg_len)
    g = torch.Generator().manual_seed(2)
    run = torch.randperm(1000, generator=g)[:total].cuda()
    idx_pad, gid, pos, dest, lens = hip_backend.rls_pad_index(run, seg_off, seg_len, npad)
    lens_h = torch.tensor(seg_len)
    gid_w = torch.repeat_interleave(torch.arange(len(seg_len)), lens_h)
    starts = torch.tensor(np.concatenate(([0], np.cumsum(seg_len)[:-1])))
    pos_w = torch.arange(total) - starts[gid_w]
    dest_w = torch.tensor(seg_off)[gid_w] + pos_w
    idx_w = torch.full((npad,), -1, dtype=torch.int64)
    idx_w[dest_w] = run.cpu()
    assert torch.equal(idx_pad.cpu(), idx_w) and torch.equal(gid.cpu(), gid_w) and torch.equal(pos.cpu(), pos_w)
    assert torch.equal(dest.cpu(), dest_w) and lens.cpu().tolist() == seg_len
    with pytest.raises(RuntimeError):                     # lengths that do not add up to the ids handed over
        hip_backend.rls_pad_index(run[:-1], seg_off, seg_len, npad)


def test_rls_raw_target_products_equal_the_second_sweep(hip_backend, monkeypatch):
    """The whitened targets' X' Yw from the RAW targets' products formed inside the Gram sweep, (X' Y - X' 1 mu') T
    (odx_rls_gram_raw_batched_f64 + odx_rls_fold_whitened_f64: the default with f32 targets), against the second sweep over the
    rows with the whitened targets (odx.rls.RAW_TARGETS = False): the same f64 sums in another order.  Targets with means far above
    their spread (the subtraction cancels four digits), ragged class sizes, a ragged last tile; f64 targets keep the two-sweep
    form (nothing may be rounded to f32 on the way)."""
    from odx.rls import RegionRefinerTrainer
    rng = np.random.default_rng(12)
    D, C = 328, 6
    sizes = [1, 15, 16, 17, 700, 1300]
    X = torch.from_numpy(rng.standard_normal((sum(sizes), D)).astype(np.float32) * 0.5 + 0.1).cuda()
    Y = torch.from_numpy((rng.standard_normal((sum(sizes), 4)) * 0.05 + np.array([40.0, -25.0, 3.0, 0.0])).astype(np.float32)).cuda()
    Cl = torch.from_numpy(np.repeat(np.arange(1, C + 1), sizes).astype(np.float32)).cuda()
    perm = torch.from_numpy(rng.permutation(sum(sizes))).cuda()
    cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}, "REGION_REFINER": {"opts": {"lambda": 10.0}}}
    calls = []
    real = hip_backend.rls_gram_raw_begin

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    monkeypatch.setattr(hip_backend, "rls_gram_raw_begin", spy)
    out = {}
    for mode, ydt in (("raw", torch.float32), ("sweep", torch.float32), ("f64", torch.float64)):
        monkeypatch.setattr("odx.rls.RAW_TARGETS", mode != "sweep")
        tr = RegionRefinerTrainer(cfg, 10.0, False)
        tr.COXY = {"C": Cl[perm].view(-1, 1), "O": None, "X": X[perm], "Y": Y[perm].to(ydt)}
        n0 = len(calls)
        out[mode] = quiet(tr._train_batched, hip_backend)
        assert (len(calls) - n0 == 1) == (mode == "raw"), mode
    for c in range(C):
        for k in range(4):
            wb = out["sweep"][c]["Beta"][str(k)]["weights"]
            for other in ("raw", "f64"):
                wa = out[other][c]["Beta"][str(k)]["weights"]
                assert float((wa - wb).abs().max()) <= 1e-6 * max(1.0, float(wb.abs().max())), (other, c, k)
                assert torch.allclose(out[other][c]["Beta"][str(k)]["losses"], out["sweep"][c]["Beta"][str(k)]["losses"], atol=1e-6), (other, c, k)


class OracleFalkonClassifier:
    """Test-side classifier plug-in: the reference wrapper's index rule + the f64 oracle fit."""

    def __init__(self, M, maxiter=20):
        self.M, self.maxiter = M, maxiter
        self.sizes = []

    def train(self, X, y, sigma=None, lam=None):
        from oracle import falkon_ref as fr
        Xn, yn = X.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64)
        idx = fr.compute_indices_selection(yn, self.M, lambda high, size: torch.randint(high, (size,)).numpy())
        if isinstance(idx, int):
            idx = [idx]
        alpha, Z = fr.falkon_fit(Xn, yn, idx, sigma, lam, maxiter=self.maxiter, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        self.sizes.append((int((yn == 1).sum()), int((yn == -1).sum())))
        return {"alpha": alpha, "Z": Z, "sigma": sigma}

    def predict(self, model, X, y=None):
        from oracle import falkon_ref as fr
        p = fr.falkon_predict(X.cpu().numpy().astype(np.float64), model["Z"], model["alpha"], model["sigma"])
        return torch.from_numpy(p).float().to(X.device)


def test_minibootstrap_with_falkon_on_gpu_matches_oracle_pipeline(tmp_path):
    """Reference-regime training: 3 classes x 4 negative batches, FALKON on the GPU inside the
    hard/easy-negative loop, against the identical loop with the f64 oracle plugged in."""
    import yaml
    D, C, ITER, M = 64, 3, 4, 120
    classes = ["_background_", "a", "b", "c"]
    cfg = {"NUM_CLASSES": 4, "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                                          "CLASSIFIER": {"lambda": 0.001, "sigma": 8, "M": M, "kernel_type": "gauss"}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(classes)}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    g = torch.Generator().manual_seed(17)
    mus = torch.randn(C, D, generator=g) * 1.2

    def data():
        gg = torch.Generator().manual_seed(18)
        pos, neg = [], []
        for c in range(C):
            npos = [150, 0, 90][c]
            pos.append((mus[c] + 0.6 * torch.randn(npos, D, generator=gg)).cuda() if npos else torch.empty((0, D)).cuda())
            neg.append([(mus[(c + 1 + j % 2) % C] * (0.4 + 0.15 * j) + 0.9 * torch.randn(200, D, generator=gg)).cuda() for j in range(ITER)])
        return pos, neg

    stats = {"mean": torch.zeros(D).cuda(), "std": torch.ones(D).cuda(), "mean_norm": torch.tensor(8.0).cuda()}
    wrapper = dropin.load("FALKONWrapper_with_centers_selection_incore").FALKONWrapper(cfg_path=path)
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    pos, neg = data()
    torch.manual_seed(5)
    models = quiet(orc_mod.OnlineRegionClassifier(wrapper, pos, neg, stats, cfg_path=path).trainRegionClassifier,
                   output_dir=str(tmp_path))
    pos2, neg2 = data()
    oc = OracleFalkonClassifier(M)
    torch.manual_seed(5)
    ref_models = quiet(orc_mod.OnlineRegionClassifier(oc, pos2, neg2, stats, cfg_path=path).trainRegionClassifier)
    assert [m is None for m in models] == [False, True, False] == [m is None for m in ref_models]
    for m, r in zip(models, ref_models):
        if m is None:
            continue
        assert m.M == r["Z"].shape[0] and tuple(m.alpha_.shape) == (m.M, 1)
        assert np.allclose(m.ny_points_.cpu().numpy(), r["Z"], atol=1e-6)        # same centres => same selection trace
        a = m.alpha_.cpu().numpy()
        assert np.linalg.norm(a - r["alpha"]) / np.linalg.norm(r["alpha"]) < 1e-4
    assert "Detector's Online Classifier training time" in open(os.path.join(str(tmp_path), "result.txt")).read()
    # stand-alone scoring and the model-list device helper
    u = dropin.load("py_od_utils")
    cpu_models = [None if m is None else m for m in models]
    for m in cpu_models:
        if m is not None:
            m.alpha_, m.ny_points_ = m.alpha_.cpu(), m.ny_points_.cpu()
    moved = u.falkon_models_to_cuda(cpu_models)
    assert moved[0].alpha_.is_cuda and moved[0].ny_points_.is_cuda


def test_falkon_cpu_variant_returns_host_model(tmp_path):
    import yaml
    cfg = {"ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                        "CLASSIFIER": {"lambda": 0.001, "sigma": 6, "M": 50}}, "CHOSEN_CLASSES": {0: "bg", 1: "a"}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    w = dropin.load("FALKONWrapper_with_centers_selection").FALKONWrapper(cfg_path=path)
    from tests.synth import blob_problem
    X, y, _ = blob_problem(400, 32, seed=2)
    torch.manual_seed(0)
    m = w.train(torch.from_numpy(X), torch.from_numpy(y))
    assert not m.alpha_.is_cuda and not m.ny_points_.is_cuda and m.M == 50
    p = w.predict(m, torch.from_numpy(X[:9]))
    assert tuple(p.shape) == (9, 1) and not p.is_cuda


def test_edge_cases(hip_backend):
    """Single centre, centres fewer than M, duplicate centres, all-same labels are all legal."""
    import odx
    from oracle import falkon_ref as fr
    from tests.synth import blob_problem
    be = hip_backend
    X, y, rng = blob_problem(500, 36, seed=12)
    F = be.features(torch.from_numpy(X))
    # duplicate centres (sampling with replacement) => singular K_MM, the jitter keeps chol alive
    idx = [3, 3, 3, 10, 10, 42, 99, 99, 7, 8, 9, 11]
    Zf = be.rows(F, idx)
    alpha = odx.falkon_fit(be, F, be.vec(y), Zf, 6.0, 1e-3, 20)
    ref, Z = fr.falkon_fit(X.astype(np.float64), y, idx, 6.0, 1e-3, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
    pred = be.mmv(F, Zf, 6.0, alpha).cpu().numpy()
    assert np.abs(pred - fr.falkon_predict(X.astype(np.float64), Z, ref, 6.0)).max() < 1e-4
    # a single centre
    Z1 = be.rows(F, [5])
    a1 = odx.falkon_fit(be, F, be.vec(y), Z1, 6.0, 1e-3, 20)
    r1, _ = fr.falkon_fit(X.astype(np.float64), y, [5], 6.0, 1e-3, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
    assert abs(float(a1[0]) - r1[0, 0]) < 1e-6 * max(1.0, abs(r1[0, 0]))
    # empty prediction batch
    assert tuple(be.mmv(be.features(torch.zeros(0, 36)), Zf, 6.0, alpha).shape) == (0, 1)


def test_heads_on_gpu_match_reference(hip_backend):
    """Batched FALKON scoring + folded RLS GEMM against the reference's own head code outputs."""
    from tests.test_host_logic import check_heads_against_reference
    check_heads_against_reference(atol=2e-4)


def test_heads_at_reference_scale(hip_backend):
    """300 RoIs x D=2048 against 30 classes x M=2000 (the detector's test-time shape)."""
    from odx.heads import OnlineBoxPredictor
    from oracle import falkon_ref as fr
    import odx
    rng = np.random.default_rng(0)
    D, C, M, R = 2048, 30, 2000, 300

    class Mdl:
        pass
    cls = []
    for c in range(C):
        if c == 7:
            cls.append(None)
            continue
        m = Mdl()
        m.ny_points_ = torch.from_numpy((rng.standard_normal((M, D)) * (20 / D ** 0.5)).astype(np.float32))
        m.alpha_ = torch.from_numpy(rng.standard_normal((M, 1)))
        m.M, m.kernel = M, odx.GaussianKernel(20.0)
        cls.append(m)
    regs = np.array([{"mu": torch.zeros(4), "T": torch.eye(4), "T_inv": torch.eye(4),
                      "Beta": {str(k): {"weights": torch.from_numpy(rng.standard_normal(D + 1).astype(np.float32) * 0.01), "losses": None}
                               for k in range(4)}} for _ in range(C)], dtype=object)
    x = torch.from_numpy((rng.standard_normal((R, D)) * (20 / D ** 0.5)).astype(np.float32))
    sc, bb = OnlineBoxPredictor(cls, regs, None)(x)
    assert tuple(sc.shape) == (R, C + 1) and tuple(bb.shape) == (R, 4 * (C + 1))
    for c in (0, 12, 29):
        ref = fr.falkon_predict(x.numpy().astype(np.float64), cls[c].ny_points_.numpy().astype(np.float64), cls[c].alpha_.numpy(), 20.0)
        assert np.abs(sc[:, c + 1].cpu().numpy() - ref[:, 0]).max() < 1e-4 * max(1.0, np.abs(ref).max())
    assert torch.all(sc[:, 8] == 0) and torch.all(sc[:, 0] == -2)
    W = np.stack([regs[3]["Beta"][str(k)]["weights"].numpy() for k in range(4)]).astype(np.float64)
    refb = x.numpy().astype(np.float64) @ W[:, :-1].T + W[:, -1]
    assert np.abs(bb[:, 16:20].cpu().numpy() - refb).max() < 1e-4


def test_rccl_collectives_of_the_lockstep_fit_single_rank():
    """The RCCL calls of the sharded fit (odx/dist.py: all_gather_into_tensor / reduce_scatter_tensor / all_reduce /
    broadcast on f64 and f32 device tensors) in a one-rank process group on the real backend: same entry points, dtypes
    and shapes as an 8-GPU run, which this box cannot host.  Runs in a child process so the group never leaks into the
    other tests."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.getcwd(), "online-detection_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://")
from odx.dist import RowShard
sh = RowShard()
sh.enabled, sh.world, sh.rank = True, 1, 0            # one rank, but through the collective code path
M = 10001
Mp = (M + 1) // 2 * 2
mine = torch.randn(Mp, dtype=torch.float64, device="cuda")
out = torch.zeros((1, Mp), dtype=torch.float64, device="cuda")
sh.gather_rows(mine, out)
assert torch.equal(out[0], mine)
partials = torch.randn((1, Mp), dtype=torch.float64, device="cuda")
got = torch.zeros(Mp, dtype=torch.float64, device="cuda")
sh.reduce_scatter_rows(partials, got)
assert torch.equal(got, partials[0])
Z = torch.randn((1000, 1024), device="cuda")
assert torch.equal(sh.allreduce(Z.clone()), Z)
v = torch.randn(M, dtype=torch.float64, device="cuda")
assert torch.equal(sh.broadcast(v.clone(), src=0), v)
assert sh.total(12345) == 12345
# the centres' all-gather of equal-sized blocks (LockstepClassJob.gather_centres): (world, r, D) f32 from (r, D)
blk = torch.randn((137, 1024), device="cuda")
allb = torch.zeros((1, 137, 1024), device="cuda")
sh.gather_blocks(blk, allb)
assert torch.equal(allb[0], blk)
# ... and through the job itself: gather_centres on a one-rank nccl group goes down the collective path
import odx
from odx.job import LockstepClassJob
be = odx.get_backend()
X = torch.randn((500, 64), device="cuda")
idx = torch.tensor([499, 0, 0, 17, 250, 3], device="cuda")
job = LockstepClassJob(be, X, 500, 6, lambda c: None, [idx], 6.0, 1e-4, 20, shard=sh)
job.world = 1
Zf = job.gather_centres(idx) if False else None          # (world == 1 short-cuts; exercise the plan + gather by hand)
mine, cmax, slot, n_mine = job._centre_plan(idx)
b2 = torch.zeros((cmax, 64), device="cuda")
torch.index_select(X, 0, mine, out=b2[:n_mine])
a2 = torch.empty((1, cmax, 64), device="cuda")
sh.gather_blocks(b2, a2)
assert torch.equal(a2.view(cmax, 64).index_select(0, slot), X[idx])
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-OK")
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("ranks,classes,rows,centres,storage,batch", [(2, 4, 80000, 4200, "u24", 0), (3, 4, 40000, 1024, "auto", 0),
                                                                       (4, 7, 40000, 1024, "auto", 2), (2, 4, 80000, 4200, "u24", -1)])
def test_bench_starts_its_own_ranks_and_matches_the_oracle(ranks, classes, rows, centres, storage, batch):
    """`python bench.py --gpus 2` (the driver's form, no launcher around it): the parent starts the two ranks itself, the
    ranks shard the rows and run the lock-step fit with its per-iteration exchange (gloo here: both ranks share this
    box's one GPU), and the result line reports the rank count the collective saw plus the oracle check."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ODX_KNM"] = storage          # 2 ranks: the row shards stored as 24-bit fixed point, as the headline's are (compact passes, folded two-vector pass)
    # (3 ranks, 4 classes: lock-step batches of 3 and 1 — two ranks own nothing in the second batch; class-batched
    # preconditioner groups of different sizes per rank, identical collectives on all of them; 4 ranks, 7 classes, batches of 2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1",
                        "--single-device", "--dist-backend", "gloo", "--rows", str(rows), "--centres", str(centres), "--classes", str(classes),
                        "--check", "--no-cpu-baseline", "--no-extras"] + (["--lockstep-batch", str(batch)] if batch > 0 else [])
                       # batch = -1: the replicated one-all-reduce-per-iteration form (every rank builds every preconditioner,
                       # folded two-vector iteration included) with the real kernels
                       + (["--cg-exchange", "allreduce"] if batch < 0 else []),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    # (2 ranks, M = 4200, 40 000 rows per rank: the sharded fit folds its periodic full residual — two vectors per class in
    # that iteration's exchange)
    assert out["n_gpus"] == ranks and out["ranks"] == ranks and out["config"]["rows_per_gpu"] == rows // ranks + (1 if rows % ranks else 0)
    assert out["check"]["max_abs_score_diff_vs_oracle_predict"] < 1e-4
    # the sharded lock-step fit against the oracle's single-process fit (bench.py compares alpha up to N M = 1e9)
    assert out["check"]["alpha_rel_err_vs_oracle_fit"] < 1e-4
    # (4 ranks, lock-step batches of 2: the owners of consecutive batches rotate through the ranks — odx/plan.py — with the
    # real kernels and collectives under them)
    assert out["config"]["lockstep_batch"] == (1 if batch < 0 else (batch or ranks))
    assert out["config"]["cg_exchange"] == ("allreduce" if batch < 0 else "lockstep")
    assert out["health"]["failed_choleskys"] == 0 and out["health"]["ranks_with_nonfinite_scores"] == 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the first run of the lock-step fit's RCCL collectives "
                    "(all_gather_into_tensor / reduce_scatter_tensor) between real ranks; this pool's test boxes have one")
@pytest.mark.parametrize("form", ["lockstep", "allreduce"])
def test_bench_on_two_gpus_over_rccl_matches_the_oracle(form):
    """`python bench.py --gpus 2 --check` on the `nccl` backend, one rank per GPU — what the driver's scaling run starts, at
    the size the gloo form of this test uses on one GPU (80 000 rows, M = 4200, 4 classes, 24-bit shards, folded two-vector
    iteration).  Skips by itself on a one-GPU box; on the first box with two it is the first execution of the RCCL branches
    of odx/dist.py between real ranks.  `allreduce`: the replicated one-all-reduce-per-iteration form of the north star."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ODX_KNM"] = "u24"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "80000",
           "--centres", "4200", "--classes", "4", "--check", "--no-cpu-baseline", "--no-extras"]
    if form == "allreduce":
        cmd += ["--cg-exchange", "allreduce"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 40000
    assert out["check"]["max_abs_score_diff_vs_oracle_predict"] < 1e-4
    assert out["check"]["alpha_rel_err_vs_oracle_fit"] < 1e-4
    assert out["health"]["failed_choleskys"] == 0 and out["health"]["ranks_with_nonfinite_scores"] == 0


def test_batched_minibootstrap_equals_the_sequential_one_bit_for_bit(tmp_path, monkeypatch):
    """opts['class_batch'] = k fits all classes of a Minibootstrap round with one batched preconditioner launch chain and
    runs their K_nM builds / CG loops on k streams.  Fed the same Nystroem indices (the wrapper's index rule replaced by a
    deterministic one, so that the order in which classes consume the RNG does not matter) it must reproduce the
    reference-order sequential training bit for bit: same models, same caches — for k = 1 and k = 3, with a class without
    positives, a class with fewer negative batches and classes whose fits have different numbers of centres.  "b3g": the
    rounds in two groups (falkon.BatchFit: a chain per group, queued while the other group is prepared); "b3h": one group
    whose chain fit_batch splits into two half chains — both forced onto this small problem."""
    import yaml
    import odx.falkon as odx_falkon
    from odx.region_classifier import OnlineRegionClassifierBase
    D, C, ITER, M = 64, 5, 4, 120
    classes = ["_background_", "a", "b", "c", "d", "e"]
    cfg = {"NUM_CLASSES": 6, "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                                          "CLASSIFIER": {"lambda": 0.001, "sigma": 8, "M": M, "kernel_type": "gauss"}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(classes)}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    g = torch.Generator().manual_seed(41)
    mus = torch.randn(C, D, generator=g) * 1.2

    def data():
        gg = torch.Generator().manual_seed(42)
        pos, neg = [], []
        for c in range(C):
            npos = [150, 0, 90, 200, 20][c]                              # class 4: 20 positives + 60 negatives < M centres
            pos.append((mus[c] + 0.6 * torch.randn(npos, D, generator=gg)).cuda() if npos else torch.empty((0, D)).cuda())
            nbatch = ITER if c != 3 else ITER - 1
            nneg = 200 if c != 4 else 60
            neg.append([(mus[(c + 1 + j % 2) % C] * (0.4 + 0.15 * j) + 0.9 * torch.randn(nneg, D, generator=gg)).cuda() for j in range(nbatch)])
        return pos, neg

    stats = {"mean": torch.zeros(D).cuda(), "std": torch.ones(D).cuda(), "mean_norm": torch.tensor(8.0).cuda()}
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    wrap_mod = dropin.load("FALKONWrapper_with_centers_selection_incore")

    def injected(y):
        """<= M / 2 positives first, then negatives up to M, evenly spaced: the reference rule without its random draws."""
        pos, neg = (y == 1).nonzero().reshape(-1).tolist(), (y == -1).nonzero().reshape(-1).tolist()
        pos = pos[:: max(1, len(pos) // (M // 2))][: M // 2]
        room = M - len(pos)
        neg = neg[:: max(1, len(neg) // room)][:room]
        return pos + neg

    out = {}
    two_phase = []          # (mode, fast path kept, chains queued) of every BatchFit.finish
    plain_finish = odx_falkon.BatchFit.finish

    def recording_finish(self):
        two_phase.append((mode, self.fast, len(self.segments)))
        return plain_finish(self)
    monkeypatch.setattr(odx_falkon.BatchFit, "finish", recording_finish)
    for mode, opts in (("seq", {"return_caches": True}), ("b1", {"class_batch": 1, "return_caches": True}),
                       ("b3", {"class_batch": 3, "return_caches": True}), ("s3", {"class_streams": 3, "return_caches": True}),
                       ("b3g", {"class_batch": 3, "return_caches": True}), ("b3h", {"class_batch": 3, "return_caches": True})):
        monkeypatch.setattr(OnlineRegionClassifierBase, "GROUP_MIN_CLASSES", {"b3g": 1, "b3h": 99}.get(mode, 4))
        import odx
        monkeypatch.setattr(odx.options.current(), "chain_split_min", 1 if mode == "b3h" else 4)
        pos, neg = data()
        w = wrap_mod.FALKONWrapper(cfg_path=path)
        w.compute_indices_selection = injected
        torch.manual_seed(5)
        out[mode] = quiet(orc_mod.OnlineRegionClassifier(w, pos, neg, stats, cfg_path=path).trainRegionClassifier, opts=opts)
    # the two-group rounds went through the two-phase fit and kept its fast path (a chain per group), nobody else used it
    assert two_phase and {m for m, _, _ in two_phase} == {"b3g"} and all(fast and 1 <= n <= 2 for _, fast, n in two_phase)
    assert any(n == 2 for _, _, n in two_phase)
    (ms, cs) = out["seq"]
    assert [m is None for m in ms] == [False, True, False, False, False]
    assert ms[4].M < M                                                   # the ragged member of the batch
    for mode in ("b1", "b3", "s3", "b3g", "b3h"):  # the class-streams mode too: same per-class draws => same bits
        mb, cb = out[mode]
        for c in range(C):
            assert (ms[c] is None) == (mb[c] is None)
            if ms[c] is None:
                continue
            assert torch.equal(ms[c].ny_points_, mb[c].ny_points_), (mode, c)
            assert torch.equal(ms[c].alpha_, mb[c].alpha_), (mode, c, float((ms[c].alpha_ - mb[c].alpha_).abs().max()))
            assert torch.equal(cs[c]["neg"], cb[c]["neg"]) and torch.equal(cs[c]["pos"], cb[c]["pos"])


def test_minibootstrap_on_class_streams_does_not_depend_on_the_stream_count(tmp_path):
    """opts['class_streams'] = k trains the classes concurrently on k streams with one RNG stream per class: the models
    are the same bits for k = 1 and k = 3 (deterministic kernels, per-stream scratch), classes without data stay None,
    and every class follows the same state machine as the sequential mode (same cache sizes per class as a sequential
    run whose draws are replaced by that class's own stream is not observable here; the invariant checked is
    stream-count independence plus agreement of the scores with a sequentially trained model on held-out rows)."""
    import yaml
    D, C, ITER, M = 64, 4, 4, 120
    classes = ["_background_", "a", "b", "c", "d"]
    cfg = {"NUM_CLASSES": 5, "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                                          "CLASSIFIER": {"lambda": 0.001, "sigma": 8, "M": M, "kernel_type": "gauss"}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(classes)}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    g = torch.Generator().manual_seed(27)
    mus = torch.randn(C, D, generator=g) * 1.2

    def data():
        gg = torch.Generator().manual_seed(28)
        pos, neg = [], []
        for c in range(C):
            npos = [150, 0, 90, 200][c]
            pos.append((mus[c] + 0.6 * torch.randn(npos, D, generator=gg)).cuda() if npos else torch.empty((0, D)).cuda())
            nbatch = ITER if c != 3 else ITER - 1                      # a class with fewer negative batches
            neg.append([(mus[(c + 1 + j % 2) % C] * (0.4 + 0.15 * j) + 0.9 * torch.randn(200, D, generator=gg)).cuda() for j in range(nbatch)])
        return pos, neg

    stats = {"mean": torch.zeros(D).cuda(), "std": torch.ones(D).cuda(), "mean_norm": torch.tensor(8.0).cuda()}
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    wrap_mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    out = {}
    for k in (1, 3, 0, "seq_rng", "batch"):
        pos, neg = data()
        torch.manual_seed(5)
        orc = orc_mod.OnlineRegionClassifier(wrap_mod.FALKONWrapper(cfg_path=path), pos, neg, stats, cfg_path=path)
        opts = ({"class_rng": True, "return_caches": True} if k == "seq_rng" else {"class_batch": 2, "return_caches": True} if k == "batch"
                else {"class_streams": k, "return_caches": True} if k else None)
        out[k] = quiet(orc.trainRegionClassifier, opts=opts)
    (m1, c1), (m3, c3), ms = out[1], out[3], out[0]
    # the reference-order sequential loop with one RNG stream per class (opts['class_rng']) and the class-batch mode draw
    # what the streams mode draws: same models and caches, bit for bit, with the REAL index rule
    for other in ("seq_rng", "batch"):
        mo, co = out[other]
        for c in range(C):
            assert (m1[c] is None) == (mo[c] is None)
            if m1[c] is not None:
                assert torch.equal(m1[c].alpha_, mo[c].alpha_) and torch.equal(m1[c].ny_points_, mo[c].ny_points_), (other, c)
                assert torch.equal(c1[c]["neg"], co[c]["neg"]), (other, c)
    assert [m is None for m in m1] == [False, True, False, False] == [m is None for m in m3] == [m is None for m in ms]
    for c, (a, b, s) in enumerate(zip(m1, m3, ms)):
        if a is None:
            continue
        assert torch.equal(a.alpha_, b.alpha_) and torch.equal(a.ny_points_, b.ny_points_)
        # a different draw of the Nystroem centres than the sequential mode, the same classifier up to that: fresh rows of
        # the class score positive, fresh rows of another class negative, with both
        fresh_pos = ((mus[c] + 0.6 * torch.randn(200, D, generator=g)) * (20.0 / 8.0)).cuda()
        fresh_neg = ((mus[(c + 1) % C] * 0.4 + 0.9 * torch.randn(200, D, generator=g)) * (20.0 / 8.0)).cuda()
        for m in (a, s):
            assert float((m.predict(fresh_pos) > 0).float().mean()) > 0.9
            assert float((m.predict(fresh_neg) < 0).float().mean()) > 0.9
    for ca, cb in zip(c1, c3):
        if ca:
            assert torch.equal(ca["neg"], cb["neg"]) and torch.equal(ca["pos"], cb["pos"])


def test_class_sharded_minibootstrap_on_gpu(tmp_path):
    """opts['class_shard'] on the HIP backend: two ranks (one process each, sharing this box's GPU, gloo for the final
    model exchange) train the classes i % 2 == rank in the class-batch mode and both end with ALL models — bit for bit the
    models of one process in the same mode (per-class RNG streams; batched factors do not depend on the batch's members)."""
    import subprocess
    import socket
    from tests import dist_minibootstrap_worker as w
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "dist_minibootstrap_worker.py"), str(tmp_path)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = w.train(str(tmp_path / "cfg_ref.yaml"), {"class_batch": 2})
    assert [m is None for m in ref] == [False, True, False, False, False]
    for rank in range(2):
        got = torch.load(str(tmp_path / ("models_%d.pt" % rank)), weights_only=False)
        for c, (a, b) in enumerate(zip(ref, got)):
            assert (a is None) == (b is None), (rank, c)
            if a is not None:
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (rank, c)


@pytest.mark.parametrize("group_min", [4, 1])
@pytest.mark.parametrize("neg_rows,expect", [(200, "reference order, classes together"), (60, "reference order, class by class")])
def test_default_mode_keeps_the_reference_order_of_draws(tmp_path, monkeypatch, neg_rows, expect, group_min):
    """(group_min = 1: the rounds in two groups, as 8 classes and more run them.)  Without options the classes advance together AND every class draws its Nystroem centres from the global torch RNG
    exactly where the reference's class-by-class loop would: models, caches and the RNG state afterwards equal the forced
    class-by-class loop bit for bit (real index rule).  With 60-row negative batches class 4 has fewer negatives than room
    for negative centres, draws nothing, the prediction of the stream positions fails and the plain loop runs instead."""
    import yaml
    from odx.region_classifier import OnlineRegionClassifierBase
    monkeypatch.setattr(OnlineRegionClassifierBase, "GROUP_MIN_CLASSES", group_min)
    D, C, ITER, M = 64, 5, 4, 120
    cfg = {"NUM_CLASSES": 6, "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                                          "CLASSIFIER": {"lambda": 0.001, "sigma": 8, "M": M, "kernel_type": "gauss"}},
           "CHOSEN_CLASSES": {i: "c%d" % i for i in range(6)}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    g = torch.Generator().manual_seed(71)
    mus = torch.randn(C, D, generator=g) * 1.2

    def data():
        gg = torch.Generator().manual_seed(72)
        pos, neg = [], []
        for c in range(C):
            npos = [150, 0, 90, 200, 20][c]
            pos.append((mus[c] + 0.6 * torch.randn(npos, D, generator=gg)).cuda() if npos else torch.empty((0, D)).cuda())
            nneg = neg_rows if c == 4 else 200
            neg.append([(mus[(c + 1 + j % 2) % C] * (0.4 + 0.15 * j) + 0.9 * torch.randn(nneg, D, generator=gg)).cuda()
                        for j in range(ITER if c != 3 else ITER - 1)])
        return pos, neg

    stats = {"mean": torch.zeros(D).cuda(), "std": torch.ones(D).cuda(), "mean_norm": torch.tensor(8.0).cuda()}
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    wrap_mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    out = {}
    for mode, opts in (("auto", {"return_caches": True}), ("loop", {"return_caches": True, "reference_order": "sequential"})):
        pos, neg = data()
        torch.manual_seed(5)
        orc = orc_mod.OnlineRegionClassifier(wrap_mod.FALKONWrapper(cfg_path=path), pos, neg, stats, cfg_path=path)
        models, caches = quiet(orc.trainRegionClassifier, opts=opts)
        out[mode] = (models, caches, torch.get_rng_state(), orc.last_order)
    assert out["auto"][3] == expect and out["loop"][3] == "reference order, class by class"
    assert torch.equal(out["auto"][2], out["loop"][2])                      # the global stream ends where the loop leaves it
    for c in range(C):
        a, b = out["auto"][0][c], out["loop"][0][c]
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a.ny_points_, b.ny_points_) and torch.equal(a.alpha_, b.alpha_), c
            assert torch.equal(out["auto"][1][c]["neg"], out["loop"][1][c]["neg"]), c


@pytest.mark.parametrize("n,D1", [(101, 37), (64, 2), (513, 130)])
def test_rls_solve_on_a_general_design_matrix_odd_sizes(hip_backend, n, D1):
    """RegionRefinerTrainer.solve on a matrix that is NOT f32 features + a column of ones (f64 rows that are not f32-exact, no
    bias column): Gram, X'y, Cholesky + solves and predictions on the library's f64 kernels (odx_gemm_nt_f64, odx_rls_solve_f64)
    — against the dense normal equations in numpy, at odd n and odd D1 (the f64 GEMM's even-stride padding; advisor, round 5),
    with and without per-coordinate row subsets."""
    from odx.rls import RegionRefinerTrainer
    rng = np.random.default_rng(n * 1000 + D1)
    X = rng.standard_normal((n, D1)) * 0.7 + 0.1 / 3.0                      # f64, not representable in f32
    y = rng.standard_normal((n, 4)) * 0.3
    lam = 0.5
    tr = RegionRefinerTrainer({"CHOSEN_CLASSES": {}}, lam, False)
    for indices in (None, [np.sort(rng.choice(n, size=max(D1, n // 2) if n > D1 else n, replace=False)) for _ in range(4)]):
        out = quiet(tr.solve, torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda(), lam,
                    indices=None if indices is None else [torch.from_numpy(i) for i in indices])
        for k in range(4):
            I = np.arange(n) if indices is None else indices[k]
            A = X[I].T @ X[I] + lam * np.eye(D1)
            w = np.linalg.solve(A, X[I].T @ y[I, k])
            got = out[str(k)]["weights"].double().cpu().numpy()
            assert np.abs(got - w).max() < 2e-6 * max(1.0, np.abs(w).max()), (k, indices is None)        # (f32 hand-out of f64 weights)
            loss = 0.5 * (X[I] @ w - y[I, k]) ** 2
            assert np.abs(out[str(k)]["losses"].double().cpu().numpy() - loss).max() < 1e-5 * max(1.0, loss.max())
