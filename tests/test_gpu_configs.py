"""Parity at the shapes BASELINE.json names as configs 2, 4 and 5 and config 3's RLS half (config 1 is
`test_falkon_fit_alpha_parity`'s first case, config 3's single-GPU FALKON shard is `test_headline_shape_one_class`): the
whole fit on the MI355X against the f64 oracle (oracle/falkon_ref.py, oracle/rls_ref.py) on the same seeded inputs,
alphas within 1e-4 relative as BASELINE.json's north star states."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def be():
    import odx
    return odx.get_backend()


def _blobs(n, D, C, seed):
    """C class blobs, rows i of class i % C, normalised with the reference rule (OnlineRegionClassifier.py:224-227)."""
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((C, D))
    X = mu[np.arange(n) % C] + 0.7 * rng.standard_normal((n, D))
    X -= X.mean(0)
    X *= 20.0 / np.linalg.norm(X, axis=1).mean()
    return np.ascontiguousarray(X.astype(np.float32)), rng


def test_config2_thirty_classes_n1e5_m2000(be):
    """Config 2's FALKON half: 30 one-vs-rest classes on N = 1e5, D = 1024, M = 2000 (sigma = 15, lambda = 1e-5, 20 CG
    steps).  All 30 are fitted and scored on the GPU; two of them are checked against the oracle (alpha 1e-4, scores),
    all of them through the labels they were fitted on."""
    import odx
    from oracle import falkon_ref as fr
    n, D, M, C, sigma, lam = 100_000, 1024, 2000, 30, 15.0, 1e-5
    X, rng = _blobs(n, D, C, seed=1234 + 2)
    F = be.features(torch.from_numpy(X))
    cls = np.arange(n) % C
    scores = torch.empty((n, C), dtype=torch.float32, device="cuda")
    kept = {}
    for c in range(C):
        y = np.where(cls == c, 1.0, -1.0).astype(np.float32)
        idx = fr.compute_indices_selection(y, M, lambda high, size: rng.integers(0, high, size))
        Zf = be.rows(F, idx)
        alpha = odx.falkon_fit(be, F, be.vec(y), Zf, sigma, lam, 20)
        be.mmv(F, Zf, sigma, alpha, None, out=scores[:, c:c + 1])
        if c in (0, 17):
            kept[c] = (y, idx, alpha.cpu().numpy())
    pred = scores.argmax(1).cpu().numpy()
    assert (pred == cls).mean() > 0.99                      # separable blobs: every class learned
    Xd = X.astype(np.float64)
    for c, (y, idx, alpha) in kept.items():
        ref, Z = fr.falkon_fit(Xd, y.astype(np.float64), idx, sigma, lam, maxiter=20, dtype=np.float64, pc_eps=1e-5,
                               cg_epsilon=1e-7)
        rel = np.linalg.norm(alpha - ref[:, 0]) / np.linalg.norm(ref[:, 0])
        assert rel < 1e-4, (c, rel)
        rows = np.arange(0, n, 997)
        pref = fr.falkon_predict(Xd[rows], Z, ref, sigma)
        assert np.abs(scores[rows, c].cpu().numpy() - pref[:, 0]).max() < 1e-4


def test_config2_as_stated_bf16_stored_knm(be):
    """Config 2 AS STATED (K_nM stored as bf16; throughput only — bf16 entries move alpha by 1e-2 .. 6e-1 against the
    f32-accurate fit, tools/precision_storage_study.py) at its own size, N = 1e5, D = 1024, M = 2000: (i) the stored block is,
    entry for entry, the round-to-nearest-even bf16 of the f32 block the default build stores; (ii) the fit on it — fused
    right-hand side, the compact-format pass kernels, the class-batched chain and lock-step CG — gives the alpha the f64
    oracle gives when it iterates on that SAME rounded block (oracle/falkon_ref.falkon_fit(knm=...)): 1e-4 relative;
    (iii) scores of those alphas against the oracle's (the fused scoring kernel recomputes K at f32 accuracy, as
    FALKONWrapper.predict does whatever the fit stored)."""
    import odx
    from odx.solver import SolverOptions
    from oracle import falkon_ref as fr
    n, D, M, C, sigma, lam = 100_000, 1024, 2000, 30, 15.0, 1e-5
    X, rng = _blobs(n, D, C, seed=1234 + 2)
    F = be.features(torch.from_numpy(X))
    cls = np.arange(n) % C
    run = (0, 17)
    ys = [np.where(cls == c, 1.0, -1.0) for c in run]
    idx = [fr.compute_indices_selection(y, M, lambda high, size: rng.integers(0, high, size)) for y in ys]
    Zfs = [be.rows(F, i) for i in idx]
    old = be.knm_storage
    try:
        be.knm_storage = "f32"
        K32 = be.knm(F, Zfs[0], sigma).dense()
        want0 = K32.to(torch.bfloat16)
        del K32
        be.knm_storage = "bf16"
        assert be.knm_format(n, M) == "bf16"
        opt = SolverOptions(check_pivots=False)
        Ps = be.precond_batched(Zfs, sigma, lam, opt.pc_epsilon)
        b0s = torch.zeros((len(run), M), dtype=torch.float64, device="cuda")
        Ks = [be.knm_rhs(F, Zfs[k], sigma, torch.from_numpy(ys[k]).cuda() * (1.0 / n), rhs_out=b0s[k])[0] for k in range(len(run))]
        assert all(K.fmt == "bf16" for K in Ks)
        stored = [K.dense().cpu().numpy().astype(np.float64) for K in Ks]
        assert torch.equal(Ks[0].dense().to(torch.bfloat16), want0)                         # (i)
        assert be.cg_batched_supported([n] * len(run), [M] * len(run), "bf16")
        alphas = be.cg_solve_batched(Ks, Ps, b0s, [n] * len(run), lam, 20, opt).cpu().numpy()
        one = odx.falkon_fit(be, F, be.vec(ys[1].astype(np.float32)), Zfs[1], sigma, lam, 20, opt).cpu().numpy()     # a batch of one
    finally:
        be.knm_storage = old
    Xd = X.astype(np.float64)
    rows = np.arange(0, n, 997)
    for k in range(len(run)):
        ref, Z = fr.falkon_fit(Xd, ys[k], idx[k], sigma, lam, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7, knm=stored[k])
        rel = np.linalg.norm(alphas[k, :M] - ref[:, 0]) / np.linalg.norm(ref[:, 0])
        print("config 2, bf16-stored K_nM, class %d: alpha rel err vs the oracle on the same block %.2e" % (run[k], rel))
        assert rel < 1e-4, (run[k], rel)                                                    # (ii)
        if k == 1:
            assert np.linalg.norm(one - ref[:, 0]) / np.linalg.norm(ref[:, 0]) < 1e-4
        got = be.mmv(be.features(torch.from_numpy(X[rows])), Zfs[k], sigma, torch.from_numpy(alphas[k, :M])).cpu().numpy()
        pref = fr.falkon_predict(Xd[rows], Z, ref, sigma)
        assert np.abs(got - pref).max() < 1e-4 * max(1.0, float(np.abs(pref).max()))       # (iii)
    be.release_workspaces()
    torch.cuda.empty_cache()


def test_headline_kernels_alpha_at_m1e4(be):
    """alpha of the headline's OWN kernels at the headline's own width (round-3 review, item 1): bench.py's synthetic job
    (same generator, centre rule, sigma = 15, lambda = 1e-5, 20 CG steps) at N = 1e5, D = 1024, **M = 1e4**, storage left on
    `auto` — 24-bit fixed-point K_nM, gauss_knm_h2w256_kernel with the fused right-hand side, knm_passq_stag_kernel and the
    folded two-vector pass, the class-batched M = 1e4 preconditioner chain (one chain of 1 class, one of 2), the lock-step
    schedule of odx/job.py, gauss_mmv_h2w256_kernel — against oracle/falkon_ref.falkon_fit in f64 (8 GB of K_nM on the host):
    alpha within 1e-4 relative, scores within 1e-4.  The same classes on f32-stored blocks give the price of the 24-bit
    storage at this size as a number.  (tools/alpha_at_scale.py is this comparison at the full N = 1e6.)"""
    import bench
    from odx.job import LockstepClassJob
    from odx.solver import SolverOptions
    from oracle import falkon_ref as fr
    N, D, M, C, sigma, lam = 100_000, 1024, 10_000, 30, 15.0, 1e-5
    dev = be.device
    seed = 1234 + 3
    X = bench.synth_rows(0, N, D, C, seed, dev)
    cidx = bench.centre_indices(N, C, M, seed)
    row_ids = torch.arange(N, device=dev)
    labels = lambda c: torch.where((row_ids % C) == c, 1.0, -1.0).to(torch.float64)            # noqa: E731
    run = [0, 1, 2]
    got = {}
    for storage in ("auto", "f32"):
        prev, be.knm_storage = be.knm_storage, storage
        try:
            assert be.knm_format(N, M) == {"auto": "u24", "f32": "f32"}[storage]
            assert be.lib.odx_gauss_h2_tile(N, M) == 256
            infos, alphas = [], {}
            job = LockstepClassJob(be, X, N, M, labels, [torch.from_numpy(i).to(dev) for i in cidx], sigma, lam, 20,
                                   SolverOptions(check_pivots=False), classes=len(run))
            assert job.G == 3                                   # chains: [class 0], then [classes 1, 2] in one batched call
            job.run(be.features(X), run, infos=infos, alphas_out=alphas)
            torch.cuda.synchronize()
            assert [t for t in job.trace if t[0] == "precond"] == [("precond", (0,)), ("precond", (1, 2))]
            assert len(infos) == 3 and all(int(i.item()) == 0 for i in infos)
            got[storage] = ({c: alphas[c].cpu().numpy() for c in run}, job.scores[:, :len(run)].cpu().numpy())
            job.release()
        finally:
            be.knm_storage = prev
        be.release_workspaces()
        torch.cuda.empty_cache()
    Xd = X.cpu().numpy().astype(np.float64)
    srows = np.arange(0, N, 97)
    report = {}
    for c in (0, 2):                                            # one class of each chain
        y = labels(c).cpu().numpy()
        ref, Z = fr.falkon_fit(Xd, y, cidx[c], sigma, lam, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        pref = fr.falkon_predict(Xd[srows], Z, ref, sigma)[:, 0]
        for storage in ("auto", "f32"):
            a, sc = got[storage][0][c], got[storage][1][srows, c]
            report[(c, storage)] = (float(np.linalg.norm(a - ref[:, 0]) / np.linalg.norm(ref[:, 0])), float(np.abs(sc - pref).max()))
        del ref, Z
    print("alpha rel err / max score err at M = 1e4:", report)
    for key, (rel, serr) in report.items():
        assert rel < 1e-4 and serr < 1e-4, (key, rel, serr, report)


def test_headline_full_size_alpha_one_class(be):
    """One class of the headline job at its FULL size — N = 1e6, D = 1024, M = 1e4, storage `auto` (30 GB of 24-bit K_nM) —
    against oracle/falkon_ref.falkon_fit in f64 with the 80 GB K_nM stored on the host (tools/alpha_at_scale.py; ~1.5 min of
    host time on the GPU box's cores): alpha within 1e-4 relative, sampled scores within 1e-4.  Skipped where the host
    cannot hold the oracle's matrix."""
    from tools import alpha_at_scale
    if alpha_at_scale.host_free_bytes() < 130e9:
        pytest.skip("the f64 oracle at N = 1e6, M = 1e4 needs ~100 GB of host memory")
    res = alpha_at_scale.compare(1_000_000, 1024, 10_000, 30, classes_run=(0,), storages=("auto",))
    e = res["classes"]["0"]["auto"]
    print("full-size alpha:", res["classes"]["0"])
    assert e["stored_as"] == "u24"
    assert e["alpha_rel_err"] < 1e-4 and e["score_max_abs_err_sampled"] < 1e-4, e
    be.release_workspaces()
    torch.cuda.empty_cache()


def test_config4_mask_pixels_d256_m2000(be):
    """Config 4's shape: one class of the on-line segmentation head, 7 x 7 x 256 mask features as D = 256 pixel rows,
    M = 2000, sigma = 10, a single fit per class (no minibootstrap, run_experiment_online_rpn_ood_oos.py:254) — at
    2e5 rows, the size the f64 oracle holds on the host (the reference scale is 5e5 per class; nothing in the path
    depends on n beyond the row loop the headline-shape test covers)."""
    import odx
    from oracle import falkon_ref as fr
    n, D, M, sigma, lam = 200_000, 256, 2000, 10.0, 1e-5
    X, rng = _blobs(n, D, 2, seed=1234 + 4)
    y = np.where(np.arange(n) % 2 == 0, 1.0, -1.0).astype(np.float32)      # foreground / background pixels of the class
    flip = rng.random(n) < 0.05
    y[flip] = -y[flip]                                                      # label noise: not separable
    idx = fr.compute_indices_selection(y, M, lambda high, size: rng.integers(0, high, size))
    F = be.features(torch.from_numpy(X))
    Zf = be.rows(F, idx)
    alpha = odx.falkon_fit(be, F, be.vec(y), Zf, sigma, lam, 20).cpu().numpy()
    ref, Z = fr.falkon_fit(X.astype(np.float64), y.astype(np.float64), idx, sigma, lam, maxiter=20, dtype=np.float64,
                           pc_eps=1e-5, cg_epsilon=1e-7)
    rel = np.linalg.norm(alpha - ref[:, 0]) / np.linalg.norm(ref[:, 0])
    assert rel < 1e-4, rel
    rows = np.arange(0, n, 499)
    got = be.mmv(be.features(torch.from_numpy(X[rows])), Zf, sigma, torch.from_numpy(alpha)).cpu().numpy()
    assert np.abs(got - fr.falkon_predict(X[rows].astype(np.float64), Z, ref, sigma)).max() < 1e-4


def test_config4_at_its_own_size_5e5_rows(be):
    """Config 4 at the size BASELINE states — 5e5 mask-pixel rows per class, D = 256, M = 2000, sigma = 10 — where the f64
    oracle no longer fits a test's time: the size-independent properties of the headline-shape test (sampled K_nM entries and
    scores against the f64 oracle's entries, pass additivity / linearity over row splits, the fused right-hand side = one
    pass over the stored block, bitwise repeatability), beside the alpha < 1e-4 oracle case at 2e5 rows above.  D = 256 is
    the one BASELINE shape whose build is bound by the K store (4 GB per launch), not by the contraction."""
    from tests.test_gpu_kernels import full_size_properties
    full_size_properties(be, 500_000, 256, 2000, 10.0, 1e-5)
    be.release_workspaces()
    torch.cuda.empty_cache()


def test_config5_shard_shape_one_class(be):
    """Config 5 (100 classes, N = 5e6, D = 1024, M = 2e4 on 8 GPUs): the shard one GPU holds — 625 000 rows, a 50 GB f32
    K_nM block — for one class through the shipping f16-split path (f32-accurate K; the fp8 contraction BASELINE names is
    a throughput-only variant), with the size-independent properties of the headline-shape test: sampled
    K_nM entries and scores against the f64 oracle, pass additivity / linearity, the fused right-hand side, bitwise
    repeatability.  M = 2e4 is the widest configuration of the pass kernel and the largest preconditioner (four
    20 000^2 f64 factors)."""
    from tests.test_gpu_kernels import full_size_properties
    full_size_properties(be, 625_000, 1024, 20_000, 15.0, 1e-5)
    be.release_workspaces()
    torch.cuda.empty_cache()


def test_config5_shard_shape_fp8_contraction(be):
    """Config 5 AS STATED (fp8 e4m3 inputs to the X Z' MFMA, f32 accumulate; throughput only) at the shard one of its 8 GPUs
    holds: 625 000 x 20 000, D = 1024.  The 37.5 GB block the fp8 build stores and the fp8 fused scoring are compared on
    sampled rows with the numpy restatement of that arithmetic (operands scaled by the matrix's power of two and rounded once
    to e4m3, exact products, norms of the ROUNDED rows), plus the size-independent pass properties on the stored block."""
    from tests.test_gpu_kernels import _e4m3
    n, D, M, sigma = 625_000, 1024, 20_000, 15.0
    g = torch.Generator(device="cuda").manual_seed(11)
    X = torch.randn((n, D), generator=g, device="cuda") * (20.0 / D ** 0.5)
    idx = torch.arange(0, n, n // M, device="cuda")[:M]
    F = be.features(X)
    Zf = be.features(X.index_select(0, idx) + 0.05 * torch.randn((M, D), generator=g, device="cuda"))
    old, oldk = be.gauss, be.knm_storage
    try:
        be.gauss = "f8"
        assert be.knm_format(n, M) == "u24"
        w = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        K, ktw = be.knm_rhs(F, Zf, sigma, w)
        sx, sz = float(F.meta8[0]), float(Zf.meta8[0])
        rows = np.array([0, 1, 255, 256, 4097, n // 2, n - 257, n - 1])
        cols = np.unique(np.concatenate([np.arange(0, 600), np.arange(M // 2 - 130, M // 2 + 130), np.arange(M - 300, M)]))
        Xq = _e4m3(X[rows].cpu().numpy().astype(np.float64) * sx)
        Zq = _e4m3(Zf.X.cpu().numpy().astype(np.float64) * sz)
        sqx = ((Xq / sx) ** 2).sum(1)[:, None]
        sqz = ((Zq / sz) ** 2).sum(1)[None, :]
        Kq = np.exp(-np.maximum(sqx + sqz - 2.0 * (Xq @ Zq.T) / (sx * sz), 0.0) / (2 * sigma ** 2))
        got = torch.stack([K.rows(int(r), int(r) + 1).dense()[0] for r in rows]).cpu().numpy().astype(np.float64)
        assert np.abs(got[:, cols] - Kq[:, cols]).max() < 5e-4, np.abs(got[:, cols] - Kq[:, cols]).max()
        assert np.abs(got - Kq).max() < 5e-4
        # the fused right-hand side = one pass over the block the build stored; the pass: halves add up, repeatable
        b0 = be.ktk(K, w=w)
        assert float((ktw - b0).abs().max()) <= 1e-9 * max(1.0, float(b0.abs().max()))
        v = torch.randn(M, dtype=torch.float64, device="cuda", generator=g)
        full = be.ktk(K, v=v)
        assert torch.equal(full, be.ktk(K, v=v))
        parts = be.ktk(K.rows(0, n // 2 - 3), v=v) + be.ktk(K.rows(n // 2 - 3, n), v=v)
        assert float((parts - full).abs().max()) <= 1e-11 * float(full.abs().max())
        # fused scoring on the fp8 core, sampled rows against the restatement
        al = torch.randn(M, dtype=torch.float64, device="cuda", generator=g) * 0.05
        sc = be.mmv(F, Zf, sigma, al)
        want = Kq @ al.cpu().numpy()
        assert np.abs(sc[torch.from_numpy(rows).cuda(), 0].cpu().numpy() - want).max() < 5e-4 * float(al.abs().sum())
        del K
    finally:
        be.gauss, be.knm_storage = old, oldk
        be.release_workspaces()
        torch.cuda.empty_cache()


def test_config3_rls_thirty_regressors_n3e5(be):
    """Config 3's RLS half: COXY with n = 3e5 rows, D = 1024, 30 classes, lambda = 1000 through the drop-in
    RegionRefinerTrainer on the GPU; three classes (first, middle, last) against oracle/rls_ref.py, which is pinned to the
    reference's own RegionRefinerTrainer (tests/test_oracle_rls.py), every class through its normal equations."""
    from contextlib import redirect_stdout
    import io
    from odx.rls import RegionRefinerTrainer
    from oracle import rls_ref
    n, D, C, lam = 300_000, 1024, 30, 1000.0
    g = torch.Generator(device="cuda").manual_seed(1234 + 3)
    X = torch.randn((n, D), generator=g, device="cuda") * 0.6 + 0.15
    cls = (torch.arange(n, device="cuda") % C) + 1
    Wtrue = torch.randn((C, D, 4), generator=g, device="cuda") * 0.02
    Y = torch.empty((n, 4), device="cuda")
    for c in range(C):
        I = torch.where(cls == c + 1)[0]
        Y[I] = X[I] @ Wtrue[c] + 0.1 * torch.randn((len(I), 4), generator=g, device="cuda")
    cfg = {"CHOSEN_CLASSES": {i: ("c%d" % i if i else "_background_") for i in range(C + 1)},
           "REGION_REFINER": {"opts": {"lambda": lam}}}
    with redirect_stdout(io.StringIO()):
        models = RegionRefinerTrainer(cfg, lam, False)({"C": cls.float().view(-1, 1), "O": None, "X": X, "Y": Y})
    assert len(models) == C
    Xh, Yh, ch = X.cpu().numpy(), Y.cpu().numpy(), cls.cpu().numpy()
    for c in range(C):
        m = models[c]
        W = torch.stack([m["Beta"][str(k)]["weights"] for k in range(4)]).double()        # (4, D + 1)
        assert W.is_cuda and torch.isfinite(W).all()
        if c in (0, 14, 29):
            I = np.where(ch == c + 1)[0]
            ref = rls_ref.train_class(Xh[I], Yh[I], lam)
            scale = max(1.0, float(np.abs(ref["W"]).max()))
            assert np.abs(W.cpu().numpy() - ref["W"]).max() < 2e-6 * scale, c                # f32 hand-out of f64 weights
            for key in ("mu", "T", "T_inv"):
                assert np.allclose(m[key].cpu().numpy(), ref[key], atol=2e-6), (c, key)
            L = np.stack([m["Beta"][str(k)]["losses"].cpu().numpy() for k in range(4)])
            assert np.abs(L - ref["losses"]).max() < 1e-5 * max(1.0, float(ref["losses"].max())), c
