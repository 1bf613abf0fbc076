"""The FALKON oracle (oracle/falkon_ref.py) against independent mathematics: the
reference has no golden vectors at the falkon boundary ("parity unpinned"), so the restatement is
cross-checked with a dense f64 solve of the same normal equations, scipy's cdist, and the
preconditioner's defining identities."""
import numpy as np
import pytest
from scipy.spatial.distance import cdist

from oracle import falkon_ref as fr
from tests.synth import blob_problem, centres


def test_gaussian_kernel_vs_cdist():
    X, _, _ = blob_problem(300, 64, seed=1)
    Z = X[:50]
    ref = np.exp(-cdist(X.astype(np.float64), Z.astype(np.float64), "sqeuclidean") / (2 * 7.0 ** 2))
    assert np.abs(fr.gaussian_kernel(X.astype(np.float64), Z.astype(np.float64), 7.0) - ref).max() < 1e-12
    assert np.abs(fr.gaussian_kernel(X, Z, 7.0, np.float32) - ref).max() < 5e-6
    assert np.allclose(fr.kernel_mmv(X.astype(np.float64), Z.astype(np.float64), np.ones(50), 7.0, block=64)[:, 0], ref.sum(1))


def test_preconditioner_identities():
    X, _, _ = blob_problem(400, 32, seed=2)
    Z = X[:120].astype(np.float64)
    lam, eps, sigma = 1e-4, 1e-5, 6.0
    P = fr.Preconditioner(Z, sigma, lam, eps, np.float64)
    Kmm = fr.gaussian_kernel(Z, Z, sigma)
    assert np.abs(P.T.T @ P.T - (Kmm + eps * 120 * np.eye(120))).max() < 1e-12
    assert np.abs(P.A.T @ P.A - (P.T @ P.T.T / 120 + lam * np.eye(120))).max() < 1e-12
    v = np.random.default_rng(0).standard_normal((120, 1))
    assert np.allclose(P.T @ P.A @ P.apply(v) if False else P.A @ (P.T @ P.apply(v)), v)
    assert np.allclose(P.T.T @ (P.A.T @ P.apply_t(v)), v)


@pytest.mark.parametrize("sigma,lam", [(5.0, 1e-3), (5.0, 1e-4), (10.0, 1e-5)])
def test_config1_converges_to_dense_solution(sigma, lam):
    """BASELINE config 1: 1-class fit on 5k x 256 random RoI features, M = 500."""
    X, y, rng = blob_problem(5000, 256, seed=1234 + 1)
    idx = centres(y, 500, rng)
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    trace = []
    alpha, Z = fr.falkon_fit(X64, y64, idx, sigma, lam, maxiter=60, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-15,
                             trace=trace)
    dense = fr.dense_nystrom_krr(X64, y64, Z, sigma, lam, jitter=1e-5 * 500)
    # same fixed point: compare the fitted function, which is well conditioned
    f_it = fr.falkon_predict(X64, Z, alpha, sigma)
    f_dn = fr.falkon_predict(X64, Z, dense, sigma)
    assert np.abs(f_it - f_dn).max() < 1e-5  # 60 CG iterations; the (10, 1e-5) case is the slowest
    res = [np.sqrt(r.max()) for _, r in trace]
    assert res[-1] < 1e-3 * res[0]


def test_f32_regime_matches_f64_on_a_well_conditioned_problem():
    X, y, rng = blob_problem(2000, 64, seed=9)
    idx = centres(y, 200, rng)
    a64, _ = fr.falkon_fit(X.astype(np.float64), y.astype(np.float64), idx, 5.0, 1e-3, dtype=np.float64, pc_eps=1e-5,
                           cg_epsilon=1e-7)
    a32, _ = fr.falkon_fit(X, y, idx, 5.0, 1e-3, dtype=np.float32)
    assert np.linalg.norm(a32 - a64) / np.linalg.norm(a64) < 1e-3


def test_stored_and_recomputed_kernel_agree_and_shapes():
    X, y, rng = blob_problem(700, 32, seed=4)
    idx = centres(y, 64, rng)
    a1, Z = fr.falkon_fit(X.astype(np.float64), y, idx, 6.0, 1e-4, dtype=np.float64, pc_eps=1e-5)
    a2, _ = fr.falkon_fit(X.astype(np.float64), y, idx, 6.0, 1e-4, dtype=np.float64, pc_eps=1e-5, store_knm=False, row_block=128)
    assert a1.shape == (64, 1) and Z.shape == (64, 32)
    assert np.abs(a1 - a2).max() < 1e-6 * np.abs(a1).max()  # summation order x conditioning
    assert fr.falkon_predict(X[:10].astype(np.float64), Z, a1, 6.0).shape == (10, 1)


def test_compute_indices_selection_rule():
    y = np.array([1] * 30 + [-1] * 70, dtype=np.float32)
    draws = []

    def randint(high, size):
        draws.append((high, size))
        return np.arange(size) % high

    idx = fr.compute_indices_selection(y, 40, randint)
    assert draws == [(30, 20), (70, 20)]
    assert len(idx) == 40 and all(i < 30 for i in idx[:20]) and all(i >= 30 for i in idx[20:])
    small = fr.compute_indices_selection(np.array([1] * 5 + [-1] * 9), 40, randint)
    assert small == list(range(14))


def test_compute_indices_selection_matches_reference_fixture():
    """The reference's own compute_indices_selection, run with torch.manual_seed(77) in the build
    container (tests/golden/make_golden.py), reproduced by injecting the same torch draws."""
    import json
    import os
    import torch
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wrapper_contract.json")))
    y = np.array([1] * 30 + [-1] * 70, dtype=np.float32)
    torch.manual_seed(77)
    idx = fr.compute_indices_selection(y, 40, lambda high, size: torch.randint(high, (size,)).numpy())
    assert idx == c["incore"]["indices_seed77"] == c["cpu"]["indices_seed77"]


def test_scores_parallel_missing_classifier_conventions():
    X, _, rng = blob_problem(50, 16, seed=6)
    Z, _, _ = blob_problem(20, 16, seed=7)
    models = [(Z.astype(np.float64), rng.standard_normal(20)), None]
    det = fr.scores_parallel(X.astype(np.float64), models, 5.0, missing_fill=0.0, background=True)
    assert det.shape == (50, 3) and np.all(det[:, 0] == -2) and np.all(det[:, 2] == 0)
    rpn = fr.scores_parallel(X.astype(np.float64), models, 5.0, missing_fill=-2.0, background=False)
    assert rpn.shape == (50, 2) and np.all(rpn[:, 1] == -2)
    single = fr.falkon_predict(X.astype(np.float64), models[0][0], models[0][1][:, None], 5.0)
    assert np.allclose(det[:, 1], single[:, 0])
