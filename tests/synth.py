"""Seeded synthetic RoI-feature problems shaped like the reference's: a Gaussian
blob per class, normalised with the reference rule (x - mean) * 20 / mean_norm
(OnlineRegionClassifier.py:224-227), labels +-1 with 10 % positives."""
import numpy as np


def blob_problem(n, D, seed, pos_frac=0.1, noise=0.7):
    rng = np.random.default_rng(seed)
    npos = max(1, int(round(n * pos_frac)))
    mu = rng.standard_normal(D)
    others = rng.standard_normal((4, D))
    Xp = mu + noise * rng.standard_normal((npos, D))
    Xn = noise * rng.standard_normal((n - npos, D)) + others[rng.integers(0, 4, n - npos)] * 0.6
    X = np.concatenate([Xp, Xn])
    y = np.concatenate([np.ones(npos), -np.ones(n - npos)])
    X = X - X.mean(0)
    X *= 20.0 / np.linalg.norm(X, axis=1).mean()
    perm = rng.permutation(n)
    return np.ascontiguousarray(X[perm].astype(np.float32)), y[perm].astype(np.float32), rng


def centres(y, M, rng):
    from oracle import falkon_ref as fr
    return fr.compute_indices_selection(y, M, lambda high, size: rng.integers(0, high, size))
