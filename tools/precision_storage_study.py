"""CPU study: how the STORAGE format of K_nM (the 40 GB per class the CG passes stream) moves the fitted alpha.

The entries are formed at f32 accuracy (two-term f16 split, tools/precision_study.py `f16x3`) and then stored as
  f32        what ships
  u24        fixed point on [0, 1]: round(K * 2^24) in 24 bits (a u16 `hi` plane + a u8 `lo` plane), step 2^-24 —
             the absolute error of f32 on [0.5, 1), 2 x / 4 x / ... that of f32 on [0.25, 0.5) / [0.125, 0.25) / ...
  u24s       the same with the step halved by a sqrt companding: stores round(sqrt(K) * 2^24); K = q^2 (relative
             error 2^-24 / sqrt(K): uniform in sqrt(K) rather than in K)
  fl22       f32 rounded to 22 significant bits (round 2's rejected truncated float, for comparison)
  bf16       K rounded to bf16 (BASELINE config 2's throughput storage)
Everything else is evaluated in f64 with the f32-regime constants: the column is the relative distance of
alpha from the all-f64 evaluation — the quantity the 1e-4 bar of north_star is stated on.
Problems: the six (sigma, lambda) problems of tools/precision_study.py and the five shapes of
tests/test_gpu_kernels.py::test_falkon_fit_alpha_parity.
Usage: python tools/precision_storage_study.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from precision_study import fit, knm  # noqa: E402
from tests.synth import blob_problem, centres  # noqa: E402


def store(K32, fmt):
    """K32: f64 array holding f32 values in [0, 1]."""
    if fmt == "f32":
        return K32
    if fmt == "u24":
        q = np.minimum(np.rint(K32 * 16777216.0), 16777215.0)
        return q / 16777216.0
    if fmt == "u24s":
        q = np.minimum(np.rint(np.sqrt(K32) * 16777216.0), 16777215.0) / 16777216.0
        return q * q
    if fmt == "fl22":
        u = K32.astype(np.float32).view(np.uint32)
        u = (u + np.uint32(1)) & np.uint32(0xFFFFFFFC)        # round to 22 significant bits (2 dropped)
        return u.view(np.float32).astype(np.float64)
    if fmt == "bf16":
        u = K32.astype(np.float32).view(np.uint32)
        r = ((u >> 16) & 1) + 0x7FFF
        return ((u + r) & 0xFFFF0000).view(np.float32).astype(np.float64)
    raise ValueError(fmt)


if __name__ == "__main__":
    fmts = ["f32", "u24", "u24s", "fl22", "bf16"]
    probs = [("study s%d" % s, 20000, 1000, 256, s, sg, lm) for s, sg, lm in
             [(1, 10.0, 1e-5), (2, 15.0, 1e-5), (3, 15.0, 1e-6), (4, 25.0, 1e-6), (5, 5.0, 1e-4), (6, 15.0, 1e-7)]]
    probs += [("parity", n, M, D, n + M, sg, lm) for n, M, D, sg, lm in
              [(5000, 500, 256, 10.0, 1e-5), (5000, 500, 256, 15.0, 1e-5), (3000, 300, 1024, 15.0, 1e-5),
               (2500, 1000, 2048, 5.0, 1e-4), (777, 129, 36, 5.0, 1e-3)]]
    print("%-44s" % "problem", *("%9s" % m for m in fmts), "  (alpha rel err vs the f64 evaluation)")
    for name, n, M, D, seed, sigma, lam in probs:
        X, y, rng = blob_problem(n, D, seed)
        Z = X[np.asarray(centres(y, M, rng))]
        a0 = fit(X, y, Z, sigma, lam, knm(X, Z, sigma, "f64"))
        K32 = knm(X, Z, sigma, "f16x3")
        row = []
        for f in fmts:
            a = fit(X, y, Z, sigma, lam, store(K32, f))
            row.append("%.1e" % (np.linalg.norm(a - a0) / np.linalg.norm(a0)))
        print("%-44s" % ("%s n=%d M=%d D=%d sigma=%g lam=%g" % (name, n, M, D, sigma, lam)), *("%9s" % r for r in row),
              flush=True)
