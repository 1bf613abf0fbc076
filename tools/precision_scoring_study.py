"""CPU study for BASELINE config 5's reduced-precision contraction, SCORING side only (the fit keeps f32-accurate K_nM):
how a single-term bf16 or OCP-e4m3 X Z' inside the Gaussian kernel moves the scores K(X, Z) alpha of an f64-fitted model,
and how many of the Minibootstrap's negative-mining decisions it flips (hard negatives: score > -0.7, easy negatives:
score < -0.9, OnlineRegionClassifier_incore.py:108-140) at the statistics of this path (rows normalised to norm 20,
sigma = 15, lambda = 1e-5).  Variants of the inner product x.z:
  f16x3   two-term f16 split, three products, f32 accumulation  (what ships: odx_gauss_mmv_h2)
  f16x1   the "hi" term of that split alone                     (one f16 MFMA per product, same packed operands)
  bf16x1  operands rounded once to bf16, f32 accumulation       (one bf16 MFMA per product)
  e4m3x1  operands scaled by a power of two per matrix and rounded once to OCP fp8 e4m3, f32 accumulation (one fp8 MFMA)
  e4m3x2  two-term e4m3 split, three products
Usage: python tools/precision_scoring_study.py [n M D]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from tests.synth import blob_problem, centres  # noqa: E402
from tools.precision_study import fit, knm, split, to_bf16  # noqa: E402


def to_e4m3(a):
    """Round to nearest-even OCP fp8 e4m3 (3 mantissa bits, exponents -6 .. 8, max 448, subnormals)."""
    a = np.asarray(a, dtype=np.float32)
    mag = np.minimum(np.abs(a), np.float32(448.0))
    e = np.floor(np.log2(np.maximum(mag, np.float32(2.0 ** -9))))
    e = np.maximum(e, -6.0)
    step = np.exp2(e - 3.0).astype(np.float32)
    return (np.sign(a) * np.round(mag / step) * step).astype(np.float32)


def xz_low(X, Z, mode):
    if mode == "bf16x1":
        return (to_bf16(X) @ to_bf16(Z).T).astype(np.float64)
    if mode == "f16x1":
        s = np.float32(2.0 ** np.floor(np.log2(16384.0 / max(np.abs(X).max(), np.abs(Z).max()))))
        h = lambda a: (a * s).astype(np.float16).astype(np.float32)    # noqa: E731
        return (h(X) @ h(Z).T).astype(np.float64) / float(s) ** 2
    if mode.startswith("e4m3"):
        s = np.float32(2.0 ** np.floor(np.log2(224.0 / max(np.abs(X).max(), np.abs(Z).max()))))
        terms = 2 if mode.endswith("x2") else 1
        xs, zs = split(X * s, to_e4m3, terms), split(Z * s, to_e4m3, terms)
        acc = np.zeros((X.shape[0], Z.shape[0]), dtype=np.float32)
        for i, j in ([(0, 0)] if terms == 1 else [(1, 0), (0, 1), (0, 0)]):
            acc += xs[i] @ zs[j].T
        return acc.astype(np.float64) / float(s) ** 2
    raise ValueError(mode)


def kernel(X, Z, sigma, mode):
    if mode in ("f64", "f16x3"):
        return knm(X, Z, sigma, mode)
    sq1 = np.sum(X * X, axis=1, dtype=np.float32).astype(np.float64)[:, None]
    sq2 = np.sum(Z * Z, axis=1, dtype=np.float32).astype(np.float64)[None, :]
    return np.exp(np.maximum(sq1 + sq2 - 2.0 * xz_low(X, Z, mode), 0) * (-0.5 / sigma ** 2))


if __name__ == "__main__":
    n, M, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (12000, 1500, 1024)
    modes = ["f16x3", "f16x1", "bf16x1", "e4m3x2", "e4m3x1"]
    print("%-22s %-8s %12s %12s %10s %10s %10s" % ("problem", "x.z", "max|ds|", "mean|ds|", "flip>-0.7", "flip<-0.9", "|alpha|max"))
    for seed, sigma, lam in [(1, 15.0, 1e-5), (2, 15.0, 1e-4), (3, 10.0, 1e-5), (4, 20.0, 1e-5)]:
        X, y, rng = blob_problem(n, D, seed)
        Z = X[np.asarray(centres(y, M, rng))]
        a0 = fit(X, y, Z, sigma, lam, knm(X, Z, sigma, "f64"))
        # fresh rows of the same distribution (negatives of the mining loop): second half of a second draw
        Xt, yt, _ = blob_problem(n, D, seed + 100)
        Xt = Xt[yt < 0][: n // 2]
        s0 = kernel(Xt, Z, sigma, "f64") @ a0
        for m in modes:
            s = kernel(Xt, Z, sigma, m) @ a0
            d = np.abs(s - s0)
            print("%-22s %-8s %12.3e %12.3e %9.3f%% %9.3f%% %10.3g" % (
                "s%d sig=%g lam=%g" % (seed, sigma, lam), m, d.max(), d.mean(),
                100.0 * np.mean((s > -0.7) != (s0 > -0.7)), 100.0 * np.mean((s < -0.9) != (s0 < -0.9)), np.abs(a0).max()))
