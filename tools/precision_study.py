"""CPU study: how the precision of the X Z' product inside K_nM moves the fitted alpha.

Everything except the K_nM inner product is evaluated in f64 with the f32-regime constants (the parity
target).  Variants of the inner product:
  f64      exact
  f32      f32 products, f32 accumulation (what v_mfma_f32_32x32x2_f32 does)
  f16x3    x = hi + lo (two f16), hi*hi + hi*lo + lo*hi with f32 accumulation (three f16 MFMAs)
  bf16x3 / bf16x6   the same idea with two / three bf16 terms
Usage: python tools/precision_study.py [n M D]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from oracle import falkon_ref as fr  # noqa: E402
from tests.synth import blob_problem, centres  # noqa: E402


def to_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split(a, conv, terms):
    parts, rem = [], a.astype(np.float32)
    for _ in range(terms):
        p = conv(rem)
        parts.append(p)
        rem = (rem - p).astype(np.float32)
    return parts


def xz(X, Z, mode):
    if mode == "f64":
        return X.astype(np.float64) @ Z.astype(np.float64).T
    if mode == "f32":
        return (X @ Z.T).astype(np.float64)
    conv = (lambda a: a.astype(np.float16).astype(np.float32)) if mode.startswith("f16") else to_bf16
    terms = 3 if mode.endswith("x6") else 2
    scale = np.float32(2.0 ** np.floor(np.log2(16384.0 / max(np.abs(X).max(), np.abs(Z).max())))) if mode.startswith("f16") else np.float32(1)
    xs, zs = split(X * scale, conv, terms), split(Z * scale, conv, terms)
    acc = np.zeros((X.shape[0], Z.shape[0]), dtype=np.float32)
    pairs = [(0, 0), (0, 1), (1, 0)] if terms == 2 else [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    for i, j in pairs[::-1]:
        acc += xs[i] @ zs[j].T
    return acc.astype(np.float64) / float(scale) ** 2


def knm(X, Z, sigma, mode):
    sq1 = np.sum(X * X, axis=1, dtype=np.float32).astype(np.float64)[:, None]
    sq2 = np.sum(Z * Z, axis=1, dtype=np.float32).astype(np.float64)[None, :]
    if mode == "f64":
        sq1 = np.sum(X.astype(np.float64) ** 2, axis=1)[:, None]
        sq2 = np.sum(Z.astype(np.float64) ** 2, axis=1)[None, :]
    d2 = np.maximum(sq1 + sq2 - 2.0 * xz(X, Z, mode), 0)
    K = np.exp(d2 * (-0.5 / sigma ** 2))
    return K if mode == "f64" else K.astype(np.float32).astype(np.float64)


def fit(X, y, Z, sigma, lam, K):
    n = X.shape[0]
    prec = fr.Preconditioner(Z.astype(np.float64), sigma, lam, fr.PC_EPSILON[np.dtype(np.float32)], np.float64)
    Y = y.astype(np.float64)[:, None]
    B = prec.apply_t(K.T @ (Y / n))

    def mmv(sol):
        v = prec.invA(sol)
        cc = K.T @ (K @ prec.invT(v)) / n
        return prec.invAt(prec.invTt(cc) + lam * v)
    beta = fr.conjugate_gradient(B, mmv, 20, np.float64, fr.CG_EPSILON[np.dtype(np.float32)])
    return prec.apply(beta)


if __name__ == "__main__":
    n, M, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (20000, 1000, 256)
    modes = ["f32", "f16x3", "bf16x3", "bf16x6"]
    print("%-28s" % "problem", *("%10s" % m for m in modes), " (alpha rel err vs f64 | max K abs err)")
    for seed, sigma, lam in [(1, 10.0, 1e-5), (2, 15.0, 1e-5), (3, 15.0, 1e-6), (4, 25.0, 1e-6), (5, 5.0, 1e-4), (6, 15.0, 1e-7)]:
        X, y, rng = blob_problem(n, D, seed)
        Z = X[np.asarray(centres(y, M, rng))]
        K0 = knm(X, Z, sigma, "f64")
        a0 = fit(X, y, Z, sigma, lam, K0)
        row = []
        for m in modes:
            K = knm(X, Z, sigma, m)
            a = fit(X, y, Z, sigma, lam, K)
            row.append("%.1e|%.0e" % (np.linalg.norm(a - a0) / np.linalg.norm(a0), np.abs(K - K0).max()))
        print("%-28s" % ("s%d sigma=%g lam=%g" % (seed, sigma, lam)), *("%10s" % r for r in row))
