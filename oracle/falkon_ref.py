"""CPU restatement (numpy) of the FALKON fit / predict path — TEST ORACLE ONLY.

PARITY UNPINNED BY THE REFERENCE: the arithmetic of this path lives in the
third-party package ``falkon`` (FalkonML/falkon pinned at git
0d96c685dbdff7048e7410e5ca419b21e337789d, INSTALLATION_GUIDE.md:71-75) which is not
vendored under /root/reference and cannot be installed offline; the reference has no
tests or golden vectors at that boundary.  What this file restates, and from where:

* the call contract and constants of the reference's own call sites
  (src/modules/region-classifier/FALKONWrapper_with_centers_selection_incore.py:43-73:
  GaussianKernel(sigma), penalty=lam, M=len(indices), maxiter=20,
  center_selection=MyCenterSelector(indices), FalkonOptions(... store_kernel_d_threshold=250)
  => K_nM is materialised once and CG runs on the stored matrix);
* the published algorithm (Rudi, Carratino, Rosasco: "FALKON: An Optimal Large Scale
  Kernel Method", NeurIPS 2017, Alg. 1) in the form the pinned falkon release implements
  it: preconditioner  T = chol(K_MM + eps*M*I) (upper, T'T = .),
  A = chol(T T'/M + lam*I) (upper); right-hand side  b = A^-T T^-T K_nM' (y/n);
  CG on  beta -> A^-T [ T^-T K_nM'(K_nM T^-1 A^-1 beta)/n + lam * A^-1 beta ];
  alpha = T^-1 A^-1 beta;  with falkon's CG details (per-column step sizes,
  ``cg_epsilon`` added to both denominators, residual recomputed from scratch every
  ``cg_full_gradient_every`` iterations, stop when sqrt(max_col ||r||^2) < cg_tolerance^2).
  The numeric defaults (pc_epsilon_32=1e-5, pc_epsilon_64=1e-13, cg_epsilon_32=1e-7,
  cg_epsilon_64=1e-15, cg_tolerance=1e-7, cg_full_gradient_every=10) are falkon's
  upstream defaults recalled from that release; they are parameters here.

Independent check: ``dense_nystrom_krr`` solves the same normal equations directly in
f64; ``falkon_fit`` converges to it (tests/test_oracle_falkon.py).
"""
import numpy as np
import scipy.linalg as sla

PC_EPSILON = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-13}
CG_EPSILON = {np.dtype(np.float32): 1e-7, np.dtype(np.float64): 1e-15}
CG_TOLERANCE = 1e-7
CG_FULL_GRADIENT_EVERY = 10


def gaussian_kernel(X1, X2, sigma, dtype=None):
    """K_ij = exp(-||x_i - z_j||^2 / (2 sigma^2)) the way falkon's GaussianKernel forms it
    (kernel ctor: FALKONWrapper_with_centers_selection_incore.py:50): X1 X2' by GEMM,
    squared-norm broadcast, clamp at 0, scale by gamma = -1/(2 sigma^2), exp."""
    dtype = np.dtype(dtype or X1.dtype)
    X1 = np.asarray(X1, dtype=dtype)
    X2 = np.asarray(X2, dtype=dtype)
    gamma = dtype.type(-0.5 / (float(sigma) ** 2))
    sq1 = np.sum(X1 * X1, axis=1, dtype=dtype)[:, None]
    sq2 = np.sum(X2 * X2, axis=1, dtype=dtype)[None, :]
    D2 = X1 @ X2.T
    D2 *= dtype.type(-2.0)
    D2 += sq1
    D2 += sq2
    np.maximum(D2, 0, out=D2)
    D2 *= gamma
    np.exp(D2, out=D2)
    return D2


def kernel_mmv(X1, X2, V, sigma, dtype=None, block=8192):
    """kernel.mmv(X1, X2, V) = K(X1, X2) @ V, row-blocked so K is never held whole
    (call sites: roi_box_predictors.py:158, rpn.py:225, roi_mask_predictors.py:90)."""
    dtype = np.dtype(dtype or X1.dtype)
    V = np.asarray(V, dtype=dtype)
    if V.ndim == 1:
        V = V[:, None]
    out = np.empty((X1.shape[0], V.shape[1]), dtype=dtype)
    for s in range(0, X1.shape[0], block):
        out[s:s + block] = gaussian_kernel(X1[s:s + block], X2, sigma, dtype) @ V
    return out


class Preconditioner:
    """T, A factors of the FALKON preconditioner (both upper triangular)."""

    def __init__(self, Z, sigma, lam, eps=None, dtype=None):
        dtype = np.dtype(dtype or Z.dtype)
        if eps is None:
            eps = PC_EPSILON[dtype]
        M = Z.shape[0]
        C = gaussian_kernel(Z, Z, sigma, dtype)
        C[np.diag_indices(M)] += dtype.type(eps * M)
        # T'T = K_MM + eps*M*I
        self.T = np.ascontiguousarray(sla.cholesky(C, lower=False, check_finite=False).astype(dtype))
        AA = (self.T @ self.T.T) / dtype.type(M)
        AA[np.diag_indices(M)] += dtype.type(lam)
        # A'A = T T'/M + lam*I
        self.A = np.ascontiguousarray(sla.cholesky(AA, lower=False, check_finite=False).astype(dtype))
        self.dtype = dtype

    def _solve(self, U, v, trans):
        return sla.solve_triangular(U, v, lower=False, trans=trans, check_finite=False).astype(self.dtype)

    def invT(self, v): return self._solve(self.T, v, 0)
    def invTt(self, v): return self._solve(self.T, v, 1)
    def invA(self, v): return self._solve(self.A, v, 0)
    def invAt(self, v): return self._solve(self.A, v, 1)
    def apply(self, v): return self.invT(self.invA(v))          # T^-1 A^-1 v
    def apply_t(self, v): return self.invAt(self.invTt(v))       # A^-T T^-T v


def conjugate_gradient(B, mmv, max_iter, dtype, cg_epsilon=None, cg_tolerance=CG_TOLERANCE,
                       full_gradient_every=CG_FULL_GRADIENT_EVERY, trace=None):
    """falkon's batched CG (one step size per right-hand-side column)."""
    dtype = np.dtype(dtype)
    m_eps = dtype.type(CG_EPSILON[dtype] if cg_epsilon is None else cg_epsilon)
    tol = cg_tolerance ** 2
    R = B.copy()
    X = np.zeros_like(B)
    P = R.copy()
    Rsold = np.sum(R * R, axis=0, dtype=dtype)
    for it in range(max_iter):
        AP = mmv(P)
        alpha = Rsold / (np.sum(P * AP, axis=0, dtype=dtype) + m_eps)
        X = X + P * alpha[None, :]
        if (it + 1) % full_gradient_every == 0:
            R = B - mmv(X)
        else:
            R = R - AP * alpha[None, :]
        Rsnew = np.sum(R * R, axis=0, dtype=dtype)
        if trace is not None:
            trace.append((X.copy(), Rsnew.copy()))
        if np.sqrt(np.max(np.abs(Rsnew))) < tol:
            break
        P = R + P * (Rsnew / (Rsold + m_eps))[None, :]
        Rsold = Rsnew
    return X


def falkon_fit(X, y, center_idx, sigma, lam, maxiter=20, dtype=None, pc_eps=None,
               cg_epsilon=None, cg_tolerance=CG_TOLERANCE,
               full_gradient_every=CG_FULL_GRADIENT_EVERY, store_knm=True,
               row_block=8192, trace=None, allreduce=None, knm=None):
    """InCoreFalkon(kernel=GaussianKernel(sigma), penalty=lam, M=len(center_idx),
    maxiter=maxiter, center_selection=MyCenterSelector(center_idx)).fit(X, y)
    (FALKONWrapper_with_centers_selection_incore.py:58-68).

    X (n, D), y (n,) or (n, T).  Returns (alpha (M, T), ny_points (M, D)).
    ``allreduce`` (optional) sums an (M, T) array over row shards: with it, X/y are one
    shard and ``n_total`` rows is obtained by all-reducing the local count (used by the
    world_size>1 tests; the reference itself is single-process).
    ``knm`` (optional): the stored K_nM block to iterate on instead of gaussian_kernel(X, Z) — for storage formats that
    round the entries (bf16 / fp8 throughput variants): the oracle then states what the SOLVER must produce on that block.
    """
    dtype = np.dtype(dtype or X.dtype)
    X = np.asarray(X, dtype=dtype)
    Y = np.asarray(y, dtype=dtype)
    if Y.ndim == 1:
        Y = Y[:, None]
    n_local = X.shape[0]
    n = n_local if allreduce is None else int(allreduce(np.array([[float(n_local)]]))[0, 0])
    Z = np.ascontiguousarray(X[np.asarray(center_idx, dtype=np.int64)]) if center_idx is not None else None
    return falkon_fit_centers(X, Y, Z, sigma, lam, n, maxiter, dtype, pc_eps, cg_epsilon, cg_tolerance,
                              full_gradient_every, store_knm, row_block, trace, allreduce, knm)


def falkon_fit_centers(X, Y, Z, sigma, lam, n, maxiter=20, dtype=None, pc_eps=None, cg_epsilon=None,
                       cg_tolerance=CG_TOLERANCE, full_gradient_every=CG_FULL_GRADIENT_EVERY,
                       store_knm=True, row_block=8192, trace=None, allreduce=None, knm=None):
    dtype = np.dtype(dtype or X.dtype)
    X = np.asarray(X, dtype=dtype)
    Y = np.asarray(Y, dtype=dtype)
    if Y.ndim == 1:
        Y = Y[:, None]
    Z = np.asarray(Z, dtype=dtype)
    prec = Preconditioner(Z, sigma, lam, pc_eps, dtype)
    ar = (lambda a: a) if allreduce is None else allreduce
    nn = dtype.type(n)
    lam_t = dtype.type(lam)

    if knm is not None:
        Knm = np.asarray(knm, dtype=dtype)
        assert Knm.shape == (X.shape[0], Z.shape[0])
    else:
        Knm = gaussian_kernel(X, Z, sigma, dtype) if store_knm else None

    def ktk(v, w=None):
        """K_nM' (K_nM v + w) summed over row shards."""
        if Knm is not None:
            t = Knm @ v if v is not None else 0
            if w is not None:
                t = t + w
            return ar(Knm.T @ t)
        out = np.zeros((Z.shape[0], (v if v is not None else w).shape[1]), dtype=dtype)
        for s in range(0, X.shape[0], row_block):
            Kb = gaussian_kernel(X[s:s + row_block], Z, sigma, dtype)
            t = Kb @ v if v is not None else 0
            if w is not None:
                t = t + w[s:s + row_block]
            out += Kb.T @ t
        return ar(out)

    B = prec.apply_t(ktk(None, Y / nn))

    def mmv(sol):
        v = prec.invA(sol)
        cc = ktk(prec.invT(v)) / nn
        return prec.invAt(prec.invTt(cc) + lam_t * v)

    beta = conjugate_gradient(B, mmv, maxiter, dtype, cg_epsilon, cg_tolerance, full_gradient_every, trace)
    alpha = prec.apply(beta)
    return alpha.astype(dtype), Z


def falkon_predict(X, ny_points, alpha, sigma, dtype=None, block=8192):
    """model.predict(X) = K(X, ny_points_) @ alpha_ -> (n, T)
    (FALKONWrapper_with_centers_selection_incore.py:75-82)."""
    return kernel_mmv(X, ny_points, alpha, sigma, dtype, block)


def dense_nystrom_krr(X, y, Z, sigma, lam, jitter=0.0):
    """Independent oracle: direct f64 solve of the Nystroem-KRR normal equations FALKON's
    iteration converges to:  (K_nM' K_nM / n + lam (K_MM + jitter I)) alpha = K_nM' y / n."""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    Y = np.asarray(y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    n = X.shape[0]
    Knm = gaussian_kernel(X, Z, sigma, np.float64)
    Kmm = gaussian_kernel(Z, Z, sigma, np.float64)
    H = Knm.T @ Knm / n + lam * (Kmm + jitter * np.eye(Z.shape[0]))
    return np.linalg.lstsq(H, Knm.T @ Y / n, rcond=None)[0]


def compute_indices_selection(y, nyst_centers, randint):
    """Nystroem index rule of FALKONWrapper.compute_indices_selection
    (FALKONWrapper_with_centers_selection_incore.py:87-99): all positives if there are at
    most floor(M/2) of them else a with-replacement sample of floor(M/2); negatives fill up
    to M the same way; order = positives then negatives.  ``randint(high, size)`` supplies
    the random draws so tests can inject them."""
    y = np.asarray(y).reshape(-1)
    pos = np.nonzero(y == 1)[0]
    if pos.shape[0] > int(nyst_centers / 2):
        pos = pos[np.asarray(randint(pos.shape[0], int(nyst_centers / 2)), dtype=np.int64)]
    neg = np.nonzero(y == -1)[0]
    if neg.shape[0] > nyst_centers - pos.shape[0]:
        neg = neg[np.asarray(randint(neg.shape[0], nyst_centers - pos.shape[0]), dtype=np.int64)]
    return np.concatenate([pos, neg]).tolist()


def scores_parallel(F, models, sigma, dtype=None, missing_fill=0.0, background=True):
    """Batched multi-class scoring of the test-time heads: one mmv against the concatenated
    centres with a block-structured alpha (class i's alpha in the rows of its own centres,
    column i).  ``models`` = list of (ny_points, alpha) or None.

    * detector head, predict_clss_FALKON_parallel (roi_box_predictors.py:140-160): a -2
      background column is prepended (``background=True``); a missing classifier's column
      stays at K@0 = 0 because its alpha_parallel column is all zero (:143,153-155) —
      only the sequential path (:127-138) writes -2 there.  ``missing_fill=0``.
    * RPN head, compute_objectness_FALKON_parallel (rpn.py:201-227): no background column
      (``background=False``); ``matrix_to_subtract`` makes a missing classifier score
      0 - 2 = -2 (:216-218,226).  ``missing_fill=-2``.
    """
    dtype = np.dtype(dtype or F.dtype)
    C = len(models)
    scores = np.zeros((F.shape[0], C), dtype=dtype)
    live = [m for m in models if m is not None]
    if live:
        total = sum(m[0].shape[0] for m in live)
        alpha_par = np.zeros((total, C), dtype=dtype)
        row = 0
        for i, m in enumerate(models):
            if m is not None:
                alpha_par[row:row + m[0].shape[0], i] = np.asarray(m[1]).reshape(-1)
                row += m[0].shape[0]
        ny_par = np.concatenate([m[0] for m in live]).astype(dtype)
        scores = kernel_mmv(F, ny_par, alpha_par, sigma, dtype)
    for i, m in enumerate(models):
        if m is None:
            scores[:, i] = missing_fill
    if background:
        scores = np.concatenate([np.full((F.shape[0], 1), -2.0, dtype=dtype), scores], axis=1)
    return scores
