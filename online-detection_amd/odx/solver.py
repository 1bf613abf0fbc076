"""FALKON fit: host-side driver of the preconditioned conjugate gradient (A4).

Restates what ``InCoreFalkon(...).fit(X, y)`` does at the reference's call site
(src/modules/region-classifier/FALKONWrapper_with_centers_selection_incore.py:56-68) with
falkon's upstream defaults for the f32 regime the reference runs in:
    T = chol(K_MM + eps M I)', A = chol(T T'/M + lam I)'          (preconditioner, f64 here)
    b = A^-T T^-T K_nM' (y / n)
    CG, maxiter steps, on  beta -> A^-T [ T^-T K_nM' (K_nM T^-1 A^-1 beta) / n + lam A^-1 beta ]
    alpha = T^-1 A^-1 beta
Every array op is a libodx kernel reached through the backend; this file only sequences
them and places the (optional) row-shard all-reduce.  No host synchronisation inside the loop:
step sizes, residual norms and the stop flag stay on the device.
"""
from dataclasses import dataclass


@dataclass
class SolverOptions:
    """falkon's numeric knobs (upstream FalkonOptions defaults for float32 data)."""
    pc_epsilon: float = 1e-5            # jitter eps: K_MM + eps*M*I
    cg_epsilon: float = 1e-7            # added to both CG denominators
    cg_tolerance: float = 1e-7          # stop when sqrt(||r||^2) < cg_tolerance^2
    cg_full_gradient_every: int = 10    # recompute r = b - A x from scratch every k iterations
    check_pivots: bool = True           # one host sync after the fit to report a failed Cholesky


def falkon_fit(be, F, y, Zf, sigma, lam, maxiter=20, opt=None, n_total=None, allreduce=None, knm_out=None,
               return_knm=False):
    """Fit one binary FALKON problem.

    be        backend (odx.backend.HipBackend in the product)
    F         Features of this rank's rows (n_local x D)
    y         f64 device vector of labels for those rows (n_local,)
    Zf        Features of the M Nystroem centres (identical on every rank)
    n_total   number of rows over all shards (default: F.n)
    allreduce callable summing an f64 device vector in place over the row shards
              (odx.dist.RowShard.allreduce); None for a single shard
    returns   alpha (M,) f64 device vector
    """
    opt = opt or SolverOptions()
    n = float(F.n if n_total is None else n_total)
    M = Zf.n
    ar = allreduce if allreduce is not None else (lambda v: v)

    P = be.precond(Zf, sigma, lam, opt.pc_epsilon)
    K = be.knm(F, Zf, sigma, out=knm_out)

    def mmv(s, out):
        v = be.trmv(P, "LAit", s)                      # A^-1 s
        t = be.trmv(P, "LTit", v)                      # T^-1 A^-1 s
        cc = ar(be.ktk(K, v=t))                        # K' K t, summed over shards
        u = be.trmv(P, "LTi", cc, alpha=1.0 / n, beta=lam, z=v)   # T^-T cc / n + lam v
        return be.trmv(P, "LAi", u, out=out)           # A^-T u

    yn = y * (1.0 / n)
    b0 = ar(be.ktk(K, w=yn))                           # K' (y / n)
    B = be.trmv(P, "LAi", be.trmv(P, "LTi", b0))       # A^-T T^-T b0

    X, R, Pv, AP = be.zeros(M), be.zeros(M), be.zeros(M), be.zeros(M)
    state = be.zeros(4)
    be.cg_init(B, X, R, Pv, state)
    tol = opt.cg_tolerance ** 2
    for it in range(maxiter):
        mmv(Pv, AP)
        full = (it + 1) % opt.cg_full_gradient_every == 0
        be.cg_step(X, R, Pv, AP, state, opt.cg_epsilon, full)
        if full:
            mmv(X, AP)
            R.copy_(B)
            be.axpby(-1.0, AP, 1.0, R)                 # R = B - mmv(X)
        be.cg_finish(R, Pv, state, opt.cg_epsilon, tol)
    alpha = be.trmv(P, "LTit", be.trmv(P, "LAit", X))  # T^-1 A^-1 beta
    if opt.check_pivots:
        be.check_precond(P)
    if return_knm:
        return alpha, K
    return alpha
