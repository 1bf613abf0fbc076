"""FALKON fit: host-side driver of the preconditioned conjugate gradient (A4).

Restates what ``InCoreFalkon(...).fit(X, y)`` does at the reference's call site
(src/modules/region-classifier/FALKONWrapper_with_centers_selection_incore.py:56-68) with
falkon's upstream defaults for the f32 regime the reference runs in:
    T = chol(K_MM + eps M I)', A = chol(T T'/M + lam I)'          (preconditioner, f64 here)
    b = A^-T T^-T K_nM' (y / n)                                   (K_nM' (y / n) comes out of the K_nM build)
    CG, maxiter steps, on  beta -> A^-T [ T^-T K_nM' (K_nM T^-1 A^-1 beta) / n + lam A^-1 beta ]
    alpha = T^-1 A^-1 beta
Every array op is a libodx kernel reached through the backend; this file only sequences
them and places the (optional) row-shard all-reduce.  No host synchronisation inside the loop:
step sizes, residual norms and the stop flag stay on the device.
"""
from dataclasses import dataclass

import torch


@dataclass
class SolverOptions:
    """falkon's numeric knobs (upstream FalkonOptions defaults for float32 data)."""
    pc_epsilon: float = 1e-5            # jitter eps: K_MM + eps*M*I
    cg_epsilon: float = 1e-7            # added to both CG denominators
    cg_tolerance: float = 1e-7          # stop when sqrt(||r||^2) < cg_tolerance^2
    cg_full_gradient_every: int = 10    # recompute r = b - A x from scratch every k iterations
    check_pivots: bool = True           # one host sync after the fit to report a failed Cholesky


_DEFERRED = None     # while a deferred_pivot_checks() block is open: the (backend, info) pairs still to be looked at


class deferred_pivot_checks:
    """Fits issued inside the block do not synchronise with the host to look at their Cholesky status; every status is
    checked when the block closes.  Lets the host enqueue several independent fits (on several streams) back to back."""

    def __enter__(self):
        global _DEFERRED
        self._prev, self.items = _DEFERRED, []
        _DEFERRED = self.items
        return self

    def __exit__(self, exc_type, exc, tb):
        global _DEFERRED
        _DEFERRED = self._prev
        if exc_type is None and self.items:
            # one host read for all status words of the block (they are 1-element device tensors)
            vals = torch.cat([info.reshape(-1)[:1] for _, info in self.items]).tolist()
            for (be, info), v in zip(self.items, vals):
                if v != 0:
                    be.check_info(info)          # raises with the backend's message
        return False


def _check_pivots(be, P):
    if _DEFERRED is not None and hasattr(be, "check_info") and hasattr(P, "info"):
        _DEFERRED.append((be, P.info))
    else:
        be.check_precond(P)


class _NoPhase:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _can_fold(be, K, M, rows_per_rank):
    """The two-vector pass exists for this block width and pays for itself: a pass reads 4 n M bytes, the second vector's
    four extra triangular products 16 M^2.  Decided from quantities every rank agrees on (M and the job's rows per rank),
    never from the local shard length, so that all ranks issue the same collectives."""
    return bool(hasattr(be, "ktk2") and be.can_ktk2(K) and rows_per_rank >= 8 * M)


def falkon_fit(be, F, y, Zf, sigma, lam, maxiter=20, opt=None, n_total=None, allreduce=None, knm_out=None,
               return_knm=False, phase=None, precond=None, shard=None, owner=None, precond_ready=None):
    """Fit one binary FALKON problem.

    be        backend (odx.backend.HipBackend in the product)
    F         Features of this rank's rows (n_local x D)
    y         f64 device vector of labels for those rows (n_local,)
    Zf        Features of the M Nystroem centres (identical on every rank)
    n_total   number of rows over all shards (default: F.n)
    allreduce callable summing an f64 device vector in place over the row shards
              (odx.dist.RowShard.allreduce); None for a single shard
    shard, owner
              owner-computes mode for row shards (odx.dist.RowShard + a rank id): only `owner`
              holds the preconditioner and the CG state.  Per CG step the owner broadcasts the
              (M,) direction T^-1 A^-1 p, every rank runs its K_nM pass, the partials are
              all-reduced, and the owner alone applies the preconditioner and updates the
              iterate; alpha is broadcast at the end.  Ranks other than the owner pass
              precond=None and never build one.  Without `shard` every rank does everything
              (replicated mode) and only `allreduce` is used.
    phase     optional callable name -> context manager bracketing the launches of one kernel
              ("precond", "knm", "ktk" = the one-vector pass, "ktk2" = the two-vector pass); bench.py hangs
              HIP-event timers on it
    precond   an already computed preconditioner for (Zf, sigma, lam) to reuse
    precond_ready
              optional callable invoked once, right before the preconditioner is first applied
              (after the K_nM build and the right-hand-side pass were issued): lets a preconditioner
              that is still being computed on another stream overlap with them
    returns   alpha (M,) f64 device vector (on every rank)
    """
    opt = opt or SolverOptions()
    n = float(F.n if n_total is None else n_total)
    M = Zf.n
    if shard is not None and allreduce is None:
        allreduce = shard.allreduce
    ar = allreduce if allreduce is not None else (lambda v: v)
    ph = phase if phase is not None else (lambda name: _NoPhase())
    owned = shard is None or owner is None or shard.rank == owner   # this rank runs the M-sized algebra
    bcast = (lambda v: shard.broadcast(v, src=owner)) if (shard is not None and owner is not None) else (lambda v: v)

    P = precond
    if owned and P is None:
        with ph("precond"):
            P = be.precond(Zf, sigma, lam, opt.pc_epsilon)
    yn = y * (1.0 / n)
    with ph("knm"):
        K, b0 = be.knm_rhs(F, Zf, sigma, yn, out=knm_out)   # K_nM and this shard's K' (y / n), out of the same launch

    def ktk(**kw):
        with ph("ktk"):
            r = be.ktk(K, **kw)
        return ar(r)

    t = be.zeros(M)
    v = be.zeros(M)

    def mmv(s, out):
        """out = A^-T [ T^-T K'K (T^-1 A^-1 s) / n + lam A^-1 s ]  (owner); every rank passes over K."""
        if owned:
            be.trmv(P, "LAit", s, out=v)               # A^-1 s
            be.trmv(P, "LTit", v, out=t)               # T^-1 A^-1 s
        bcast(t)
        cc = ktk(v=t)                                  # K' K t, summed over shards
        if owned:
            u = be.trmv(P, "LTi", cc, alpha=1.0 / n, beta=lam, z=v)   # T^-T cc / n + lam v
            be.trmv(P, "LAi", u, out=out)              # A^-T u

    # The periodic full residual R = B - W x (falkon: every cg_full_gradient_every-th step) without a pass of its own:
    # W is linear and x_new = x_old + a p, so W x_new = W x_old + a W p, and W x_old comes out of the SAME read of K_nM
    # as this step's W p (a two-vector pass, backend.ktk2).  Same value up to f64 rounding, and still computed from fresh
    # products, so it keeps removing the recursive drift the recomputation exists for.  Used where the pass dominates
    # the second vector's four extra triangular products (wide, tall blocks); otherwise the plain sequence below.
    if shard is not None:
        can_fold = _can_fold(be, K, M, int(n) // shard.world)
    else:
        can_fold = allreduce is None and _can_fold(be, K, M, int(n))
    tt2 = be.zeros(2 * ((M + 1) // 2 * 2)).view(2, -1) if can_fold else None
    v2 = be.zeros(M) if can_fold else None

    def mmv2(s1, out1, s2, out2):
        """out1 = W s1 and out2 = W s2 from one read of K."""
        t1, t2 = tt2[0, :M], tt2[1, :M]
        if owned:
            be.trmv(P, "LAit", s1, out=v)
            be.trmv(P, "LTit", v, out=t1)
            be.trmv(P, "LAit", s2, out=v2)
            be.trmv(P, "LTit", v2, out=t2)
        bcast(tt2)
        with ph("ktk2"):
            c1, c2 = be.ktk2(K, t1, t2)
        cc2 = torch.stack((c1, c2))
        ar(cc2)
        if owned:
            u = be.trmv(P, "LTi", cc2[0], alpha=1.0 / n, beta=lam, z=v)
            be.trmv(P, "LAi", u, out=out1)
            u = be.trmv(P, "LTi", cc2[1], alpha=1.0 / n, beta=lam, z=v2)
            be.trmv(P, "LAi", u, out=out2)

    b0 = ar(b0)                                        # K' (y / n), summed over shards
    one_call = shard is None and allreduce is None and phase is None and hasattr(be, "cg_solve")
    if one_call and getattr(K, "fmt", "f32") != "f32":
        # compact-format block: the class-batched library loop with a batch of one (odx_falkon_cg_batched_q_f64), where the
        # block's pass configuration has one and the factors are one contiguous block
        one_call = bool(hasattr(be, "cg_batched_supported") and getattr(P, "block_rows", None) is not None
                        and be.cg_batched_supported([K.n], [K.M], K.fmt))
    if one_call:
        # one shard, nothing to time per kernel family: the loop below as one library call (odx_falkon_cg_f64)
        if precond_ready is not None:
            precond_ready()
            precond_ready = None           # (waited for: the statement-by-statement loop below must not wait again)
        if K.fmt != "f32":
            b0s = be.zeros((M + 1) // 2 * 2).view(1, -1)
            b0s[0, :M].copy_(b0)
            got = be.cg_solve_batched([K], [P], b0s, [n], lam, maxiter, opt)
            alpha = None if got is None else got[0, :M].clone()
        else:
            alpha = be.cg_solve(K, P, b0, n, lam, maxiter, opt)
    if one_call and alpha is not None:
        if opt.check_pivots:
            _check_pivots(be, P)
        if return_knm:
            return alpha, K
        return alpha
    X, R, Pv, AP = be.zeros(M), be.zeros(M), be.zeros(M), be.zeros(M)
    AX = be.zeros(M) if can_fold else None
    state = be.zeros(4)
    if owned:
        if precond_ready is not None:
            precond_ready()
        B = be.trmv(P, "LAi", be.trmv(P, "LTi", b0))   # A^-T T^-T b0
        be.cg_init(B, X, R, Pv, state)
    tol = opt.cg_tolerance ** 2
    for it in range(maxiter):
        full = (it + 1) % opt.cg_full_gradient_every == 0
        fold = can_fold and full and it != maxiter - 1
        if fold:
            mmv2(Pv, AP, X, AX)                        # W p and W x_old
        else:
            mmv(Pv, AP)
        if owned:
            be.cg_step(X, R, Pv, AP, state, opt.cg_epsilon, full)
        if it == maxiter - 1:
            break    # the residual / direction update of the last step cannot change the returned X
        if full and fold:
            if owned:
                be.cg_residual(B, AX, AP, state, R)    # R = B - (W x_old + a W p) = B - W x_new
        elif full:
            mmv(X, AP)
            if owned:
                R.copy_(B)
                be.axpby(-1.0, AP, 1.0, R)             # R = B - mmv(X)
        if owned:
            be.cg_finish(R, Pv, state, opt.cg_epsilon, tol)
    alpha = be.zeros(M)
    if owned:
        be.trmv(P, "LTit", be.trmv(P, "LAit", X), out=alpha)   # T^-1 A^-1 beta
        if opt.check_pivots:
            _check_pivots(be, P)
    bcast(alpha)
    if return_knm:
        return alpha, K
    return alpha


def falkon_fit_lockstep(be, F, ys, Zfs, sigma, lam, maxiter=20, opt=None, n_total=None, shard=None, knm_outs=None,
                        phase=None, precond=None, precond_ready=None, owners=None):
    """Fit up to `world` binary problems at once over row shards, one owner rank per problem.

    Problem b (labels ys[b], centres Zfs[b]) is owned by rank owners[b] (default: rank b): only that rank holds its
    preconditioner and CG state.  (`owners`: B distinct ranks, the same list on every rank — odx.plan rotates them over
    the ranks when a batch is smaller than the world, so that the preconditioner work stays balanced.)  All problems advance through the same CG schedule in lock step, so one iteration costs every rank
        its own problem's triangular products                      (all ranks busy: no owner-only serial section)
        one all-gather of the B directions T^-1 A^-1 p             ((world, M) f64)
        B passes over its row shard of the B stored K_nM           (HBM-bound, the bulk)
        one reduce-scatter handing each owner the sum of its partials
    instead of, per problem, a broadcast, a pass and an all-reduce with the other ranks idle during the owner's
    algebra (falkon_fit's owner mode).  The arithmetic per problem is exactly falkon_fit's.

    ys, Zfs     lists of B <= world label vectors / centre Features (identical on every rank)
    knm_outs    optional list of B preallocated f32 buffers for the K_nM shards
    precond     this rank's problem's preconditioner (when it owns one), or None to build it here
    returns     list of B alpha vectors (M,) f64, on every rank
    """
    from .dist import RowShard
    opt = opt or SolverOptions()
    shard = shard if shard is not None else RowShard()
    world, rank = shard.world, shard.rank
    B = len(Zfs)
    if B > world or len(ys) != B:
        raise ValueError("falkon_fit_lockstep: %d problems for %d ranks" % (B, world))
    n = float(F.n if n_total is None else n_total)
    M = Zfs[0].n
    if any(z.n != M for z in Zfs):
        raise ValueError("falkon_fit_lockstep: every problem needs the same number of centres")
    ph = phase if phase is not None else (lambda name: _NoPhase())
    owners = list(range(B)) if owners is None else [int(o) for o in owners]
    if len(owners) != B or len(set(owners)) != B or any(not 0 <= o < world for o in owners):
        raise ValueError("falkon_fit_lockstep: owners must be %d distinct ranks below %d, got %r" % (B, world, owners))
    owned = rank in owners
    P = precond
    if owned and P is None:
        with ph("precond"):
            P = be.precond(Zfs[owners.index(rank)], sigma, lam, opt.pc_epsilon)
    Mp = (M + 1) // 2 * 2                             # rows of the exchanged matrices stay 16-byte aligned
    Tall = be.zeros(world * Mp).view(world, Mp)       # gathered directions: row r = the direction of the problem rank r owns
    CC = be.zeros(world * Mp).view(world, Mp)         # this rank's partials: row r = its partial of the problem rank r owns
    Ks = []
    for b in range(B):
        with ph("knm"):                               # K_nM shard and this shard's K' (y / n) of problem b in one launch
            Ks.append(be.knm_rhs(F, Zfs[b], sigma, ys[b] * (1.0 / n), out=None if knm_outs is None else knm_outs[b],
                                 rhs_out=CC[owners[b], :M])[0])
    tbuf, ccbuf, v = be.zeros(Mp), be.zeros(Mp), be.zeros(M)
    t, cc = tbuf[:M], ccbuf[:M]

    def passes():
        for b in range(B):
            with ph("ktk"):
                be.ktk(Ks[b], v=Tall[owners[b], :M], out=CC[owners[b], :M])
        return shard.reduce_scatter_rows(CC, ccbuf)

    def mmv(s, out):
        if owned:
            be.trmv(P, "LAit", s, out=v)
            be.trmv(P, "LTit", v, out=t)
        shard.gather_rows(tbuf, Tall)
        passes()
        if owned:
            u = be.trmv(P, "LTi", cc, alpha=1.0 / n, beta=lam, z=v)
            be.trmv(P, "LAi", u, out=out)

    # the periodic full residual folded into the neighbouring step's pass (see falkon_fit): the exchanged matrices carry
    # two vectors per problem for that one iteration — still one all-gather and one reduce-scatter
    can_fold = B > 0 and _can_fold(be, Ks[0], M, int(n) // world)
    if can_fold:
        tbuf2, ccbuf2, v2 = be.zeros(2 * Mp), be.zeros(2 * Mp), be.zeros(M)
        Tall2 = be.zeros(world * 2 * Mp).view(world, 2 * Mp)
        CC2 = be.zeros(world * 2 * Mp).view(world, 2 * Mp)

    def mmv2(s1, out1, s2, out2):
        if owned:
            be.trmv(P, "LAit", s1, out=v)
            be.trmv(P, "LTit", v, out=tbuf2[:M])
            be.trmv(P, "LAit", s2, out=v2)
            be.trmv(P, "LTit", v2, out=tbuf2[Mp:Mp + M])
        shard.gather_rows(tbuf2, Tall2)
        for b in range(B):
            with ph("ktk2"):
                o = owners[b]
                be.ktk2(Ks[b], Tall2[o, :M], Tall2[o, Mp:Mp + M], out1=CC2[o, :M], out2=CC2[o, Mp:Mp + M])
        shard.reduce_scatter_rows(CC2, ccbuf2)
        if owned:
            u = be.trmv(P, "LTi", ccbuf2[:M], alpha=1.0 / n, beta=lam, z=v)
            be.trmv(P, "LAi", u, out=out1)
            u = be.trmv(P, "LTi", ccbuf2[Mp:Mp + M], alpha=1.0 / n, beta=lam, z=v2)
            be.trmv(P, "LAi", u, out=out2)

    shard.reduce_scatter_rows(CC, ccbuf)             # K' (y / n) of every problem: each owner gets the sum of its row
    X, R, Pv, AP, AX = be.zeros(M), be.zeros(M), be.zeros(M), be.zeros(M), be.zeros(M)
    state = be.zeros(4)
    Bv = None
    if owned:
        if precond_ready is not None:
            precond_ready()
        Bv = be.trmv(P, "LAi", be.trmv(P, "LTi", cc))
        be.cg_init(Bv, X, R, Pv, state)
    tol = opt.cg_tolerance ** 2
    for it in range(maxiter):
        full = (it + 1) % opt.cg_full_gradient_every == 0
        fold = can_fold and full and it != maxiter - 1
        if fold:
            mmv2(Pv, AP, X, AX)
        else:
            mmv(Pv, AP)
        if owned:
            be.cg_step(X, R, Pv, AP, state, opt.cg_epsilon, full)
        if it == maxiter - 1:
            break
        if full and fold:
            if owned:
                be.cg_residual(Bv, AX, AP, state, R)
        elif full:
            mmv(X, AP)
            if owned:
                R.copy_(Bv)
                be.axpby(-1.0, AP, 1.0, R)
        if owned:
            be.cg_finish(R, Pv, state, opt.cg_epsilon, tol)
    tbuf.zero_()
    if owned:
        be.trmv(P, "LTit", be.trmv(P, "LAit", X), out=t)
        if opt.check_pivots:
            _check_pivots(be, P)
    shard.gather_rows(tbuf, Tall)
    return [Tall[owners[b], :M].clone() for b in range(B)]
