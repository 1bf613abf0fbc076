"""falkon-shaped estimator objects backed by libodx.

The reference never touches FALKON's arithmetic itself; it constructs
``kernels.GaussianKernel(sigma)``, ``FalkonOptions(...)`` and ``InCoreFalkon`` / ``Falkon`` and
calls ``fit`` / ``predict`` / ``kernel.mmv``
(src/modules/region-classifier/FALKONWrapper_with_centers_selection_incore.py:50-82,
 FALKONWrapper_with_centers_selection.py:49-78,
 mrcnn_modified/modeling/roi_heads/box_head/roi_box_predictors.py:136-158,
 mrcnn_modified/modeling/rpn/rpn.py:197-225).  This module provides objects with that
surface — constructor keywords, ``fit(X, Y)``, ``predict(X) -> (n, 1)``, attributes
``ny_points_ / alpha_ / M / kernel`` (re-assignable, deep-copyable, picklable) — so that
the drop-in wrappers in ``src/modules/region-classifier`` stay as thin as the reference's,
and so that ``sys.modules['falkon'] = odx.falkon`` is a valid binding for the reference's
own wrapper files.
"""
import os
import types

import torch

from . import backend as _backend
from . import options as _options
from .solver import SolverOptions, falkon_fit


class FalkonOptions:
    """Accepts falkon's option keywords.  The ones that steer arithmetic are honoured
    (pc_epsilon_32, cg_epsilon_32, cg_tolerance, cg_full_gradient_every); the memory /
    dispatch knobs the reference passes (min_cuda_iter_size_*, min_cuda_pc_size_*,
    keops_active, store_kernel_d_threshold — FALKONWrapper_..._incore.py:56) have nothing to
    select here: K_nM is always materialised on the GPU and nothing runs on the CPU."""

    def __init__(self, **kw):
        self.pc_epsilon_32 = kw.pop("pc_epsilon_32", 1e-5)
        self.cg_epsilon_32 = kw.pop("cg_epsilon_32", 1e-7)
        self.cg_tolerance = kw.pop("cg_tolerance", 1e-7)
        self.cg_full_gradient_every = kw.pop("cg_full_gradient_every", 10)
        self.extra = dict(kw)

    def solver_options(self):
        return SolverOptions(pc_epsilon=self.pc_epsilon_32, cg_epsilon=self.cg_epsilon_32,
                             cg_tolerance=self.cg_tolerance, cg_full_gradient_every=self.cg_full_gradient_every)


class GaussianKernel:
    """K(x, z) = exp(-|x - z|^2 / (2 sigma^2)); ``mmv`` is the fused scoring kernel."""
    kernel_name = "gaussian"

    def __init__(self, sigma, opt=None):
        self.sigma = float(sigma)

    def mmv(self, X1, X2, v, out=None, opt=None):
        """K(X1, X2) @ v -> (n, T) f32.  A block-structured v (the heads' alpha_parallel) is
        detected and only its non-zero row range per column is visited."""
        be = _backend.get_backend()
        F = be.features(X1)
        Zf = X2 if (hasattr(X2, "sq") or hasattr(X2, "n")) and not torch.is_tensor(X2) else be.features(X2)     # a backend's Features
        v = torch.as_tensor(v)
        if v.dim() == 1:
            v = v[:, None]
        ranges = block_ranges(v) if v.shape[1] > 1 else None      # one column (model.predict): every centre counts
        res = be.mmv(F, Zf, self.sigma, v, ranges)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def __repr__(self):
        return "GaussianKernel(sigma=%g)" % self.sigma


def block_ranges(v):
    """Per column: [first non-zero row, last non-zero row + 1) as an int32 (T, 2) tensor."""
    nz = v != 0
    Mtot, T = v.shape
    if Mtot == 0:
        return torch.zeros((T, 2), dtype=torch.int32, device=v.device)
    idx = torch.arange(Mtot, device=v.device)[:, None]
    big = torch.where(nz, idx, torch.full_like(idx, Mtot)).amin(0)
    end = torch.where(nz, idx + 1, torch.zeros_like(idx)).amax(0)
    lo = torch.minimum(big, end)
    return torch.stack([lo, end], dim=1).to(torch.int32)


class _FalkonBase:
    """Common fit / predict.  ``center_selection.select(X, Y)`` follows MyCenterSelector
    (src/modules/region-classifier/MyCenterSelector.py:7-15)."""
    _cpu_model = False

    def __init__(self, kernel, penalty, M, center_selection="uniform", maxiter=20, seed=None,
                 error_fn=None, error_every=1, options=None, **_ignored):
        self.kernel = kernel
        self.penalty = penalty
        self.M = M
        self.maxiter = maxiter
        self.center_selection = center_selection
        self.options = options if options is not None else FalkonOptions()
        self.seed = seed
        self.alpha_ = None
        self.ny_points_ = None

    def fit(self, X, Y, Xts=None, Yts=None):
        be = _backend.get_backend()
        F = be.features(X)
        if isinstance(self.center_selection, str):
            idx = torch.randperm(F.n)[: self.M]
            Zf = be.rows(F, idx)
        else:
            sel = self.center_selection.select(F.X, None)
            Z = sel[0] if isinstance(sel, tuple) else sel
            Zf = be.features(Z)
        self.M = Zf.n
        y = torch.as_tensor(Y).reshape(F.n, -1)
        if y.shape[1] != 1:
            raise ValueError("odx FALKON fits one right-hand side per model (the reference trains one "
                             "binary classifier per fit); got Y with %d columns" % y.shape[1])
        yv = be.vec(y[:, 0])
        alpha = falkon_fit(be, F, yv, Zf, self.kernel.sigma, float(self.penalty), int(self.maxiter),
                           self.options.solver_options())
        if hasattr(be, "release_helper_streams"):
            be.release_helper_streams()          # (a chain of 4096 centres or more made helper streams: not left behind idle)
        ny = Zf.X.contiguous() if Zf.X.stride(0) != Zf.D else Zf.X
        self.alpha_ = alpha.reshape(-1, 1)
        self.ny_points_ = ny
        if self._cpu_model:
            self.alpha_ = self.alpha_.cpu()
            self.ny_points_ = self.ny_points_.cpu()
        else:
            self._centres(Zf)
        return self

    def _centres(self, Zf=None):
        """The centres as kernel operands (row norms, packed f16 split), kept with the model while `ny_points_` is the
        same, unmodified tensor: a predict is ~100 us of GPU work and re-deriving them was a third of its launches.
        Same bits as a fresh derivation (the split's scale comes from the centres alone)."""
        be, ny = _backend.get_backend(), self.ny_points_
        if Zf is not None:
            if ny is Zf.X and getattr(Zf, "own_pack", True):
                self._zf = (ny, ny._version, be, Zf)
            return Zf
        c = self.__dict__.get("_zf")
        if c is not None and c[0] is ny and c[1] == ny._version and c[2] is be:
            if ny.is_cuda:          # possibly made on another stream than the one that scores with them now
                cur = torch.cuda.current_stream()
                for t in (c[3].X, c[3].sq, getattr(c[3], "P", None), getattr(c[3], "meta", None)):
                    if t is not None:
                        t.record_stream(cur)
            return c[3]
        Zf = be.features(ny)
        if torch.is_tensor(ny):
            self._zf = (ny, ny._version, be, Zf)
        return Zf

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_zf", None)          # derived data, tied to this process's backend: not copied, not pickled
        return d

    def predict(self, X):
        if self.alpha_ is None:
            raise RuntimeError("predict called before fit")
        res = self.kernel.mmv(X, self._centres(), self.alpha_)
        if self._cpu_model and not (torch.is_tensor(X) and X.is_cuda):
            res = res.cpu()
        return res

    def __bool__(self):
        return True


def fit_batch(estimators, Xs, Ys, streams=None):
    """Fit several independent estimators (the classes of one Minibootstrap round) with ONE batched preconditioner
    launch chain (backend.precond_batched) instead of one ~400-launch chain each; the K_nM build and the CG of every
    class then run as in ``fit`` — on `streams` round-robin when given (the classes are independent), each reusing its
    slice of the batched factors.  Every estimator ends up exactly as its own ``fit(X, Y)`` would leave it (same bits:
    the batched factors equal the single-class ones).  The reference has no counterpart: it trains the classes one
    after the other (OnlineRegionClassifier_incore.py:96-155)."""
    be = _backend.get_backend()
    if not hasattr(be, "precond_batched"):
        for est, X, Y in zip(estimators, Xs, Ys):
            est.fit(X, Y)
        return estimators
    # First only what the factorisation chains need — the centres — so that the GPU starts on K_MM while the host is still
    # preparing the rest: the row norms of the training sets and the uploads of the labels (a launch / a copy per class, ~2 ms
    # of host time for the 30 classes of a round) are issued behind the chains (`finish` below), for the K_nM builds.
    Fs, Zfs, yvs = [], [], []
    for est, X, Y in zip(estimators, Xs, Ys):
        if isinstance(est.center_selection, str):
            F = be.features(X)                           # (the uniform selection gathers norms and packed rows along)
            Zf = be.rows(F, torch.randperm(F.n)[: est.M])
        else:
            F = be.features(X, norms=False) if hasattr(be, "row_matrix") else be.features(X)
            sel = est.center_selection.select(F.X, None)
            Zf = be.features(sel[0] if isinstance(sel, tuple) else sel)
        est.M = Zf.n
        y = torch.as_tensor(Y).reshape(F.n, -1)
        if y.shape[1] != 1:
            raise ValueError("odx FALKON fits one right-hand side per model; got Y with %d columns" % y.shape[1])
        Fs.append(F), Zfs.append(Zf), yvs.append(y)

    def finish(i):
        if getattr(Fs[i], "sq", 0) is None:
            Fs[i] = be.features(Fs[i].X)
        if yvs[i].dim() == 2:
            yvs[i] = be.vec(yvs[i][:, 0])
    # classes that share (sigma, penalty, jitter) share a chain; the reference's classes always do
    groups = {}
    for i, est in enumerate(estimators):
        o = est.options.solver_options()
        groups.setdefault((float(est.kernel.sigma), float(est.penalty), o.pc_epsilon), []).append(i)
    from . import solver as _solver
    cur = torch.cuda.current_stream() if streams else None
    for (sigma, lam, eps), members in groups.items():
        for c0 in range(0, len(members), be.MAX_CLASS_BATCH):
            chunk = members[c0:c0 + be.MAX_CLASS_BATCH]
            opts = [estimators[i].options.solver_options() for i in chunk]
            iters = {int(estimators[i].maxiter) for i in chunk}
            same = len(iters) == 1 and all(o == opts[0] for o in opts) and hasattr(be, "cg_solve_batched")
            if same and hasattr(be, "knm_format"):
                # the lock-step library loop needs one storage format and one pass configuration for the chunk: decided BEFORE
                # anything is built (otherwise every K_nM block would be built here, held, and built again below)
                fmts = {be.knm_format(Fs[i].n, Zfs[i].n) for i in chunk}
                same = len(fmts) == 1 and (not hasattr(be, "cg_batched_supported") or
                                           be.cg_batched_supported([Fs[i].n for i in chunk], [Zfs[i].n for i in chunk], fmts.pop()))
            if same:
                Mmax = max(Zfs[i].n for i in chunk)
                b0s = torch.zeros((len(chunk), (Mmax + 1) // 2 * 2), dtype=torch.float64, device=be.device)
                if streams:
                    for s in streams:           # before the chain is queued: the builds below start beside it, not after it
                        s.wait_stream(cur)
            if streams and len(chunk) >= 2 * int(_options.current().chain_split_min):
                # two half chains side by side: a chain is a dependent sequence of short launches, a third of them one
                # workgroup per class (the 128 x 128 diagonal factorisations), and a half's products fill what the other
                # half's diagonal steps leave idle.  Both write their classes' slots of ONE factor block (the lock-step CG
                # wants a uniform stride); per class the factors are those of any other grouping, bit for bit.
                Mmax_c = max(Zfs[i].n for i in chunk)
                block = torch.empty((len(chunk), 4, Mmax_c, (Mmax_c + 1) // 2 * 2), dtype=torch.float64, device=be.device)
                h = (len(chunk) + 1) // 2
                side = _chain_stream(0)
                side.wait_stream(cur)
                # (neither half on the caller's stream — usually the device's default stream: measured 4 % of a Minibootstrap
                # round faster than with the first half there)
                side_a = _chain_stream(1)
                side_a.wait_stream(cur)
                with torch.cuda.stream(side_a):
                    Ps = be.precond_batched([Zfs[i] for i in chunk[:h]], sigma, lam, eps, out=block[:h], ws_key="precond_batched_fit", Mmax=Mmax_c)
                for P in Ps:
                    P.info.record_stream(cur)
                with torch.cuda.stream(side):
                    Pb = be.precond_batched([Zfs[i] for i in chunk[h:]], sigma, lam, eps, out=block[h:], ws_key="precond_batched_fit_b", Mmax=Mmax_c)
                for P in Pb:
                    P.info.record_stream(cur)
                Ps = Ps + Pb
            else:
                side = side_a = None
                Ps = be.precond_batched([Zfs[i] for i in chunk], sigma, lam, eps, ws_key="precond_batched_fit")
            for i in chunk:
                finish(i)
            if same and streams:
                for s in streams:               # the builds read the norms and labels just queued on the caller's stream
                    s.wait_stream(cur)
            alphas = None
            if same:
                # K_nM builds (f16 split of the rows, one Gaussian launch, the right-hand side: ~10 short kernels per
                # class) on the streams while the factorisation chain runs, then ALL CG loops of the chunk in lock step
                Ks = []
                for row, i in enumerate(chunk):
                    with (torch.cuda.stream(streams[row % len(streams)]) if streams else _nullcontext()):
                        K, _ = be.knm_rhs(Fs[i], Zfs[i], sigma, yvs[i] * (1.0 / Fs[i].n), rhs_out=b0s[row, :Zfs[i].n])
                        if streams:
                            K.K.record_stream(cur)      # allocated on a side stream, read by the CG on the caller's
                    Ks.append(K)
                if streams:
                    for s in streams:
                        cur.wait_stream(s)
                # (ONE lock-step CG over both halves: a CG per half, the first started beside the tail of the second chain,
                # measured slower — twice the launches, each half the width the triangular products stream best at)
                if side is not None:
                    cur.wait_stream(side_a)
                    cur.wait_stream(side)
                alphas = be.cg_solve_batched(Ks, Ps, b0s, [Fs[i].n for i in chunk], lam, iters.pop(), opts[0])
            if side is not None:
                cur.wait_stream(side_a)
                cur.wait_stream(side)
            for row, (i, P) in enumerate(zip(chunk, Ps)):
                est = estimators[i]
                if alphas is not None:
                    alpha = alphas[row, :Zfs[i].n].clone()
                    if opts[row].check_pivots:
                        _solver._check_pivots(be, P)
                else:       # classes in different pass configurations (or a backend without the batched loop): one CG each
                    if streams:
                        streams[i % len(streams)].wait_stream(cur)
                    with (torch.cuda.stream(streams[i % len(streams)]) if streams else _nullcontext()):
                        alpha = falkon_fit(be, Fs[i], yvs[i], Zfs[i], est.kernel.sigma, float(est.penalty), int(est.maxiter),
                                           opts[row], precond=P)
                ny = Zfs[i].X.contiguous() if Zfs[i].X.stride(0) != Zfs[i].D else Zfs[i].X
                est.alpha_, est.ny_points_ = alpha.reshape(-1, 1), ny
                if est._cpu_model:
                    est.alpha_, est.ny_points_ = est.alpha_.cpu(), est.ny_points_.cpu()
                else:
                    est._centres(Zfs[i])
    if streams:
        for s in streams:
            cur.wait_stream(s)
    if hasattr(be, "release_helper_streams"):
        be.release_helper_streams()              # (see InCoreFalkon.fit)
    return estimators


class BatchFit:
    """fit_batch in two phases, for callers that can hand over the classes of a round in groups: ``add`` prepares a group's
    centres and queues its factorisation chain at once (on a side stream of its own), ``finish`` queues everything that
    needs all groups — row norms, K_nM builds, ONE lock-step CG over all classes — and leaves every estimator as its own
    ``fit`` would (same bits).  Between two ``add`` calls the caller is free to synchronise with the host and prepare the
    next group while the first group's chain already runs (the Minibootstrap reads a group's selections, gathers its rows
    and draws its centres: ~2.5 ms of host work per half round that used to pass with the GPU idle).
    The fast path needs what a Minibootstrap round has: a GPU backend with the batched chain and CG, side streams, at most
    32 classes with one (sigma, penalty, jitter, iterations, options) and pre-chosen centres; anything else is collected and
    goes through fit_batch in ``finish``."""

    def __init__(self, streams=None):
        self.be = _backend.get_backend()
        self.streams = list(streams) if streams else None
        self.est, self.Xs, self.Ys = [], [], []
        self.Fs, self.Zfs, self.ys = [], [], []
        self.segments = []            # (first class, count, Preconds of the group, its chain's stream)
        self.block = None
        self.key = None
        self.fast = bool(self.streams) and hasattr(self.be, "precond_batched") and hasattr(self.be, "cg_solve_batched") \
            and hasattr(self.be, "row_matrix") and torch.cuda.is_available()
        self.cur = torch.cuda.current_stream() if self.fast else None

    def _key(self, est):
        o = est.options.solver_options()
        return (float(est.kernel.sigma), float(est.penalty), o.pc_epsilon, int(est.maxiter),
                (o.cg_epsilon, o.cg_tolerance, o.cg_full_gradient_every, o.check_pivots))

    def add(self, estimators, Xs, Ys, expect_total=None, centres_cap=None):
        """expect_total: classes the caller will add in all (sizes the shared factor block at the first group); centres_cap:
        the most centres any class can have (the block's slot size: a later group with more centres than the first group's
        largest class would otherwise force everything through the general path)."""
        be = self.be
        first = len(self.est)
        self.est += list(estimators)
        self.Xs += list(Xs)
        self.Ys += list(Ys)
        if not self.fast or not estimators:
            return
        if any(isinstance(e.center_selection, str) or e.center_selection is None for e in estimators):
            self.fast = False
            return
        keys = {self._key(e) for e in estimators}
        if len(keys) != 1 or (self.key is not None and keys != {self.key}):
            self.fast = False
            return
        self.key = next(iter(keys))
        self.opt = estimators[0].options.solver_options()
        total = max(int(expect_total or 0), len(self.est))
        if total > be.MAX_CLASS_BATCH:
            self.fast = False
            return
        Fs, Zfs, ys = [], [], []
        for est, X, Y in zip(estimators, Xs, Ys):
            F = be.features(X, norms=False)
            sel = est.center_selection.select(F.X, None)
            Zf = be.features(sel[0] if isinstance(sel, tuple) else sel)
            est.M = Zf.n
            y = torch.as_tensor(Y).reshape(F.n, -1)
            if y.shape[1] != 1:
                raise ValueError("odx FALKON fits one right-hand side per model; got Y with %d columns" % y.shape[1])
            Fs.append(F), Zfs.append(Zf), ys.append(y)
        Mmax = max(z.n for z in Zfs)
        sigma, lam, eps = self.key[0], self.key[1], self.key[2]
        if self.block is None:
            slot = max(Mmax, int(centres_cap or 0))
            self.block = torch.empty((total, 4, slot, (slot + 1) // 2 * 2), dtype=torch.float64, device=be.device)
        if Mmax > self.block.shape[2] or first + len(estimators) > self.block.shape[0]:
            self.fast = False          # (a later group with more centres than the block was cut for: everything through fit_batch)
            return
        g = len(self.segments)
        side = _chain_stream(g % 2)
        side.wait_stream(self.cur)
        with torch.cuda.stream(side):
            Ps = be.precond_batched(Zfs, sigma, lam, eps, out=self.block[first:first + len(estimators)],
                                    ws_key="precond_batched_fit" + ("_b" if g % 2 else ""), Mmax=int(self.block.shape[2]))
        for P in Ps:
            P.info.record_stream(self.cur)
        self.Fs += Fs
        self.Zfs += Zfs
        self.ys += ys
        self.segments.append((first, len(estimators), Ps, side))

    def abort(self):
        """Give up the round: the caller's stream waits for every chain queued so far (they read rows and caches the caller
        may free or rewrite next), nothing is fitted."""
        for _, _, _, side in self.segments:
            if side is not None and self.cur is not None:
                self.cur.wait_stream(side)
        self.segments = []

    def finish(self):
        be, cur, streams = self.be, self.cur, self.streams
        done = sum(c for _, c, _, _ in self.segments)
        if not self.fast or done != len(self.est):
            if self.segments:                 # chains already queued for some groups: let them drain, then the general path
                for _, _, _, side in self.segments:
                    cur.wait_stream(side)
            return fit_batch(self.est, self.Xs, self.Ys, streams=self.streams)
        from . import solver as _solver
        sigma, lam, eps, maxiter, opt = self.key[0], self.key[1], self.key[2], self.key[3], self.opt
        n_cls = len(self.est)
        Fs, Zfs = self.Fs, self.Zfs
        fmts = {be.knm_format(Fs[i].n, Zfs[i].n) for i in range(n_cls)}
        if len(fmts) != 1 or (hasattr(be, "cg_batched_supported")
                              and not be.cg_batched_supported([f.n for f in Fs], [z.n for z in Zfs], next(iter(fmts)))):
            for _, _, _, side in self.segments:
                cur.wait_stream(side)
            return fit_batch(self.est, self.Xs, self.Ys, streams=self.streams)
        for i in range(n_cls):                # behind the chains: row norms of the training sets, label uploads
            Fs[i] = be.features(Fs[i].X)
        yvs = [be.vec(y[:, 0]) for y in self.ys]
        Mmax = int(self.block.shape[2])
        b0s = torch.zeros((n_cls, (Mmax + 1) // 2 * 2), dtype=torch.float64, device=be.device)
        for s in streams:
            s.wait_stream(cur)
        Ks = []
        for i in range(n_cls):
            with torch.cuda.stream(streams[i % len(streams)]):
                K, _ = be.knm_rhs(Fs[i], Zfs[i], sigma, yvs[i] * (1.0 / Fs[i].n), rhs_out=b0s[i, :Zfs[i].n])
                K.K.record_stream(cur)
            Ks.append(K)
        for s in streams:
            cur.wait_stream(s)
        Ps = []
        for _, _, P, side in self.segments:
            cur.wait_stream(side)
            Ps += P
        alphas = be.cg_solve_batched(Ks, Ps, b0s, [f.n for f in Fs], lam, maxiter, opt)
        if alphas is None:
            return fit_batch(self.est, self.Xs, self.Ys, streams=self.streams)
        for i, (est, P) in enumerate(zip(self.est, Ps)):
            alpha = alphas[i, :Zfs[i].n].clone()
            if opt.check_pivots:
                _solver._check_pivots(be, P)
            ny = Zfs[i].X.contiguous() if Zfs[i].X.stride(0) != Zfs[i].D else Zfs[i].X
            est.alpha_, est.ny_points_ = alpha.reshape(-1, 1), ny
            if est._cpu_model:
                est.alpha_, est.ny_points_ = est.alpha_.cpu(), est.ny_points_.cpu()
            else:
                est._centres(Zfs[i])
        if hasattr(be, "release_helper_streams"):
            be.release_helper_streams()          # (see InCoreFalkon.fit)
        return self.est


_chain_streams = {}


def _chain_stream(k=0):
    """Side stream k (0, 1) of the calling stream for the half chains of fit_batch: the first two of the streams measured to
    sit on hardware queues of their own (odx/streams.py; the callers' class streams are taken from the ones behind them), a
    plain stream when the device has none to spare."""
    from . import streams as _streams
    own = _streams.distinct(2)
    if len(own) > k:
        return own[k]
    dev = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, k)
    if dev not in _chain_streams:
        _chain_streams[dev] = torch.cuda.Stream()
    return _chain_streams[dev]


class _nullcontext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class InCoreFalkon(_FalkonBase):
    """GPU-resident estimator (FALKONWrapper_with_centers_selection_incore.py:58-68)."""


class Falkon(_FalkonBase):
    """The reference's `--CPU` estimator (FALKONWrapper_with_centers_selection.py:53-64) keeps
    data and model on the host; here the rows are staged to the GPU for the fit and the model
    tensors are handed back on the host, so the callers' device moves stay valid."""
    _cpu_model = True


# `from falkon import kernels` / `from falkon.options import *` surfaces
kernels = types.SimpleNamespace(GaussianKernel=GaussianKernel)
options = types.SimpleNamespace(FalkonOptions=FalkonOptions)
