"""The big-n job of SURVEY §8e's first row as a schedule: C one-vs-rest FALKON fits + score-all over rows sharded on the
ranks of one node, classes advancing in lock-step batches of `world`, preconditioners built ahead on side streams.

This is what `bench.py` times; it lives in the package so that its CONTROL FLOW — which rank owns which class of a batch,
how the owned classes are grouped into class-batched preconditioner chains (the group size must be the same on every rank:
the centres of a group's classes are assembled with collectives), which collectives every rank issues in which order —
runs under gloo on CPU ranks with the tests' oracle backend at world sizes no single-GPU box can host
(tests/test_dist_gloo.py: 8 ranks, 30 classes -> batches of 8, 8, 8, 6 and preconditioner groups of 1 + 3).

The reference has no counterpart (single process, class-by-class loop: OnlineRegionClassifier_incore.py:96-155); the
arithmetic per class is odx.solver.falkon_fit_lockstep's, i.e. falkon_fit's.
"""
import contextlib

import torch

from . import solver
from .dist import RowShard


class _Side:
    """A side stream with an event, or (CPU tensors) nothing: work is then simply done in program order."""

    def __init__(self, device, be=None, cus=0):
        self.cuda = torch.device(device).type == "cuda"
        self.stream = None
        if self.cuda and cus > 0 and be is not None and hasattr(be, "masked_stream"):
            # confined to `cus` compute units (spread over the XCDs): the factorisation chains then pack their workgroups on
            # those instead of holding one slot on many CUs, each of which takes no Gaussian workgroup meanwhile
            self.stream = be.masked_stream(cus)
        elif self.cuda:
            self.stream = torch.cuda.Stream(device=device)

    def after_current(self):
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())

    def __enter__(self):
        self._ctx = torch.cuda.stream(self.stream) if self.cuda else contextlib.nullcontext()
        return self._ctx.__enter__()

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)

    def mark(self):
        """An event after everything queued on the side stream so far (None on the CPU)."""
        if not self.cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev


def _wait(ev):
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def precond_groups(n_batches, G):
    """The lock-step batches grouped for the class-batched preconditioner chains: sizes 1, 2, 3 (each only while smaller
    than G), then G — nothing but the first class's preconditioner is waited for at the start of a job, every later group
    is built while the group before it is being fitted.  Returns a list of lists of batch indices."""
    groups, g0 = [], 0
    for size in (1, 2, 3):
        if g0 < n_batches and size < G:
            groups.append(list(range(g0, min(g0 + size, n_batches))))
            g0 += size
    while g0 < n_batches:
        groups.append(list(range(g0, min(g0 + G, n_batches))))
        g0 += G
    return groups


class LockstepClassJob:
    """C binary problems on the same (sharded) rows X: labels(c) -> this rank's f64 label vector, centre_idx[c] -> global
    row ids of class c's Nystroem centres (identical on every rank)."""

    def __init__(self, be, X, n_total, M, labels, centre_idx, sigma, lam, maxiter=20, opt=None, shard=None,
                 precond_batch=0, precond_depth=2, precond_after_fit=False, classes=None, precond_cus=0):
        self.be, self.X, self.N, self.M = be, X, int(n_total), int(M)
        self.labels, self.cidx = labels, centre_idx
        self.sigma, self.lam, self.maxiter = sigma, lam, maxiter
        self.opt = opt or solver.SolverOptions(check_pivots=False)
        self.shard = shard if shard is not None else RowShard()
        self.world, self.rank = self.shard.world, self.shard.rank
        self.lo, self.hi = self.shard.bounds(self.N)
        self.n_loc = self.hi - self.lo
        dev = X.device
        self.C = len(centre_idx) if classes is None else int(classes)
        self.ldk = (self.M + 3) // 4 * 4
        # N x M entries per rank in all: one K_nM shard per class of the batch in flight, in the backend's storage format
        kbytes = be.knm_bytes(self.n_loc, self.M) if hasattr(be, "knm_bytes") else self.n_loc * self.ldk * 4
        self.kbufs = [torch.empty(max(kbytes, 16), dtype=torch.uint8, device=dev) for _ in range(self.world)]
        self.scores = torch.empty((self.n_loc, self.C), dtype=torch.float32, device=dev)
        self.depth = precond_depth if precond_depth > 0 else 2
        self.after_fit = bool(precond_after_fit)
        # the group size is derived from the number of lock-step batches (the same on every rank), never from how many
        # classes this rank happens to own
        n_batches = (self.C + self.world - 1) // self.world
        self.G = precond_batch if precond_batch > 0 else max(1, min(6, n_batches))
        self.ld_p = (self.M + 1) // 2 * 2
        self.nslot = self.depth + 1
        self.sides = [_Side(dev, be, precond_cus) for _ in range(self.nslot)] if self.G == 1 else []
        self.gside = _Side(dev, be, precond_cus) if self.G > 1 else None
        self.pbuf, self.pgroup = [], []
        self.trace = []          # (kind, payload) records of the schedule this rank executed (tests read it)

    def release(self):
        self.kbufs, self.pbuf, self.pgroup, self.scores = [], [], [], None

    # ------------------------------------------------------------------ pieces
    def gather_centres(self, idx):
        """Z = X_global[idx]: every rank contributes the rows it owns, one all-reduce sums them."""
        # no boolean-mask indexing here: it would make the host wait for the GPU (nonzero), and the ~1500 launches of the
        # next preconditioner are then enqueued while the main stream has nothing to run (33 ms per class, measured)
        be, X = self.be, self.X
        if self.world == 1:
            return be.features(X.index_select(0, idx))
        mine = (idx >= self.lo) & (idx < self.hi)
        Z = X.index_select(0, (idx - self.lo).clamp_(0, max(self.n_loc - 1, 0)))
        Z *= mine.unsqueeze(1)
        self.shard.allreduce(Z)
        return be.features(Z)

    def _prepare(self, batch, slot, ph, infos):
        """Per-class mode (G == 1): centres of the batch's classes (one all-reduce each, main stream) and, on the slot's side
        stream, the preconditioner of the class this rank owns in the batch (owner = position in the batch)."""
        be, dev = self.be, self.X.device
        while len(self.pbuf) < self.nslot:
            self.pbuf.append(torch.empty((4, self.M, self.ld_p), dtype=torch.float64, device=dev))
        Zs = [self.gather_centres(self.cidx[c]) for c in batch]
        P, ev = None, None
        if self.rank < len(batch):
            side = self.sides[slot]
            side.after_current()          # the slot's last reader is done, the centres exist
            with side:
                with ph("precond"):
                    kw = {"out": self.pbuf[slot], "ws_key": "precond%d" % slot} if hasattr(be, "precond_batched") else {}
                    P = be.precond(Zs[self.rank], self.sigma, self.lam, self.opt.pc_epsilon, **kw)
                ev = side.mark()
            if infos is not None and hasattr(P, "info"):
                infos.append(P.info)
            self.trace.append(("precond", (batch[self.rank],)))
        return Zs, P, ev

    def _prepare_group(self, group, slot, ph, infos):
        """Class-batched mode: centres of every class of the group's batches (main stream) and, on the side stream, the
        preconditioners of the classes this rank owns among them, all by one batched call.  One (Zs, P, event) per batch."""
        be, dev = self.be, self.X.device
        Zs_all = [[self.gather_centres(self.cidx[c]) for c in batch] for batch in group]
        own = [k for k, batch in enumerate(group) if self.rank < len(batch)]
        Ps, ev = {}, None
        if own:
            self.gside.after_current()    # the slot's last readers were issued, the centres exist
            with self.gside:
                with ph("precond"):
                    zf = [Zs_all[k][self.rank] for k in own]
                    if hasattr(be, "precond_batched"):
                        while len(self.pgroup) < 2:
                            self.pgroup.append(torch.empty((self.G, 4, self.M, self.ld_p), dtype=torch.float64, device=dev))
                        plist = be.precond_batched(zf, self.sigma, self.lam, self.opt.pc_epsilon,
                                                   out=self.pgroup[slot][:len(own)], ws_key="precond_group")
                    else:                 # a backend without the batched chain (tests' oracle backend): one after the other
                        plist = [be.precond(z, self.sigma, self.lam, self.opt.pc_epsilon) for z in zf]
                ev = self.gside.mark()
            Ps = dict(zip(own, plist))
            if infos is not None:
                infos.extend(p.info for p in plist if hasattr(p, "info"))
            self.trace.append(("precond", tuple(group[k][self.rank] for k in own)))
        return [(Zs_all[k], Ps.get(k), ev) for k in range(len(group))]

    # ------------------------------------------------------------------ the schedule
    def run(self, F, classes=None, phases=None, infos=None, alphas_out=None):
        """Fit and score `classes` (default: all).  F: Features of this rank's rows.  phases: name -> context manager
        (bench.py's HIP-event timers) or None; infos: list collecting the Cholesky status words; alphas_out: optional dict
        that receives class -> alpha (M,) f64 of every fitted class.  Returns (alpha, Zf) of the last class."""
        be, world, rank = self.be, self.world, self.rank
        classes = list(range(self.C)) if classes is None else list(classes)
        ph = (lambda name: phases[name]) if phases is not None else (lambda name: contextlib.nullcontext())
        batches = [classes[b0:b0 + world] for b0 in range(0, len(classes), world)]
        out = None
        if self.G > 1:
            groups = precond_groups(len(batches), self.G)
            first_of = {grp[0]: gi for gi, grp in enumerate(groups)}
            ready = dict(zip(groups[0], self._prepare_group([batches[bi] for bi in groups[0]], 0, ph, infos)))
        else:
            ready = {bi: self._prepare(batches[bi], bi % self.nslot, ph, infos) for bi in range(min(self.depth, len(batches)))}
        for bi, batch in enumerate(batches):
            if self.G > 1:
                gi = first_of.get(bi)
                if gi is not None and gi + 1 < len(groups):          # one group ahead, on the side stream
                    ready.update(zip(groups[gi + 1], self._prepare_group([batches[k] for k in groups[gi + 1]], (gi + 1) % 2, ph, infos)))
            elif not self.after_fit and bi + self.depth < len(batches):
                ready[bi + self.depth] = self._prepare(batches[bi + self.depth], (bi + self.depth) % self.nslot, ph, infos)
            Zs, P, ev = ready.pop(bi)
            ys = [self.labels(c) for c in batch]
            mine = rank < len(batch)
            self.trace.append(("fit", tuple(batch)))
            alphas = solver.falkon_fit_lockstep(be, F, ys, Zs, self.sigma, self.lam, self.maxiter, self.opt, n_total=self.N,
                                                shard=self.shard, knm_outs=self.kbufs[:len(batch)],
                                                phase=(lambda name: phases[name]) if phases is not None else None,
                                                precond=P if mine else None,
                                                precond_ready=(lambda: _wait(ev)) if mine else None)
            if self.G == 1 and self.after_fit and bi + self.depth < len(batches):
                # issued behind this batch's CG in stream order: the factorisations then run beside the MFMA-bound scoring
                # of this batch and K_nM build of the next, and the HBM-bound passes keep the chip to themselves
                ready[bi + self.depth] = self._prepare(batches[bi + self.depth], (bi + self.depth) % self.nslot, ph, infos)
            if alphas_out is not None:
                alphas_out.update((c, alphas[pos]) for pos, c in enumerate(batch))
            for pos, c in enumerate(batch):
                with ph("mmv"):
                    be.mmv(F, Zs[pos], self.sigma, alphas[pos], None, out=self.scores[:, c:c + 1])
            out = (alphas[-1], Zs[-1])
        return out
