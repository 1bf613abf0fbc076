"""Per-class RLS bounding-box regressors (A7) and their application (A8).

Training follows RegionRefinerTrainer.train / solve
(src/modules/region-refiner/region_refiner_trainer/train_region_refiner.py:25-119): per class,
targets are centred and whitened in f64 (mu, S = Y'Y/n, eig, T = W diag((D + 1e-3)^-1/2) W',
T_inv), a bias column is appended to the features, R = chol(X'X + lam I) and each of the four
targets is solved with two triangular solves; per-sample losses 0.5 (X w - y)^2 ride along.
Here the O(n D^2) Gram, the Cholesky, the solves and the residuals are libodx kernels
(odx_rls_gram_f64 / odx_rls_solve_f64 / odx_rls_predict_rows_f64, all f64); the 4 x 4 whitening
is host-side glue.  The Gram accumulates over row shards, so with a RowShard the partial
(D+1)^2 + 4 (D+1) sums are all-reduced once per class before the solve.

Application follows RegionPredictor.predict (region_predictor/predict_regions.py:16-80).
"""
import os
import time

import numpy as np
import torch

from . import backend as _backend

RAW_TARGETS = True        # f32 targets: [Y 1]' X out of the Gram sweep (test hook: False = the two-sweep form f64 targets take)


def _device():
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def _rows(be, X):
    """The rows as the backend's operand container WITHOUT their squared norms (no Gaussian block is formed here: a read of
    every row saved); backends without row_matrix() (the tests' oracle backend) hand out their features()."""
    return be.row_matrix(X) if hasattr(be, "row_matrix") else be.features(X)


def sym_eig_small(S):
    """Eigen-decomposition of a batch of small symmetric matrices S (B, k, k) — the 4 x 4 target covariances of
    train_region_refiner.py:63 — in f64 ON THE HOST (numpy): (evals (B, k), V (B, k, k)) with S = V diag(evals) V'.  Sixteen
    numbers per class: a device solver's launch + synchronisation per batch costs more than the arithmetic (~8 ms against
    ~30 us), and T = V diag(.) V' depends neither on the eigenvalues' order nor on the vectors' signs."""
    A = np.asarray(S.detach().cpu().numpy(), dtype=np.float64)
    A = 0.5 * (A + np.swapaxes(A, -1, -2))
    ev, V = np.linalg.eigh(A)
    return torch.from_numpy(np.ascontiguousarray(ev)).to(S.device), torch.from_numpy(np.ascontiguousarray(V)).to(S.device)


def whitening_transforms(S, eps=0.001):
    """(T, T_inv) (B, 4, 4) f64 on S's device for a batch of target covariances S (B, 4, 4): T = V diag(1 / sqrt(ev + eps)) V',
    T_inv = V diag(sqrt(ev + eps)) V' (train_region_refiner.py:63-67) — eigen-decomposition AND the two small products on the
    host, one upload (on the device the products were five library GEMM launches for 30 x 16 numbers)."""
    A = np.asarray(S.detach().cpu().numpy(), dtype=np.float64)
    A = 0.5 * (A + np.swapaxes(A, -1, -2))
    ev, V = np.linalg.eigh(A)
    root = np.sqrt(ev + eps)
    Vt = np.swapaxes(V, -1, -2)
    both = np.stack(((V / root[..., None, :]) @ Vt, (V * root[..., None, :]) @ Vt))
    both = torch.from_numpy(np.ascontiguousarray(both)).to(S.device)
    return both[0], both[1]


def whiten_targets(Yi):
    """mu, T, T_inv (f64) of train_region_refiner.py:61-67 for targets Yi (n, 4) f64."""
    mu = torch.mean(Yi, dim=0)
    Yc = Yi - mu
    S = torch.matmul(Yc.t(), Yc) / Yc.size()[0]
    evals, W = sym_eig_small(S)         # S is symmetric: same T as the reference's general eig
    root = torch.sqrt(evals + 0.001)
    T = W @ torch.diag(1.0 / root) @ W.t()
    T_inv = W @ torch.diag(root) @ W.t()
    return mu, Yc, T, T_inv


class RegionRefinerTrainer:
    def __init__(self, cfg, lmbd, is_rpn, shard=None):
        self.cfg = cfg
        self.lambd = lmbd
        self.COXY = None
        self.is_rpn = is_rpn
        self.shard = shard

    def __call__(self, COXY, output_dir=None):
        self.COXY = COXY
        return self.train(output_dir=output_dir)

    def train(self, output_dir=None):
        be = _backend.get_backend()
        if hasattr(be, "rls_train_batched") and len(self.cfg['CHOSEN_CLASSES']) - (0 if self.is_rpn else 1) > 1:
            return self._train_batched(be, output_dir)
        return self._train_sequential(be, output_dir)

    def _train_batched(self, be, output_dir=None):
        """All classes at once.  The reference's loop (train_region_refiner.py:27-98) handles one class per iteration —
        gather its rows, whiten its targets, form its Gram, factor, solve, evaluate its losses; the classes are
        independent, and at these sizes one class neither fills the f64 matrix cores nor hides its chain of small
        factorisation kernels.  Here: rows sorted by class once (stable: a class's rows keep their order), the target
        statistics per class from contiguous segments, then ONE gather + ONE Gram GEMM + ONE X'Y GEMM + ONE batched
        Cholesky / inverse / solve chain for up to 32 classes (backend.rls_train_batched), and the per-class losses.
        Per class the arithmetic is the sequential path's; printed lines and the returned object array are the same."""
        dev = _device()
        chosen_classes = self.cfg['CHOSEN_CLASSES']
        start_index = 0 if self.is_rpn else 1
        num_clss = len(chosen_classes)
        ids = list(range(start_index, num_clss))
        F = _rows(be, self.COXY['X'])
        xdev = F.X.device
        Call = self.COXY['C'].to(xdev).reshape(-1)
        Yall = self.COXY['Y'].to(xdev)
        D, D1 = F.D, F.D + 1
        start_time = time.time()
        # (the first class batch's Gram block is cleared NOW: the 0.25-GB fill runs while the host is still busy with the sort and
        # the one read of the class sizes, instead of between the index kernel and the Grams)
        sharded_ = self.shard is not None and self.shard.enabled
        pre_zero = (be.rls_gram_zeros(F, min(len(ids), be.MAX_CLASS_BATCH)) if len(ids) and not sharded_ and hasattr(be, "rls_gram_begin")
                    and xdev.type == "cuda" and be.rls_rows_form(F) else None)
        Cl = Call.to(torch.int64)
        # rows sorted by class (stable), and the classes' runs in that order from the sorted labels: bounds[c] = the number of
        # labels below c (negative labels belong to no class, labels past the last class neither) — ONE host read; the counting
        # statements this replaces (a boolean selection, bincount, any / sum) each ended in a read of their own: 0.4 ms
        # (the sort's keys in the narrowest type that holds -1 .. num_clss: the radix sort makes one pass per key byte — eight for
        # int64 labels, 0.25 ms)
        narrow = torch.int8 if num_clss < 127 else (torch.int16 if num_clss < 32767 else torch.int64)
        sorted_labels, order = torch.sort(Cl.clamp(-1, num_clss).to(narrow), stable=True)
        bounds = torch.searchsorted(sorted_labels, torch.arange(num_clss + 1, device=xdev, dtype=narrow)).tolist()
        first = {c: bounds[c] for c in range(num_clss)}
        counts_all = [bounds[c + 1] - bounds[c] for c in range(num_clss)]
        n_loc = {c: counts_all[c] for c in ids}
        n_tot = {c: (n_loc[c] if self.shard is None else self.shard.total(n_loc[c])) for c in ids}
        models = {}
        Wall, infos, rows_of, Yw_of, whit = {}, {}, {}, {}, {}
        stat_blocks = []          # per solved group: (its classes, (G, 9, 4) f64 = [mu; T; T_inv] of each)
        solved = []
        live = [c for c in ids if n_tot[c] > 0]
        for g0 in range(0, len(live), be.MAX_CLASS_BATCH):
            group = live[g0:g0 + be.MAX_CLASS_BATCH]
            seg_off, seg_len, at = [], [], 0
            for c in group:
                seg_off.append(at)
                seg_len.append(n_loc[c])
                at += (n_loc[c] + 15) // 16 * 16
            npad = at
            sharded = self.shard is not None and self.shard.enabled
            one_launch = not sharded and hasattr(be, "rls_pad_index") and xdev.type == "cuda"
            idx_pad = None if one_launch else torch.full((max(npad, 1),), -1, dtype=torch.int64, device=xdev)[:npad]
            Yt = None
            if not sharded:
                # target statistics, whitening and the padded (index, target) arrays of the whole group in ~20 launches:
                # the rows of the group's classes are one contiguous run of `order`; they are laid out as a zero-padded
                # (class, row, 4) block, so that means, second moments and the whitening products are batched reductions /
                # matrix products (per class the same sums over the same rows as the class-by-class loop, in a different
                # order of additions: results agree to rounding)
                G_ = len(group)
                lens = torch.tensor(seg_len, dtype=torch.int64)
                nmax = max(seg_len)
                lo = first[group[0]]
                run = order[lo:lo + sum(seg_len)] if all(first[group[k + 1]] == first[group[k]] + seg_len[k] for k in range(G_ - 1)) \
                    else torch.cat([order[first[c]:first[c] + n_loc[c]] for c in group])
                total = int(sum(seg_len))
                if one_launch:
                    idx_pad, gid, pos, dest, lens_d = be.rls_pad_index(run, seg_off, seg_len, npad)
                else:
                    lens_d = lens.to(xdev)
                    gid = torch.repeat_interleave(torch.arange(G_, device=xdev), lens_d, output_size=total)      # class slot of every row
                    starts = torch.tensor(np.concatenate(([0], np.cumsum(seg_len)[:-1])), dtype=torch.int64).to(xdev)
                    pos = torch.arange(total, device=xdev) - starts[gid]                                         # row's rank inside its class
                    dest = torch.tensor(seg_off, dtype=torch.int64).to(xdev)[gid] + pos
                    idx_pad[dest] = run
                # The Grams need the rows only: queued NOW, on a side stream (odx/streams.py: a hardware queue of its own), they
                # run — 7 of the call's 12 ms at config 3 — while this stream derives the targets (statistics, the host's
                # eigen-decomposition with its synchronisation, whitening) and forms Yt [X 1] beside them.
                # With f32 targets the same sweep forms the RAW targets' products [Y 1]' X too (on the vector ALU under the Gram's
                # matrix instructions): the whitening is linear, so X' Yw follows from them and the statistics in a launch of a few
                # microseconds (rls_fold_whitened) — the 0.6-ms sweep for Yt [X 1] that used to land BEHIND the Grams (this
                # stream's small kernels starve beside them) is gone.  rls.RAW_TARGETS = False (a test hook) keeps the two-sweep form, which f64 targets always take.
                begun = side = raw5 = None
                if hasattr(be, "rls_gram_begin") and xdev.type == "cuda" and be.rls_rows_form(F):
                    from . import streams as _streams
                    own = _streams.distinct(1)
                    if own:
                        side, cur = own[0], torch.cuda.current_stream()
                        # (zeroed on THIS stream: the fold / Yt [X 1] adds to its bias row later)
                        begun = pre_zero[:G_] if pre_zero is not None and g0 == 0 else be.rls_gram_zeros(F, G_)
                        use_raw = (hasattr(be, "rls_gram_raw_begin") and Yall.dtype == torch.float32 and Yall.dim() == 2
                                   and Yall.shape[1] == 4 and RAW_TARGETS)
                        Yraw = Yall.contiguous() if use_raw else None
                        use_raw = use_raw and Yraw.data_ptr() % 16 == 0
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            if use_raw:
                                raw5 = be.rls_gram_raw_begin(F, idx_pad, seg_off, seg_len, begun, Yraw)
                                raw5.record_stream(cur)
                                Yraw.record_stream(side)
                            else:
                                be.rls_gram_begin(F, idx_pad, seg_off, seg_len, begun)
                        begun.record_stream(side)
                        idx_pad.record_stream(side)
                Ypad = torch.zeros((G_, nmax, 4), dtype=torch.float64, device=xdev)
                Ypad[gid, pos] = Yall[run].type(torch.float64)
                cnt = lens_d.type(torch.float64).clamp(min=1).view(G_, 1)
                mu_all = Ypad.sum(1) / cnt
                live_rows = (torch.arange(nmax, device=xdev).view(1, nmax) < lens_d.view(G_, 1)).unsqueeze(2)
                Yc_all = (Ypad - mu_all.unsqueeze(1)) * live_rows
                # (the 4 x 4 second moments as an outer product + one reduction: the library's batched GEMM takes 0.28 ms for
                # this (4 x n) (n x 4) shape)
                S_all = (Yc_all.unsqueeze(3) * Yc_all.unsqueeze(2)).sum(1) / cnt.view(G_, 1, 1)
            else:
                mus, Ycs, Ss = [], [], []
                for c in group:
                    I = order[first[c]:first[c] + n_loc[c]]
                    Yi = Yall[I].type(torch.float64)
                    s1 = Yi.sum(0)
                    self.shard.allreduce(s1)
                    mu = s1 / n_tot[c]
                    Yc = Yi - mu
                    S = torch.matmul(Yc.t(), Yc)
                    self.shard.allreduce(S)
                    S = S / n_tot[c]
                    mus.append(mu), Ycs.append(Yc), Ss.append(S)
                S_all = torch.stack(Ss)
            # (on the host: 4 x 4 matrices — the GPU solver costs ~8 ms for the batch, mostly launch + synchronisation)
            Ts, Tis = whitening_transforms(S_all)
            if not sharded:
                # (rows of the group, 4), class-sorted: Yw[g, r, j] = sum_i Yc[g, r, i] T[g, i, j] as a product + a reduction
                Yw_all = (Yc_all.unsqueeze(3) * Ts.unsqueeze(1)).sum(2)[gid, pos]
                if raw5 is None:
                    Yt = torch.zeros((4, max(npad, 16)), dtype=torch.float64, device=xdev)
                    Yt[:, dest] = Yw_all.t()
                a = 0
                for k, c in enumerate(group):
                    rows_of[c] = run[a:a + n_loc[c]]
                    whit[c], Yw_of[c] = (mu_all[k], Ts[k], Tis[k]), Yw_all[a:a + n_loc[c]]
                    a += n_loc[c]
            else:
                Yt = torch.zeros((4, max(npad, 16)), dtype=torch.float64, device=xdev)
                for k, (c, off) in enumerate(zip(group, seg_off)):
                    rows_of[c] = order[first[c]:first[c] + n_loc[c]]
                    Yw = torch.matmul(Ycs[k], Ts[k])
                    whit[c], Yw_of[c] = (mus[k], Ts[k], Tis[k]), Yw
                    idx_pad[off:off + n_loc[c]] = rows_of[c]
                    Yt[:, off:off + n_loc[c]] = Yw.t()
            stats_blk = None
            if not sharded and raw5 is not None:
                stats_blk = torch.cat((mu_all.view(G_, 1, 4), Ts, Tis), dim=1).contiguous()
                W, info = be.rls_train_batched(F, idx_pad, seg_off, seg_len, None, self.lambd, begun=begun, allreduce=None, after=side,
                                               raw=(raw5, stats_blk, lens_d.type(torch.float64)))
            elif not sharded and begun is not None:
                # Yt [X 1] (an HBM sweep) beside the Grams' last tiles, then the solves behind both
                W, info = be.rls_train_batched(F, idx_pad, seg_off, seg_len, Yt, self.lambd, begun=begun,
                                               allreduce=None, after=side)
            else:
                W, info = be.rls_train_batched(F, idx_pad, seg_off, seg_len, Yt, self.lambd,
                                               allreduce=self.shard.allreduce if self.shard is not None else None)
            stat_blocks.append((group, stats_blk if stats_blk is not None else
                                torch.cat((torch.stack([whit[c][0] for c in group]).view(len(group), 1, 4) if sharded
                                           else mu_all.view(len(group), 1, 4), Ts, Tis), dim=1)))
            # (the factorisations' status words are read at the END, after the predictions and the loss statements have been queued: a read here stalls
            # the host until the whole solve chain has run, and the predictions then start 0.2 ms of host work late)
            for k, c in enumerate(group):
                Wall[c] = W[k]
            solved.append((group, W, info))
        # the training losses of all classes: one prediction launch per class into ONE (rows, 4) array, the elementwise
        # part once over all of it (per class it was ~15 short launches: 5 ms of a 30-ms call); every class keeps its own
        # contiguous copies, as the class-by-class loop hands them out (a saved model must not drag the whole array along)
        entries, means = {}, []
        at, span = 0, {}
        for i in live:
            span[i] = (at, at + n_loc[i])
            at += n_loc[i]
        P_all = torch.empty((at, 4), dtype=torch.float64, device=xdev)
        if hasattr(be, "rls_predict_rows_batched"):
            for group, W, _ in solved:        # one launch per solved group of classes (its weights are one (C, 4, ld) block)
                rows = torch.cat([rows_of[c] for c in group]) if len(group) > 1 else rows_of[group[0]].contiguous()
                a0 = span[group[0]][0]
                be.rls_predict_rows_batched(F, rows, [span[c][0] - a0 for c in group], W, P_all[a0:span[group[-1]][1]])
        else:
            for i in live:
                be.rls_predict_rows(F, rows_of[i].contiguous(), Wall[i], out=P_all[span[i][0]:span[i][1]])
        losses_all = (0.5 * (P_all - torch.cat([Yw_of[i] for i in live])) ** 2).type(torch.float32) if live else None
        W32 = torch.stack([Wall[i][:, :D1] for i in live]).to(dev).type(torch.float32) if live else None
        stats32 = torch.cat([b for _, b in stat_blocks]).to(dev).type(torch.float32) if live else None      # (live classes in order)
        # Three tensors of its own per class — its losses, its four weight rows, its nine statistics rows — and views into
        # them: a class's entry keeps 16 KB + its own losses alive, not the arrays of all classes.  All 3 x classes copies in
        # ONE multi-tensor launch (was a launch per copy: 90 at config 3, the tail of the call).
        srcs = [t for j, i in enumerate(live) for t in (losses_all[span[i][0]:span[i][1]], W32[j], stats32[j])]
        own = [torch.empty_like(t) for t in srcs]
        if srcs:
            fc = getattr(torch, "_foreach_copy_", None)       # one multi-tensor copy where this torch has it (a private entry point)
            if fc is not None:
                fc(own, srcs)
            else:
                for d_, s_ in zip(own, srcs):
                    d_.copy_(s_)
        for j, i in enumerate(live):
            losses, Wc, sc = own[3 * j], own[3 * j + 1], own[3 * j + 2]
            Beta = {str(k): {'weights': Wc[k], 'losses': losses[:, k]} for k in range(4)}
            entries[i] = {'mu': sc[0], 'T': sc[1:5], 'T_inv': sc[5:9], 'Beta': Beta}
        if live:
            # the printed per-class means from ONE padded (class, row, 4) block and one reduction, instead of a reduction per
            # class (a running sum over all rows is no substitute: torch's scan of 3e5 x 4 doubles takes 0.7 ms here)
            lens_h = [n_loc[i] for i in live]
            if len(solved) == 1 and not sharded_ and gid.numel() == at:
                lens, slot, rank = lens_d, gid, pos           # (one class batch: its index maps are these)
            else:
                lens = torch.tensor(lens_h, dtype=torch.int64, device=losses_all.device)
                slot = torch.repeat_interleave(torch.arange(len(live), device=lens.device), lens, output_size=at)
                first_row = torch.tensor(np.concatenate(([0], np.cumsum(lens_h)[:-1])), dtype=torch.int64).to(lens.device)
                rank = torch.arange(at, device=lens.device) - first_row[slot]
            padded = torch.zeros((len(live), max(max(lens_h), 1), 4), dtype=torch.float32, device=lens.device)
            padded[slot, rank] = losses_all
            means = (padded.sum(1, dtype=torch.float64) / lens.view(-1, 1)).type(torch.float32)     # 0 / 0 = nan: no local rows
        for group, _, info in solved:
            bad = info.tolist()
            for k, c in enumerate(group):
                if bad[k] != 0:
                    raise RuntimeError('RLS Cholesky failed for class %s (pivot %d)' % (chosen_classes[c], bad[k] - 1))
        mean_host = dict(zip(live, means.tolist())) if live else {}      # one host read for the printed lines
        out = np.empty((0))
        for i in ids:
            print('Training regressor for class %s (%d/%d)' % (chosen_classes[i], i, num_clss - 1))
            print('Training with %i examples' % n_loc[i])
            if n_tot[i] == 0:
                out = np.append(out, {'mu': None, 'T': None, 'T_inv': None, 'Beta': None})
                print('No indices for class %s' % (chosen_classes[i]))
                continue
            out = np.append(out, entries[i])
            print('Mean losses:', mean_host[i] if n_loc[i] else None)
        self._report_time(time.time() - start_time, num_clss, output_dir)
        return out

    def _report_time(self, training_time, num_clss, output_dir):
        print('Time required to train %d regressors: %f seconds.' % (num_clss - 1, training_time))
        if output_dir:
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                if self.is_rpn:
                    fid.write("RPN's Online Region Refiner training time: {}min:{}s \n".format(
                        int(training_time / 60), round(training_time % 60)))
                else:
                    fid.write("Detector's Online Region Refiner training time: {}min:{}s \n \n".format(
                        int(training_time / 60), round(training_time % 60)))

    def _train_sequential(self, be, output_dir=None):
        dev = _device()
        chosen_classes = self.cfg['CHOSEN_CLASSES']
        start_index = 0 if self.is_rpn else 1
        num_clss = len(chosen_classes)
        models = np.empty((0))
        F = _rows(be, self.COXY['X'])
        Call = self.COXY['C'].to(F.X.device)
        Yall = self.COXY['Y'].to(F.X.device)
        D = F.D
        D1 = D + 1
        ldg = (D1 + 1) // 2 * 2
        start_time = time.time()
        for i in range(start_index, num_clss):
            print('Training regressor for class %s (%d/%d)' % (chosen_classes[i], i, num_clss - 1))
            I = torch.where(Call == i)[0]
            print('Training with %i examples' % len(I))
            n_loc = len(I)
            n_tot = n_loc if self.shard is None else self.shard.total(n_loc)
            if n_tot == 0:
                models = np.append(models, {'mu': None, 'T': None, 'T_inv': None, 'Beta': None})
                print('No indices for class %s' % (chosen_classes[i]))
                continue
            Yi = Yall[I].type(torch.float64)
            if self.shard is not None and self.shard.enabled:
                mu, Yc, T, T_inv = self._whiten_sharded(Yi, n_tot)
            else:
                mu, Yc, T, T_inv = whiten_targets(Yi)
            Yw = torch.matmul(Yc, T)                                     # (n, 4) whitened targets
            Yt = torch.zeros((4, (n_loc + 15) // 16 * 16 + 16), dtype=torch.float64, device=Yw.device)
            Yt[:, :n_loc] = Yw.t()
            G = torch.zeros((D1, ldg), dtype=torch.float64, device=Yw.device)
            XtY = torch.zeros((4, ldg), dtype=torch.float64, device=Yw.device)
            be.rls_gram(F, I.contiguous(), Yt, G, XtY)
            if self.shard is not None:
                self.shard.allreduce(G)
                self.shard.allreduce(XtY)
            W, info = be.rls_solve(G, D, self.lambd, XtY)               # (4, D+1) f64
            if int(info.item()) != 0:
                raise RuntimeError('RLS Cholesky failed for class %s (pivot %d)' % (chosen_classes[i], int(info.item()) - 1))
            P = be.rls_predict_rows(F, I.contiguous(), W)                # (n, 4) = [X 1] w
            losses = 0.5 * (P - Yw) ** 2
            Beta = {}
            for k in range(4):
                Beta[str(k)] = {'weights': W[k, :D1].to(dev).type(torch.float32),
                                'losses': losses[:, k].type(torch.float32)}
            models = np.append(models, {'mu': mu.to(dev).type(torch.float32), 'T': T.to(dev).type(torch.float32),
                                        'T_inv': T_inv.to(dev).type(torch.float32), 'Beta': Beta})
            mean_losses = torch.stack([Beta[str(k)]['losses'].mean() for k in range(4)]) if n_loc else None
            print('Mean losses:', mean_losses)
        self._report_time(time.time() - start_time, num_clss, output_dir)
        return models

    def solve(self, X, y, lmbd, X_test=None, Y_test=None, indices=None):
        """The reference's per-class solver as a method of its own (train_region_refiner.py:100-119): X (n, D + 1) with
        the bias column last, y (n, 4) whitened targets; per coordinate k the ridge solution of (X'X + lmbd I) w = X'y_k
        (over the rows indices[k] when given) and the losses 0.5 (X w - y_k)^2.  Runs on the same kernels as `train`
        (Gram, Cholesky, triangular inverses, predictions) when X is f32 features with a column of ones appended —
        what `train` builds; any other matrix goes through a dense f64 solve."""
        be = _backend.get_backend()
        dev = _device()
        X, y = torch.as_tensor(X), torch.as_tensor(y)
        n, D1 = X.shape
        feats = X[:, :-1]
        native = (hasattr(be, "rls_gram") and D1 >= 2 and bool((X[:, -1] == 1).all())
                  and bool((feats.to(torch.float32).to(X.dtype) == feats).all()))
        out = {}
        if native:
            F = _rows(be, feats.to(torch.float32))
            xdev = F.X.device
            W_all = None
            for k in range(4):
                if indices is not None or W_all is None:
                    I = (torch.arange(n, device=xdev) if indices is None else torch.as_tensor(indices[k]).to(xdev)).to(torch.int64)
                    if I.dtype == torch.bool:
                        I = I.nonzero().reshape(-1)
                    nc = I.numel()
                    Yt = torch.zeros((4, (nc + 15) // 16 * 16 + 16), dtype=torch.float64, device=xdev)
                    Yt[:, :nc] = y.to(xdev, torch.float64)[I].t()
                    ldg = (D1 + 1) // 2 * 2
                    G = torch.zeros((D1, ldg), dtype=torch.float64, device=xdev)
                    XtY = torch.zeros((4, ldg), dtype=torch.float64, device=xdev)
                    be.rls_gram(F, I.contiguous(), Yt, G, XtY)
                    W_all, info = be.rls_solve(G, D1 - 1, lmbd, XtY)
                    if int(info.item()) != 0:
                        raise RuntimeError('RLS Cholesky failed (pivot %d)' % (int(info.item()) - 1))
                    P = be.rls_predict_rows(F, I.contiguous(), W_all)
                    Ysub = y.to(xdev, torch.float64)[I]
                out[str(k)] = {'weights': W_all[k, :D1].to(dev).type(torch.float32),
                               'losses': (0.5 * (P[:, k] - Ysub[:, k]) ** 2).type(torch.float32)}
            return out
        # any other matrix (rows that are not f32-exact, no column of ones): the same normal equations through the library's
        # f64 kernels — Gram and X'y on the f64 matrix cores (odx_gemm_nt_f64), Cholesky + the four solves (odx_rls_solve_f64),
        # the predictions by one more product.  No vendor BLAS / solver on this path either.
        if not hasattr(be, "gemm_nt_f64"):
            raise RuntimeError("RegionRefinerTrainer.solve: the backend offers no f64 solve for a general design matrix")
        xdev = be.device
        X64, y64 = X.to(xdev, torch.float64), y.to(xdev, torch.float64)
        W, P, Ysub = None, None, None
        for k in range(4):
            if indices is not None or W is None:
                if indices is None:
                    Xs, Ysub = X64, y64
                else:
                    I = torch.as_tensor(indices[k]).to(xdev)
                    Xs, Ysub = X64[I], y64[I]
                Xt = Xs.t().contiguous()                                              # (D1, n_k)
                G = be.gemm_nt_f64(Xt, Xt)                                            # X'X  (D1, D1)
                XtY = be.gemm_nt_f64(Ysub.t().contiguous(), Xt)                       # (4, D1)
                W, info = be.rls_solve(G, D1 - 1, lmbd, XtY)                          # (views of even-stride blocks)
                if int(info.item()) != 0:
                    raise RuntimeError('RLS Cholesky failed (pivot %d)' % (int(info.item()) - 1))
                P = be.gemm_nt_f64(Xs, W[:, :D1])                                     # (n_k, 4)
            out[str(k)] = {'weights': W[k, :D1].to(dev).type(torch.float32),
                           'losses': (0.5 * (P[:, k] - Ysub[:, k]) ** 2).type(torch.float32)}
        return out

    def _whiten_sharded(self, Yi, n_tot):
        """Target statistics over all row shards: sum and second moment all-reduced (4 + 16 numbers)."""
        s1 = Yi.sum(0)
        self.shard.allreduce(s1)
        mu = s1 / n_tot
        Yc = Yi - mu
        S = torch.matmul(Yc.t(), Yc)
        self.shard.allreduce(S)
        S = S / n_tot
        evals, W = sym_eig_small(S)
        root = torch.sqrt(evals + 0.001)
        return mu, Yc, W @ torch.diag(1.0 / root) @ W.t(), W @ torch.diag(root) @ W.t()


class RegionPredictor:
    def __init__(self, cfg, models):
        self.cfg = cfg
        self.models = models

    def __call__(self, boxes, features, normalize_features=False, stats=None):
        # the reference drops its normalisation arguments here (predict_regions.py:13)
        return self.predict(boxes, features, normalize_features=False, stats=None)

    def predict(self, boxes, features, normalize_features=False, stats=None):
        be = _backend.get_backend()
        dev = _device()
        num_clss = len(self.cfg['CHOSEN_CLASSES'])
        img_width, img_height = boxes[0].size[0], boxes[0].size[1]
        for i in range(len(boxes)):
            keep = np.nonzero(features[i]['gt'] == 0)          # ground-truth rows are excluded
            feat = torch.tensor(features[i]['feat'][keep, :][0], device=dev)
            if normalize_features:
                feat = (feat - stats['mean']) * (20 / stats['mean_norm'].item())
            F = _rows(be, feat)
            ex_box = boxes[i].bbox.to(dev)
            num_boxes = ex_box.size()[0]
            if hasattr(be, "gemm_nt") and all(m['Beta'] is not None for m in self.models[:num_clss - 1]):
                # all classes at once (what the in-network head does, odx.heads): the regressors with T_inv and mu folded
                # in as ONE (4 (C - 1), D) weight matrix, one f32 GEMM per image, one decode over (boxes, classes) —
                # instead of ~30 launches per class and image
                fold = getattr(self, "_fold", None)
                if fold is None or fold[0] is not self.models or fold[1] != F.D:
                    from .heads import _fold_regressors
                    Wt, bias = _fold_regressors(list(self.models[:num_clss - 1]), F.D, False)
                    fold = self._fold = (self.models, F.D, _rows(be, Wt.to(F.X.device)), bias.to(F.X.device))
                Y = (be.gemm_nt(F, fold[2]) + fold[3]).view(num_boxes, num_clss - 1, 4)
                dec = decode_boxes(ex_box.unsqueeze(1), Y, img_width, img_height, plus=float(np.spacing(1)))
                boxes[i].bbox = torch.cat((ex_box.view(num_boxes, 1, 4), dec), dim=1)
                continue
            out = [ex_box]
            for j in range(1, num_clss):
                m = self.models[j - 1]
                W = torch.stack([m['Beta'][str(k)]['weights'] for k in range(4)]).to(F.X.device, torch.float64)
                Y = be.rls_predict_rows(F, None, W.contiguous()).to(torch.float32)     # F w + b
                Y = torch.matmul(Y, m['T_inv'].to(Y.device)) + m['mu'].to(Y.device)
                out.append(decode_boxes(ex_box, Y, img_width, img_height, plus=float(np.spacing(1))))
            boxes[i].bbox = torch.cat(out, dim=1).view((num_boxes, num_clss, 4))
        return boxes


def decode_boxes(ex_box, Y, img_width, img_height, plus):
    """Box decoding of predict_regions.py:50-70: widths are x2 - x1 + `plus` (np.spacing(1) there,
    1 in py_od_utils.decode_boxes_detector), the far corner is centre + w/2 - 1, clamped to the image."""
    # (ex_box (R, 4) with Y (R, 4), or ex_box (R, 1, 4) with Y (R, K, 4) for all classes at once: the last axis is the box)
    src_w = ex_box[..., 2] - ex_box[..., 0] + plus
    src_h = ex_box[..., 3] - ex_box[..., 1] + plus
    ctr_x = ex_box[..., 0] + 0.5 * src_w
    ctr_y = ex_box[..., 1] + 0.5 * src_h
    pred_ctr_x = Y[..., 0] * src_w + ctr_x
    pred_ctr_y = Y[..., 1] * src_h + ctr_y
    pred_w = torch.exp(Y[..., 2]) * src_w
    pred_h = torch.exp(Y[..., 3]) * src_h
    x1 = torch.clamp(pred_ctr_x - 0.5 * pred_w, min=0)
    y1 = torch.clamp(pred_ctr_y - 0.5 * pred_h, min=0)
    x2 = torch.clamp(pred_ctr_x + 0.5 * pred_w - 1, max=img_width - 1)
    y2 = torch.clamp(pred_ctr_y + 0.5 * pred_h - 1, max=img_height - 1)
    return torch.stack([x1, y1, x2, y2], dim=-1)
