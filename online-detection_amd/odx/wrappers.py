"""FALKONWrapper: the reference's classifier plug-in for OnlineRegionClassifier, rebuilt on the
odx estimators.  Behaviour follows
src/modules/region-classifier/FALKONWrapper_with_centers_selection.py:16-95 (CPU / out-of-core)
and ..._incore.py:16-99 (GPU resident): YAML keys, defaults with a printed notice, the Nystroem
index rule (<= M/2 positives first, then negatives, sampled with replacement through the global
torch RNG), M = len(indices), a deep copy of the fitted model as the return value."""
import copy
import sys

import torch
import yaml

from . import falkon as _falkon


class CenterSelector:
    """MyCenterSelector (MyCenterSelector.py:3-15): hand the estimator pre-chosen rows."""

    def __init__(self, center_indices):
        self.center_indices = center_indices

    def select(self, X, Y):
        picked = X[self.center_indices, :]
        if picked.dim() > 2:
            picked = picked.squeeze()
        if Y is None:
            return picked
        return picked, Y[self.center_indices, :]


class _BatchRound:
    """See FALKONWrapperBase.train_batch_begin."""

    def __init__(self, wrapper, sigma, lam, streams, expect_total):
        self.w, self.sigma, self.lam, self.expect_total = wrapper, sigma, lam, expect_total
        self.fit = _falkon.BatchFit(streams=streams)
        self.count = 0
        w = wrapper
        w.kernel = w.kernel_cls(sigma=sigma)
        own_rule = (type(w).compute_indices_selection is FALKONWrapperBase.compute_indices_selection
                    and "compute_indices_selection" not in w.__dict__)
        self.rule = w._indices_tensor if own_rule else w.compute_indices_selection

    def add(self, Xs, ys, index_rng=None):
        models = []
        for k, y in enumerate(ys):
            pos = self.count + k
            indices = index_rng(pos, lambda y=y: self.rule(y)) if index_rng else self.rule(y)
            models.append(self.w._estimator_for(indices, self.sigma, self.lam))
        self.count += len(models)
        self.fit.add(models, Xs, ys, expect_total=self.expect_total, centres_cap=getattr(self.w, "nyst_centers", None))

    def finish(self):
        models = self.fit.finish()
        self.w.model = models[-1] if models else None
        return models

    def abort(self):
        """Drop the round (odx.falkon.BatchFit.abort): what was queued for the groups added so far is joined, nothing is fitted."""
        self.fit.abort()


class FALKONWrapperBase:
    incore = True

    def __init__(self, cfg_path=None, is_rpn=False, is_segmentation=False):
        if cfg_path is not None:
            with open(cfg_path) as fid:
                self.cfg = yaml.load(fid, Loader=yaml.FullLoader)
            if is_rpn:
                self.cfg = self.cfg['RPN']
        section = 'ONLINE_SEGMENTATION' if is_segmentation else 'ONLINE_REGION_CLASSIFIER'
        opts = self.cfg[section]['CLASSIFIER']
        if 'sigma' in opts:
            self.sigma = opts['sigma']
        else:
            print('Sigma not given for creating Falkon, default value is used.')
            self.sigma = 5
        if 'lambda' in opts:
            self.lam = opts['lambda']
        else:
            print('Lambda not given for creating Falkon, default value is used.')
            self.lam = 0.001
        self.kernel = None
        self.nyst_centers = opts['M']
        if self.incore:
            self.maxiter = 20  # falkon's default number of CG iterations

    # the estimator classes are attributes so tests can swap in recorders
    estimator_incore = _falkon.InCoreFalkon
    estimator_cpu = _falkon.Falkon
    kernel_cls = _falkon.GaussianKernel
    options_cls = _falkon.FalkonOptions
    selector_cls = CenterSelector

    def train(self, X, y, sigma=None, lam=None):
        sigma = self.sigma if sigma is None else sigma
        lam = self.lam if lam is None else lam
        self.kernel = self.kernel_cls(sigma=sigma)
        indices = self.compute_indices_selection(y)
        if isinstance(indices, list):
            # handed to the selector as one tensor: same rows, but the returned deep copy of the model (and any pickle of
            # it) no longer walks a python list of M integers (~1.4 ms per fit at M = 2000)
            indices = torch.as_tensor(indices, dtype=torch.int64)
        if self.incore:
            if isinstance(indices, int):
                indices = [indices]
            opt = self.options_cls(min_cuda_iter_size_32=0, min_cuda_iter_size_64=0, keops_active="no",
                                   min_cuda_pc_size_32=0, min_cuda_pc_size_64=0, store_kernel_d_threshold=250)
            self.model = self.estimator_incore(kernel=self.kernel, penalty=lam, M=len(indices), maxiter=self.maxiter,
                                               center_selection=self.selector_cls(indices), options=opt)
        else:
            opt = self.options_cls(min_cuda_iter_size_32=0, min_cuda_iter_size_64=0, keops_active="no")
            self.model = self.estimator_cpu(kernel=self.kernel, penalty=lam, M=len(indices),
                                            center_selection=self.selector_cls(indices), options=opt)
        if self.model is None:
            print('Model is None in trainRegionClassifier function')
            sys.exit(0)
        self.model.fit(X, y)
        return copy.deepcopy(self.model)

    def train_batch(self, Xs, ys, sigma=None, lam=None, index_rng=None, streams=None):
        """`train` for several independent classes at once (one Minibootstrap round): the same index rule, estimator
        construction and deep-copied result per class, with the fits sharing one batched preconditioner launch chain
        (odx.falkon.fit_batch).  index_rng: optional callable (i, fn) that runs fn under class i's own RNG state — the
        reference draws the Nystroem indices from the global RNG class by class; here the classes of a round are
        selected back to back, so callers that need reproducible draws give every class its own stream."""
        sigma = self.sigma if sigma is None else sigma
        lam = self.lam if lam is None else lam
        self.kernel = self.kernel_cls(sigma=sigma)
        models = []
        # the index rule as shipped keeps its result as a tensor (same draws, same order; no 2000-element python list per
        # class and round); a subclass or an instance that replaces compute_indices_selection is called as is
        own_rule = (type(self).compute_indices_selection is FALKONWrapperBase.compute_indices_selection
                    and "compute_indices_selection" not in self.__dict__)
        rule = self._indices_tensor if own_rule else self.compute_indices_selection
        for i, (X, y) in enumerate(zip(Xs, ys)):
            indices = index_rng(i, lambda: rule(y)) if index_rng else rule(y)
            if isinstance(indices, int):
                indices = [indices]
            opt = self.options_cls(min_cuda_iter_size_32=0, min_cuda_iter_size_64=0, keops_active="no",
                                   min_cuda_pc_size_32=0, min_cuda_pc_size_64=0, store_kernel_d_threshold=250)
            cls = self.estimator_incore if self.incore else self.estimator_cpu
            kw = {"maxiter": self.maxiter} if self.incore else {}
            # the indices as one tensor (same rows selected; a model with a 2000-entry python list costs ~1 ms to copy or pickle)
            models.append(cls(kernel=self.kernel_cls(sigma=sigma), penalty=lam, M=len(indices),
                              center_selection=self.selector_cls(torch.as_tensor(indices, dtype=torch.int64)), options=opt, **kw))
        _falkon.fit_batch(models, Xs, ys, streams=streams)
        self.model = models[-1] if models else None
        return models       # fresh objects per call: nothing of the wrapper aliases them, so no deep copy is needed

    def train_batch_begin(self, sigma=None, lam=None, streams=None, expect_total=None):
        """train_batch in two phases (odx.falkon.BatchFit): the returned object takes the classes of a round in groups —
        ``add(Xs, ys, index_rng)`` applies the index rule, builds the estimators and queues the group's preconditioner chain
        at once; ``finish()`` fits everything and returns the estimators in the order they were added.  index_rng's first
        argument is the class's position among ALL classes added so far."""
        return _BatchRound(self, self.sigma if sigma is None else sigma, self.lam if lam is None else lam, streams, expect_total)

    def _estimator_for(self, indices, sigma, lam):
        if isinstance(indices, int):
            indices = [indices]
        opt = self.options_cls(min_cuda_iter_size_32=0, min_cuda_iter_size_64=0, keops_active="no",
                               min_cuda_pc_size_32=0, min_cuda_pc_size_64=0, store_kernel_d_threshold=250)
        cls = self.estimator_incore if self.incore else self.estimator_cpu
        kw = {"maxiter": self.maxiter} if self.incore else {}
        # the indices as one tensor (same rows selected; a model with a 2000-entry python list costs ~1 ms to copy or pickle)
        return cls(kernel=self.kernel_cls(sigma=sigma), penalty=lam, M=len(indices),
                   center_selection=self.selector_cls(torch.as_tensor(indices, dtype=torch.int64)), options=opt, **kw)

    def model_from_tensors(self, ny_points, alpha, sigma=None, lam=None):
        """An estimator as `train` returns it, around centres and coefficients that were trained elsewhere (another rank
        of a class-sharded Minibootstrap, a model file): same class, kernel, penalty and M = number of centres."""
        sigma = self.sigma if sigma is None else sigma
        lam = self.lam if lam is None else lam
        cls = self.estimator_incore if self.incore else self.estimator_cpu
        kw = {"maxiter": self.maxiter} if self.incore else {}
        m = cls(kernel=self.kernel_cls(sigma=sigma), penalty=lam, M=int(ny_points.shape[0]), center_selection=None,
                options=self.options_cls(keops_active="no"), **kw)
        m.ny_points_, m.alpha_ = ny_points, alpha.reshape(-1, 1)
        return m

    def predict(self, model, X_np, y=None):
        if y is not None:
            return model.predict(X_np, y)
        return model.predict(X_np)

    def test(self):
        pass

    def _indices_tensor(self, y):
        """compute_indices_selection up to its final `.squeeze().tolist()`: (k,) int64 tensor on y's device."""
        half = int(self.nyst_centers / 2)
        pos = (y == 1).nonzero()
        if pos.size()[0] > half:
            pos = pos[torch.randint(pos.size()[0], (half,))]
        neg = (y == -1).nonzero()
        room = self.nyst_centers - pos.size()[0]
        if neg.size()[0] > room:
            neg = neg[torch.randint(neg.size()[0], (room,))]
        return torch.cat((pos, neg), dim=0).reshape(-1)

    def compute_indices_selection(self, y):
        return self._indices_tensor(y).squeeze().tolist()
