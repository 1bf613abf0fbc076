"""Test-time on-line heads (A8 / A9): multi-class FALKON scoring and RLS box refinement for a
whole image in two kernel launches.

Behaviour follows the FALKON / RLS branches of the reference's heads:
  detector  FastRCNNPredictor ("OnlineDetectionBOXPredictor"), forward :32-75, refine_boxes[_parallel]
            :77-124, predict_clss_FALKON[_parallel] :126-160
            (mrcnn_modified/modeling/roi_heads/box_head/roi_box_predictors.py)
  RPN       RPNHead ("OnlineRPNHead"), forward :106-135, refine_boxes[_parallel] :137-187,
            compute_objectness_FALKON[_parallel] :189-227 (mrcnn_modified/modeling/rpn/rpn.py)
including their conventions: background column / missing classifier score -2 (detector
sequential path and RPN) but 0 for a missing classifier on the detector's parallel path, zero
deltas for a missing regressor and for the detector's background slot, feature normalisation
(x - mean) * 20 / mean_norm before scoring, regressors fed raw or normalised features per
`normalize_features_regressors`.

MI355X form: all classifiers are scored by ONE fused Gaussian-mmv launch over the concatenated
centres with per-class row ranges (odx_gauss_mmv_f32), and all regressors are applied by ONE
f32 MFMA GEMM whose weights already contain T_inv and mu (folded once, in f64, when the models
are set), instead of per-class Python loops or a GEMM + block-diagonal GEMM pair.
Caches are rebuilt whenever models are (re)assigned — the demo's update_model contract
(mrcnn_modified/demo/predictor_online_segmentation.py:404-425).
"""
import torch

from . import backend as _backend


def _fold_regressors(regressors, D, background):
    """(4K, D) f32 weight rows and (4K,) bias with T_inv and mu folded in; K = len + background."""
    K = len(regressors) + (1 if background else 0)
    Wt = torch.zeros((4 * K, D), dtype=torch.float64)
    bias = torch.zeros(4 * K, dtype=torch.float64)
    for j, m in enumerate(regressors):
        if m['Beta'] is None:
            continue
        W = torch.stack([m['Beta'][str(k)]['weights'].detach().cpu().double() for k in range(4)], dim=1)   # (D+1, 4)
        Ti = m['T_inv'].detach().cpu().double()
        Wf = W @ Ti                                            # (x W[:-1] + W[-1]) T_inv + mu
        o = 4 * (j + (1 if background else 0))
        Wt[o:o + 4] = Wf[:-1].t()
        bias[o:o + 4] = Wf[-1] + m['mu'].detach().cpu().double()
    return Wt.float(), bias.float()


class _OnlineHead:
    background = True          # detector: a background slot in front of the classes
    missing_parallel = 0.0     # score of a missing classifier on the batched path
    missing_sequential = -2.0

    def __init__(self, classifiers=None, regressors=None, stats=None, parallel_inference=True):
        self.parallel_inference = parallel_inference
        self.set_models(classifiers, regressors, stats)

    def set_models(self, classifiers=None, regressors=None, stats=None):
        self.classifiers = classifiers
        self.regressors = regressors
        self.stats = stats
        self._cls_cache = None
        self._reg_cache = None

    # -- scoring
    def _scores(self, F, n):
        be = _backend.get_backend()
        C = len(self.classifiers)
        live = [m for m in self.classifiers if m]
        batched = self.parallel_inference and C > 1
        fill = self.missing_parallel if batched else self.missing_sequential
        if not live:
            return torch.full((n, C), fill, dtype=torch.float32, device=F.X.device)
        if self._cls_cache is None:
            dev = F.X.device
            ny = torch.cat([m.ny_points_.to(dev, torch.float32) for m in live])
            total = ny.shape[0]
            V = torch.zeros((total, C), dtype=torch.float64, device=dev)
            ranges = torch.zeros((C, 2), dtype=torch.int32)
            row = 0
            for i, m in enumerate(self.classifiers):
                if m:
                    V[row:row + m.M, i] = m.alpha_.to(dev, torch.float64).reshape(-1)
                    ranges[i, 0], ranges[i, 1] = row, row + m.M
                    row += m.M
            longest = int((ranges[:, 1] - ranges[:, 0]).max()) if C else 0
            self._cls_cache = (be.features(ny), V, ranges.to(dev), live[0].kernel.sigma, longest)
        Zf, V, ranges, sigma, longest = self._cls_cache
        s = be.mmv(F, Zf, sigma, V, ranges, max_range=longest)
        for i, m in enumerate(self.classifiers):
            if not m:
                s[:, i] = fill
        return s

    # -- refinement
    def _deltas(self, F, n):
        be = _backend.get_backend()
        if self._reg_cache is None:
            Wt, bias = _fold_regressors(self.regressors, F.D, self.background)
            self._reg_cache = (be.features(Wt), bias.to(F.X.device))
        Wf, bias = self._reg_cache
        return be.gemm_nt(F, Wf) + bias


class OnlineBoxPredictor(_OnlineHead):
    """cls_scores (R, C+1), bbox_pred (R, 4 (C+1)) = head(x) for pooled RoI features x (R, D)
    (or (R, D, h, w), average-pooled like the reference's avgpool)."""
    background = True
    missing_parallel = 0.0

    def __init__(self, classifiers=None, regressors=None, stats=None, parallel_inference=True,
                 normalize_features_regressors=False):
        super().__init__(classifiers, regressors, stats, parallel_inference)
        self.normalize_features_regressors = normalize_features_regressors

    def __call__(self, x):
        be = _backend.get_backend()
        if x.dim() == 4:
            x = x.mean(dim=(2, 3))
        x = x.reshape(x.size(0), -1)
        n = x.shape[0]
        F = be.features(x)
        bbox = None
        if not self.normalize_features_regressors:
            bbox = self._deltas(F, n)
        if self.stats:
            x = (x - self.stats['mean'].to(x.device)) * (20 / self.stats['mean_norm'].to(x.device))
            F = be.features(x)
        if self.normalize_features_regressors:
            bbox = self._deltas(F, n)
        s = self._scores(F, n)
        scores = torch.cat((torch.full((n, 1), -2.0, dtype=torch.float32, device=s.device), s), dim=1)
        return scores, bbox

    forward = __call__


class OnlineRPNHead(_OnlineHead):
    """logits (1, A, H, W), bbox_reg (1, 4 A, H, W) from the RPN activation t = relu(conv3x3(C4))
    of shape (1, D, H, W); one classifier and one regressor per anchor type."""
    background = False
    missing_parallel = -2.0

    def __call__(self, t):
        be = _backend.get_backend()
        _, D, H, W = t.shape
        x = t.permute(0, 2, 3, 1).reshape(H * W, D)
        x = (x - self.stats['mean'].to(x.device)) * (20 / self.stats['mean_norm'].to(x.device))
        F = be.features(x)
        A = len(self.classifiers)
        s = self._scores(F, H * W)
        logits = s.t().reshape(1, A, H, W)
        bbox = self._deltas(F, H * W).t().reshape(1, 4 * len(self.regressors), H, W)
        return logits, bbox

    forward = __call__


class OnlineMaskPredictor(_OnlineHead):
    """Per-pixel FALKON scores (R, C+1, S, S) from the mask activation m = relu(conv5_mask(x)) of
    shape (R, D, S, S): MaskRCNNC4Predictor.forward's on-line branch, predict_pixel_FALKON[_parallel]
    (mrcnn_modified/modeling/roi_heads/mask_head/roi_mask_predictors.py:37-99).  Every pixel is a
    row; background channel -2; the reference's transpose / index gymnastics (:66-70,95-99) amount to
    (R, S*S, C+1) -> (R, C+1, S, S)."""
    background = True
    missing_parallel = 0.0

    def __init__(self, classifiers=None, stats=None, parallel_inference=True):
        super().__init__(classifiers, None, stats, parallel_inference)

    def __call__(self, m):
        be = _backend.get_backend()
        R, D, S, _ = m.shape
        x = m.permute(0, 2, 3, 1).reshape(-1, D)
        x = (x - self.stats['mean'].to(x.device)) * (20 / self.stats['mean_norm'].to(x.device))
        s = self._scores(be.features(x), x.shape[0])
        C = len(self.classifiers)
        s = torch.cat((torch.full((x.shape[0], 1), -2.0, dtype=torch.float32, device=s.device), s), dim=1)
        return s.view(R, S * S, C + 1).permute(0, 2, 1).reshape(R, C + 1, S, S)

    forward = __call__
