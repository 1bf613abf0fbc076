"""On-disk feature caches and model files of the reference ("next" row f2): written and read in the reference's
own layout, so that a run saved by either side can be replayed by the other.

feature caches   <output_dir>/features_{detector,RPN,segmentation}/
                     positives_cl_{c}_batch_{b}   negatives_cl_{c}_batch_{b}   reg_{x,c,y}_batch_{i}
                 each a `torch.save`d tensor.  A batch file is what the reference spills when a batch reaches
                 BATCH_SIZE rows (box_head_getProposals.py:168-171,211-221,247-249,286-288;
                 rpn_getProposals.py:232-238,302-304,359-361,402-404,439-449; mask_head_getProposals.py:121-137) plus
                 whatever is still open when the harvest ends (extract_features_detector.py:193-248;
                 extract_features_RPN.py:171-198): same file names, same rows in the same order, the same batch cuts
                 (the harvesters record where the reference closes a batch), empty tensors for classes without rows.
                 Readers: odx.utils.load_features_classifier / load_features_regressor (py_od_utils.py:120-224).
model files      <output_dir>/{classifier,regressor,stats}_{rpn,detector,segmentation}: `torch.save`d python
                 objects (run_experiment_online_rpn_ood_oos.py:117-120,280-288).  Classifier lists written by the
                 reference pickle falkon's estimator classes; `load_models` resolves those names to the odx
                 estimators (same attributes: alpha_, ny_points_, kernel.sigma), so no falkon install is needed.
"""
import io
import os
import pickle

import torch


def _save(t, path):
    torch.save(t.clone() if torch.is_tensor(t) else t, path)


def _mkdir(output_dir, name):
    d = os.path.join(output_dir, name)
    os.makedirs(d, exist_ok=True)
    return d


def _save_class_batches(d, prefix, clss, batches, D, device):
    """Non-empty batches under their own index; a class with nothing at all gets one empty batch 0."""
    if len(batches) == 1 and batches[0].shape[0] == 0:
        _save(torch.empty((0, D), device=device), os.path.join(d, '%s_cl_%d_batch_0' % (prefix, clss)))
        return
    for b, t in enumerate(batches):
        if t.shape[0] > 0:
            _save(t, os.path.join(d, '%s_cl_%d_batch_%d' % (prefix, clss, b)))


def _save_regressor(d, h):
    marks = h._X.marks
    for i, (x, c, y) in enumerate(zip(h._X.batches(), h._C.batches(marks), h._Y.batches(marks))):
        if x.shape[0] > 0:
            _save(x, os.path.join(d, 'reg_x_batch_%d' % i))
            _save(c, os.path.join(d, 'reg_c_batch_%d' % i))
            _save(y, os.path.join(d, 'reg_y_batch_%d' % i))


def save_detector_features(h, output_dir, use_only_gt_positives=True, mask_harvester=None):
    """DetectorHarvester (+ optional MaskHarvester) -> features_detector/ (+ features_segmentation/)."""
    d = _mkdir(output_dir, 'features_detector')
    for c in range(h.num_classes):
        for b, g in enumerate(h._neg[c]):
            if g.n > 0:
                _save(g.view(), os.path.join(d, 'negatives_cl_%d_batch_%d' % (c, b)))
        if use_only_gt_positives:
            _save_class_batches(d, 'positives', c, h._pos[c].batches(), h.D, h.device)
    _save_regressor(d, h)
    if mask_harvester is not None:
        m = mask_harvester
        ds = _mkdir(output_dir, 'features_segmentation')
        for c in range(m.num_classes):
            _save_class_batches(ds, 'positives', c, m._pos[c].batches(), m.D, m.device)
            _save_class_batches(ds, 'negatives', c, m._neg[c].batches(), m.D, m.device)


def save_rpn_features(h, output_dir):
    """RPNHarvester -> features_RPN/.  Anchor types without a visible anchor get empty batch-0 files when they are
    dropped (rpn_getProposals.py:232-238); the final flush walks `anchors_ids` (extract_features_RPN.py:173)."""
    d = _mkdir(output_dir, 'features_RPN')
    for i in getattr(h, 'invisible', []):
        for prefix in ('negatives', 'positives'):
            _save(torch.empty((0, h.D), device=h.device), os.path.join(d, '%s_cl_%d_batch_0' % (prefix, i)))
    for c in getattr(h, 'anchors_ids', range(h.A)):
        for b, g in enumerate(h._neg[c]):
            if g.n > 0:
                _save(g.view(), os.path.join(d, 'negatives_cl_%d_batch_%d' % (c, b)))
        _save_class_batches(d, 'positives', c, h._pos[c].batches(), h.D, h.device)
    _save_regressor(d, h)


# ------------------------------------------------------------------------------------------------ model files
MODEL_KINDS = ('classifier', 'regressor', 'stats')


def save_models(output_dir, tag, classifier=None, regressor=None, stats=None):
    """torch.save the trained objects as <kind>_<tag> (tag: rpn / detector / segmentation)."""
    os.makedirs(output_dir, exist_ok=True)
    for kind, obj in zip(MODEL_KINDS, (classifier, regressor, stats)):
        if obj is not None:
            torch.save(obj, os.path.join(output_dir, '%s_%s' % (kind, tag)))


class _FalkonNamesUnpickler(pickle.Unpickler):
    """Resolve falkon's estimator / kernel / option classes to the odx ones (attribute-compatible)."""
    _MAP = {'InCoreFalkon': 'InCoreFalkon', 'Falkon': 'Falkon', 'GaussianKernel': 'GaussianKernel',
            'FalkonOptions': 'FalkonOptions'}

    def find_class(self, module, name):
        if module == 'falkon' or module.startswith('falkon.'):
            from . import falkon as ofk
            if name in self._MAP:
                return getattr(ofk, self._MAP[name])
            raise pickle.UnpicklingError("no odx counterpart for %s.%s" % (module, name))
        return super().find_class(module, name)


class _FalkonPickle:
    """`pickle_module` for torch.load."""
    __name__ = 'odx_falkon_pickle'
    Unpickler = _FalkonNamesUnpickler
    load = staticmethod(lambda f, **kw: _FalkonNamesUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _FalkonNamesUnpickler(io.BytesIO(b), **kw).load())
    dump, dumps = staticmethod(pickle.dump), staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL
    PickleError, PicklingError, UnpicklingError = pickle.PickleError, pickle.PicklingError, pickle.UnpicklingError
    Pickler = pickle.Pickler


def load_models(output_dir, tag, map_location=None):
    """(classifier, regressor, stats) of <tag>; a missing file gives None."""
    out = []
    for kind in MODEL_KINDS:
        path = os.path.join(output_dir, '%s_%s' % (kind, tag))
        out.append(torch.load(path, map_location=map_location, pickle_module=_FalkonPickle, weights_only=False)
                   if os.path.exists(path) else None)
    return tuple(out)
