"""Row f2: the on-disk feature caches and model files, against what the reference's own harvesters write with
save_features=True on the same inputs (tests/golden/feature_cache_golden.npz, make_golden.py --only-feature-cache)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from odx import storage
from odx.utils import load_features_classifier, load_features_regressor
from tests.test_harvest import DEVICES, _run, _run_rpn

CACHE = np.load(os.path.join(os.path.dirname(__file__), "golden", "feature_cache_golden.npz"))


def _compare_dir(d, tag):
    want = CACHE["%s/__files__" % tag].tolist()
    assert sorted(os.listdir(d)) == want
    for n in want:
        got, ref = torch.load(os.path.join(d, n)).cpu().numpy(), CACHE["%s/%s" % (tag, n)]
        assert got.shape == ref.shape, (tag, n, got.shape, ref.shape)
        if n.startswith("reg_y"):
            assert np.allclose(got, ref, atol=1e-6), (tag, n)
        else:
            assert np.array_equal(got, ref), (tag, n)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("shuffle", [False, True])
def test_detector_cache_files_equal_the_reference(tmp_path, shuffle, device):
    h = _run(shuffle, device)
    storage.save_detector_features(h, str(tmp_path))
    _compare_dir(str(tmp_path / "features_detector"), "det_shuf" if shuffle else "det_fill")
    # and the readers give back what finalize() returns
    pos, neg = load_features_classifier(str(tmp_path / "features_detector"), cpu_tensor=(device == "cpu"))
    negatives, positives, COXY = h.finalize()
    for c in range(h.num_classes):
        assert pos[c].device.type == device and torch.equal(pos[c], positives[c])
        if not shuffle:
            assert all(torch.equal(a, b) for a, b in zip(neg[c], [n for n in negatives[c] if len(n)]))
    back = load_features_regressor(str(tmp_path / "features_detector"))
    assert all(torch.equal(back[k].cpu(), COXY[k].cpu()) for k in ("X", "C", "Y"))


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("shuffle", [False, True])
def test_rpn_cache_files_equal_the_reference(tmp_path, shuffle, device):
    h = _run_rpn(shuffle, device)
    storage.save_rpn_features(h, str(tmp_path))
    _compare_dir(str(tmp_path / "features_RPN"), "rpn_shuf" if shuffle else "rpn_fill")


@pytest.mark.parametrize("device", DEVICES)
def test_segmentation_cache_roundtrip(tmp_path, device):
    from odx.harvest import MaskHarvester
    torch.manual_seed(0)
    det = _run(False, device)
    m = MaskHarvester(6, 3, batch_size=40, sampling_factor=0.5, device=device)
    for _ in range(9):
        feats = torch.randn(2, 6, 7, 7).to(device)
        m.add_image(feats, (torch.rand(2, 7, 7) > 0.5).float().to(device), [1, 3])        # class 2 never appears
    storage.save_detector_features(det, str(tmp_path), mask_harvester=m)
    files = sorted(os.listdir(tmp_path / "features_segmentation"))
    assert "positives_cl_1_batch_0" in files and torch.load(tmp_path / "features_segmentation" / "positives_cl_1_batch_0").shape == (0, 6)
    assert any(f.startswith("positives_cl_0_batch_1") for f in files)            # cut where a batch reached 40 rows
    pos, neg = load_features_classifier(str(tmp_path / "features_segmentation"), is_segm=True, cpu_tensor=(device == "cpu"))
    n_ref, p_ref = m.finalize()
    for c in (0, 2):
        assert pos[c].device.type == device and torch.equal(pos[c], p_ref[c]) and torch.equal(neg[c], n_ref[c])


def test_model_files_roundtrip_and_falkon_names(tmp_path):
    from odx.falkon import GaussianKernel, InCoreFalkon
    est = InCoreFalkon(kernel=GaussianKernel(7.5), penalty=1e-4, M=3)
    est.alpha_, est.ny_points_ = torch.arange(3.0).view(3, 1), torch.ones(3, 5)
    stats = {"mean": torch.zeros(5), "std": torch.ones(5), "mean_norm": torch.tensor(2.0)}
    reg = np.array([{"mu": torch.zeros(4), "T": torch.eye(4), "T_inv": torch.eye(4), "Beta": {"0": {"weights": torch.ones(6)}}}], dtype=object)
    storage.save_models(str(tmp_path), "detector", classifier=[est, None], regressor=reg, stats=stats)
    assert sorted(os.listdir(tmp_path)) == ["classifier_detector", "regressor_detector", "stats_detector"]
    clf, reg2, stats2 = storage.load_models(str(tmp_path), "detector")
    assert clf[1] is None and torch.equal(clf[0].alpha_, est.alpha_) and clf[0].kernel.sigma == 7.5
    assert torch.equal(reg2[0]["T"], torch.eye(4)) and torch.equal(stats2["std"], torch.ones(5))
    assert storage.load_models(str(tmp_path), "rpn") == (None, None, None)
    # a classifier list pickled with falkon's own class names loads as odx estimators
    fk, fm, fi, fkern = (types.ModuleType(n) for n in ("falkon", "falkon.models", "falkon.models.incore_falkon", "falkon.kernels"))

    class _Base:
        pass
    InC = type("InCoreFalkon", (_Base,), {"__module__": "falkon.models.incore_falkon"})
    GK = type("GaussianKernel", (_Base,), {"__module__": "falkon.kernels"})
    fi.InCoreFalkon, fkern.GaussianKernel = InC, GK
    mods = {"falkon": fk, "falkon.models": fm, "falkon.models.incore_falkon": fi, "falkon.kernels": fkern}
    sys.modules.update(mods)
    try:
        theirs = InC()
        theirs.kernel = GK()
        theirs.kernel.sigma = torch.tensor([4.0])
        theirs.alpha_, theirs.ny_points_, theirs.M = torch.ones(2, 1), torch.zeros(2, 5), 2
        torch.save([theirs], tmp_path / "classifier_rpn")
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    clf, _, _ = storage.load_models(str(tmp_path), "rpn")
    assert type(clf[0]) is InCoreFalkon and type(clf[0].kernel) is GaussianKernel
    assert float(clf[0].kernel.sigma) == 4.0 and torch.equal(clf[0].alpha_, torch.ones(2, 1))
