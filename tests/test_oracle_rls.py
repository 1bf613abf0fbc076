"""The RLS oracle (oracle/rls_ref.py) against golden vectors produced by the REFERENCE'S OWN
RegionRefinerTrainer / RegionPredictor / decode_boxes_detector (tests/golden/rls_golden.npz,
generator tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import rls_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rls_golden.npz"))


@pytest.mark.parametrize("tag,is_rpn", [("det", False), ("rpn", True)])
def test_train_matches_reference(tag, is_rpn):
    C = G["C"] if not is_rpn else G["C"] - 1
    models = rls_ref.train(C, G["X"], G["Y"], 5, float(G["lambda"]), is_rpn=is_rpn)
    assert len(models) == int(G[tag + "_num_models"])
    for i, m in enumerate(models):
        assert (m is None) == bool(G["%s_%d_none" % (tag, i)])
        if m is None:
            continue
        for key, name in (("mu", "mu"), ("T", "T"), ("T_inv", "T_inv"), ("W", "W")):
            ref = G["%s_%d_%s" % (tag, i, name)]
            assert np.abs(m[key] - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (tag, i, key)
        ref = G["%s_%d_losses" % (tag, i)]
        assert np.abs(m["losses"] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_apply_matches_reference():
    models = rls_ref.train(G["C"], G["X"], G["Y"], 5, float(G["lambda"]))[:2]
    for im in range(2):
        got = rls_ref.apply(models, G["apply_boxes_%d" % im], G["apply_feat_%d" % im], G["apply_gt_%d" % im], (320, 240))
        ref = G["apply_out_%d" % im]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 2e-3  # f32 reference arithmetic on pixel coordinates up to 320


def test_decode_boxes_detector_matches_reference():
    got = rls_ref.decode_boxes_detector(G["apply_boxes_0"], G["decode_in"], (320, 240))
    assert np.abs(got - G["decode_out"]).max() < 1e-3
