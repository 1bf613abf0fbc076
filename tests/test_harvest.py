"""Detector feature harvesting (A11) against golden vectors produced by the reference's own
ROIBoxHead.forward_train / forward_test (tests/golden/harvest_golden.npz), and the IoU helper
against the reference's compute_overlap_torch."""
import os

import numpy as np
import pytest
import torch

from odx.harvest import DetectorHarvester, box_iou_plus1
from oracle import roi_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "harvest_golden.npz"))


def _run(shuffle):
    D, C, ITER, BS, NIMG = (int(G[k]) for k in ("D", "C", "ITER", "BS", "NIMG"))
    h = DetectorHarvester(D, C, ITER, BS, NIMG, shuffle_negatives=shuffle, device="cpu")
    torch.manual_seed(123)
    for im in range(NIMG):
        h.add_image(torch.from_numpy(G["x_%d" % im]), torch.from_numpy(G["prop_%d" % im]), torch.from_numpy(G["gt_%d" % im]),
                    G["labels_%d" % im].tolist(), [320, 240])
    return h


def test_fill_mode_matches_reference():
    h = _run(False)
    C, ITER = int(G["C"]), int(G["ITER"])
    assert h.still_to_complete == G["fill_still_to_complete"].tolist()
    negatives, positives, COXY = h.finalize()
    assert np.array_equal(COXY["C"].numpy(), G["fill_C"])
    assert np.array_equal(COXY["X"].numpy(), G["fill_X"])
    assert np.allclose(COXY["Y"].numpy(), G["fill_Y"], atol=1e-6)
    assert COXY["O"] is None
    for c in range(C):
        assert np.array_equal(positives[c].numpy(), G["fill_pos_%d" % c])
        assert len(negatives[c]) == ITER
        for b in range(ITER):
            assert np.array_equal(negatives[c][b].numpy(), G["fill_neg_%d_%d" % (c, b)]), (c, b)
    h.add_test_image(torch.from_numpy(G["x_0"]), torch.from_numpy(G["prop_0"]), len(G["labels_0"]), [320, 240])
    tb = h.test_boxes[0]
    assert np.array_equal(tb["boxes"], G["test_boxes"]) and np.array_equal(tb["feat"], G["test_feat"])
    assert np.array_equal(tb["gt"], G["test_gt"]) and tb["img_size"].tolist() == [320, 240]


def test_shuffle_mode_matches_reference_before_the_final_permutation():
    h = _run(True)
    C = int(G["C"])
    for c in range(C):
        total = torch.cat([g.view() for g in h._neg[c]]).numpy()
        assert np.array_equal(total, G["shuf_neg_%d" % c])
        assert np.array_equal(h._pos[c].view().numpy(), G["shuf_pos_%d" % c])
    assert np.array_equal(h._X.view().numpy(), G["shuf_X"])
    negatives, _, _ = h.finalize()
    assert all(len(n) == int(G["ITER"]) for n in negatives)
    assert sum(len(b) for b in negatives[0]) == min(len(G["shuf_neg_0"]), int(G["ITER"]) * int(G["BS"]))


def test_iou_matches_reference_and_oracle():
    gt, prop = torch.from_numpy(G["iou_gt"]), torch.from_numpy(G["iou_prop"])
    got = box_iou_plus1(gt[None], prop)[0].numpy()
    assert np.allclose(got, G["iou_out"], atol=1e-6)
    assert np.allclose(roi_ref.compute_overlap(G["iou_gt"], G["iou_prop"]), G["iou_out"], atol=1e-6)


def test_add_new_class_and_empty_images():
    h = DetectorHarvester(4, 1, 2, 3, 2, device="cpu")
    h.add_new_class()
    assert h.num_classes == 2 and h.still_to_complete == [0, 1]
    torch.manual_seed(0)
    h.add_image(torch.randn(5, 4), torch.rand(5, 4) * 50, torch.zeros(0, 4), [], [100, 100])
    negatives, positives, COXY = h.finalize()
    assert COXY["X"].shape == (0, 4) and positives[0].shape == (0, 4)
    assert sum(len(b) for b in negatives[1]) == 3


# ------------------------------------------------------------------ on-line RPN harvesting (A12)
R = np.load(os.path.join(os.path.dirname(__file__), "golden", "rpn_harvest_golden.npz"))


def _run_rpn(shuffle):
    from odx.harvest import RPNHarvester
    D, A, H, W, ITER, BS, NIMG = (int(R[k]) for k in ("D", "A", "H", "W", "ITER", "BS", "NIMG"))
    h = RPNHarvester(D, A, ITER, BS, NIMG, shuffle_negatives=shuffle, device="cpu")
    anchors = torch.from_numpy(R["anchors"])
    torch.manual_seed(321)
    for im in range(NIMG):
        h.add_image(torch.from_numpy(R["t_%d" % im]), anchors, (W * 16, H * 16), torch.from_numpy(R["gt_%d" % im]))
    return h


def test_rpn_fill_mode_matches_reference():
    """Against the reference's own RPNModule.forward (rpn_getProposals.py) run on the same inputs."""
    from odx.extract import cell_anchors, grid_anchors
    assert np.array_equal(grid_anchors(int(R["H"]), int(R["W"]), 16, cell_anchors(16)).numpy(), R["anchors"])
    h = _run_rpn(False)
    A, ITER = int(R["A"]), int(R["ITER"])
    assert h.anchors_ids == R["fill_anchors_ids"].tolist()
    assert h.still_to_complete == R["fill_still_to_complete"].tolist()
    negatives, positives, COXY = h.finalize()
    assert np.array_equal(COXY["C"].numpy(), R["fill_C"]) and np.array_equal(COXY["X"].numpy(), R["fill_X"])
    assert np.allclose(COXY["Y"].numpy(), R["fill_Y"], atol=1e-6)
    for c in range(A):
        assert np.array_equal(positives[c].numpy(), R["fill_pos_%d" % c]), c
        for b in range(ITER):
            assert np.array_equal(negatives[c][b].numpy(), R["fill_neg_%d_%d" % (c, b)]), (c, b)
    assert sum(len(p) for p in positives) > 0 and len(COXY["X"]) == sum(len(p) for p in positives)


def test_rpn_shuffle_mode_matches_reference_before_the_final_permutation():
    h = _run_rpn(True)
    for c in range(int(R["A"])):
        assert np.array_equal(torch.cat([g.view() for g in h._neg[c]]).numpy(), R["shuf_neg_%d" % c]), c
        assert np.array_equal(h._pos[c].view().numpy(), R["shuf_pos_%d" % c])
    assert np.array_equal(h._X.view().numpy(), R["shuf_X"])


# ------------------------------------------------------------------ on-line segmentation harvesting (A13)
def test_mask_harvest_matches_reference():
    """Pixel sampling / bookkeeping against the reference's own ROIMaskHead.forward
    (mask_head_getProposals.py) on the same activations, masks and RNG seed."""
    from odx.harvest import MaskHarvester, project_masks_on_boxes
    Mg = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_harvest_golden.npz"))
    D, C, S = int(Mg["D"]), int(Mg["C"]), int(Mg["S"])
    h = MaskHarvester(D, C, batch_size=60, sampling_factor=0.3, device="cpu")
    torch.manual_seed(77)
    for im in range(3):
        act = torch.from_numpy(Mg["act_%d" % im])
        mg = project_masks_on_boxes(torch.from_numpy(Mg["masks_%d" % im]), torch.from_numpy(Mg["boxes_%d" % im]), S)
        assert mg.shape == (act.shape[0], S, S) and set(np.unique(mg.numpy())) <= {0.0, 1.0}
        h.add_image(act, mg, Mg["labels_%d" % im].tolist())
    negatives, positives = h.finalize()
    for c in range(C):
        assert np.allclose(positives[c].numpy(), Mg["pos_%d" % c], atol=1e-6), c
        assert np.allclose(negatives[c].numpy(), Mg["neg_%d" % c], atol=1e-6), c
    assert sum(len(p) for p in positives) > 0 and sum(len(n) for n in negatives) > 0


def test_project_masks_on_boxes_basics():
    from odx.harvest import project_masks_on_boxes
    m = torch.zeros(2, 40, 40, dtype=torch.uint8)
    m[0, 10:30, 10:30] = 1
    m[1, :, :20] = 1
    out = project_masks_on_boxes(m, torch.tensor([[10.0, 10, 30, 30], [0.0, 0, 40, 40]]), 14)
    assert out.shape == (2, 14, 14) and bool((out[0] == 1).all())
    assert bool((out[1][:, :6] == 1).all()) and bool((out[1][:, 8:] == 0).all())
    assert project_masks_on_boxes(torch.zeros(0, 8, 8), torch.zeros(0, 4), 14).shape == (0,)
