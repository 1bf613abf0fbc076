"""Row f3: detection post-processing and VOC-style evaluation against vectors produced by the reference's own
OnlineDetectionPostProcessor / icw_eval code (tests/golden/make_golden.py --only-postprocess)."""
import os

import numpy as np
import pytest
import torch

import odx
from odx.postprocess import average_precision, average_recall, detection_prec_rec, eval_detection, postprocess_detections

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def check_postprocess_cases(device):
    g = np.load(os.path.join(GOLD, "postprocess_golden.npz"))
    for ci in range(int(g["n_cases"])):
        R, C, dpi, pw, ph, iw, ih = (int(v) for v in g["c%d_meta" % ci])
        thr, nms = (float(v) for v in g["c%d_thr" % ci])
        res = postprocess_detections(torch.from_numpy(g["c%d_scores" % ci]).to(device), torch.from_numpy(g["c%d_deltas" % ci]).to(device),
                                     torch.from_numpy(g["c%d_props" % ci]).to(device), (iw, ih), score_thresh=thr, nms_thresh=nms,
                                     detections_per_img=dpi, proposals_size=(pw, ph))
        assert res["labels"].cpu().numpy().tolist() == g["c%d_labels" % ci].tolist(), ci
        np.testing.assert_array_equal(res["scores"].cpu().numpy(), g["c%d_out_scores" % ci])
        np.testing.assert_allclose(res["boxes"].cpu().numpy().reshape(-1, 4), g["c%d_boxes" % ci].reshape(-1, 4), rtol=0, atol=1e-4)


def test_postprocess_matches_reference_on_cpu_with_oracle_backend():
    from tests.oracle_backend import OracleBackend
    odx.set_backend(OracleBackend(np.float64))
    try:
        check_postprocess_cases("cpu")
    finally:
        odx.set_backend(None)


@pytest.mark.gpu
def test_postprocess_matches_reference_on_gpu():
    odx.set_backend(None)
    check_postprocess_cases("cuda")


def _load_eval():
    g = np.load(os.path.join(GOLD, "eval_golden.npz"))
    preds, gts = [], []
    for im in range(int(g["NIMG"])):
        preds.append({"boxes": g["pb_%d" % im], "labels": g["pl_%d" % im], "scores": g["ps_%d" % im]})
        gts.append({"boxes": g["gb_%d" % im], "labels": g["gl_%d" % im], "difficult": g["gd_%d" % im]})
    return g, preds, gts


def test_precision_recall_match_reference():
    g, preds, gts = _load_eval()
    prec, rec = detection_prec_rec(preds, gts, 0.5)
    assert len(prec) == int(g["n_cls"])
    for l in range(len(prec)):
        hp, hr = g["has_%d" % l]
        assert (prec[l] is not None) == bool(hp) and (rec[l] is not None) == bool(hr)
        if hp:
            np.testing.assert_array_equal(prec[l], g["prec_%d" % l])
        if hr:
            np.testing.assert_array_equal(rec[l], g["rec_%d" % l])


@pytest.mark.parametrize("thr", [0.5, 0.7])
@pytest.mark.parametrize("m07", [True, False])
def test_ap_matches_reference(thr, m07):
    g, preds, gts = _load_eval()
    r = eval_detection(preds, gts, iou_thresh=thr, use_07_metric=m07)
    tag = "iou%02d_%s" % (int(thr * 100), "voc07" if m07 else "area")
    np.testing.assert_allclose(r["ap"], g[tag + "_ap"], rtol=0, atol=1e-12, equal_nan=True)
    assert abs(r["map"] - float(g[tag + "_map"])) < 1e-12


def test_eval_edge_cases():
    assert detection_prec_rec([], []) == ([None], [None])
    assert np.isnan(average_precision([None], [None])).all()
    # one perfect detection, one duplicate -> AP 1.0 under both metrics (duplicate ranks below)
    gt = [{"boxes": np.array([[10, 10, 50, 50]], np.float32), "labels": np.array([1])}]
    pr = [{"boxes": np.array([[10, 10, 50, 50], [11, 11, 50, 50]], np.float32), "labels": np.array([1, 1]), "scores": np.array([0.9, 0.8], np.float32)}]
    r = eval_detection(pr, gt, use_07_metric=True)
    assert r["ap"][1] == pytest.approx(1.0) and np.isnan(r["ap"][0])
    assert eval_detection(pr, gt, use_07_metric=False)["ap"][1] == pytest.approx(1.0)
    assert average_recall([0.75, 0.4, 1.0]) == pytest.approx(2 * (0.25 + 0 + 0.5) / 3)
    assert average_recall([]) == 0.0
    # segmentation AP: an image without detections (empty (0, H, W) masks) still counts its ground truth -> recall 1/2
    m = np.zeros((1, 8, 8), np.uint8)
    m[0, 2:6, 2:6] = 1
    gts = [{"boxes": gt[0]["boxes"], "labels": np.array([1]), "masks": m}] * 2
    prs = [{"boxes": gt[0]["boxes"], "labels": np.array([1]), "scores": np.array([0.9], np.float32), "masks": m},
           {"boxes": np.zeros((0, 4), np.float32), "labels": np.zeros(0, np.int64), "scores": np.zeros(0, np.float32), "masks": np.zeros((0, 8, 8), np.uint8)}]
    prec, rec = detection_prec_rec(prs, gts, 0.5, key="masks")
    assert prec[1].tolist() == [1.0] and rec[1].tolist() == [0.5]


def test_dropin_class_contract():
    """The reference-named class, imported the way the reference imports it, gives the golden result."""
    import importlib.util
    from odx.boxlist import BoxList
    from tests.oracle_backend import OracleBackend
    path = os.path.join(os.path.dirname(__file__), os.pardir, "online-detection_amd", "src", "modules", "accuracy-evaluator",
                        "OnlineDetectionPostProcessor.py")
    spec = importlib.util.spec_from_file_location("OnlineDetectionPostProcessor", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(GOLD, "postprocess_golden.npz"))
    R, C, dpi, pw, ph, iw, ih = (int(v) for v in g["c2_meta"])
    thr, nms = (float(v) for v in g["c2_thr"])
    odx.set_backend(OracleBackend(np.float64))
    try:
        pp = mod.OnlineDetectionPostProcessor(score_thresh=thr, nms=nms, detections_per_img=dpi)
        out = pp((torch.from_numpy(g["c2_scores"]), torch.from_numpy(g["c2_deltas"])), [BoxList(torch.from_numpy(g["c2_props"]), (pw, ph))],
                 C, (iw, ih))
    finally:
        odx.set_backend(None)
    assert out.size == (iw, ih) and len(out) == dpi
    assert out.get_field("labels").tolist() == g["c2_labels"].tolist()
    np.testing.assert_allclose(out.bbox.numpy(), g["c2_boxes"], atol=1e-4)


# ------------------------------------------------------------------ masks: class selection, pasting, segmentation AP
MG = np.load(os.path.join(GOLD, "masks_golden.npz"))


def check_pasting(device):
    from odx.postprocess import paste_masks, select_class_masks
    H, W = (int(v) for v in MG["HW"])
    prob = select_class_masks(torch.from_numpy(MG["logits"]).to(device), torch.from_numpy(MG["labels"]))
    np.testing.assert_allclose(prob.cpu().numpy(), MG["prob"], atol=1e-6)
    got = paste_masks(torch.from_numpy(MG["prob"]).to(device), torch.from_numpy(MG["boxes"]).to(device), (W, H)).cpu().numpy()
    ref = MG["pasted"].astype(bool)
    assert got.shape == ref.shape
    # identical apart from pixels whose interpolated value sits within rounding of the 0.5 threshold
    assert (got != ref).sum() <= 2, int((got != ref).sum())
    assert got[2].sum() == ref[2].sum() and got[0].any() and got[1].any()       # sub-pixel and out-of-image boxes


def test_mask_pasting_oracle_matches_reference():
    from tests.oracle_backend import OracleBackend
    odx.set_backend(OracleBackend(np.float64))
    try:
        check_pasting("cpu")
    finally:
        odx.set_backend(None)


@pytest.mark.gpu
def test_mask_pasting_on_gpu_matches_reference_and_oracle():
    from oracle import roi_ref
    odx.set_backend(None)
    check_pasting("cuda")
    be = odx.get_backend()
    g = torch.Generator().manual_seed(1)
    masks, boxes = torch.rand(7, 28, 28, generator=g), torch.rand(7, 4, generator=g) * 40
    boxes[:, 2:] += boxes[:, :2] + 3
    got = be.paste_masks(masks, boxes, 50, 70).cpu().numpy()
    want = np.stack([roi_ref.paste_mask(masks[i].numpy(), boxes[i].numpy(), 50, 70) for i in range(7)])
    assert (got != want).sum() <= 2
    assert be.paste_masks(masks[:0], boxes[:0], 50, 70).shape == (0, 50, 70)


def test_segmentation_ap_matches_reference():
    from tests.oracle_backend import OracleBackend
    from odx.postprocess import paste_masks
    H, W = (int(v) for v in MG["HW"])
    odx.set_backend(OracleBackend(np.float64))
    try:
        preds, gts = [], []
        for im in range(int(MG["seg_NIMG"])):
            pasted = paste_masks(torch.from_numpy(MG["seg_pm_%d" % im]), torch.from_numpy(MG["seg_pb_%d" % im]), (W, H)).numpy()
            preds.append({"masks": pasted, "labels": MG["seg_pl_%d" % im], "scores": MG["seg_ps_%d" % im]})
            gts.append({"masks": MG["seg_gm_%d" % im], "labels": MG["seg_gl_%d" % im]})
    finally:
        odx.set_backend(None)
    for m07 in (True, False):
        r = eval_detection(preds, gts, 0.5, m07, key="masks")
        np.testing.assert_allclose(r["ap"], MG["seg_ap_%s" % ("voc07" if m07 else "area")], atol=1e-12, equal_nan=True)


def test_standalone_accuracy_evaluator_dropin(tmp_path):
    """`import AccuracyEvaluator as ae` (the stand-alone O-OD evaluator of the reference): post-processing of the
    per-image BoxLists that testRegionClassifier / RegionRefiner.predict leave, then the reference's AP — on the golden
    post-processing case its kept detections are the reference's, and detections equal to the ground truth score AP 1."""
    import yaml
    from odx.boxlist import BoxList
    from tests import dropin
    from tests.oracle_backend import OracleBackend
    g = np.load(os.path.join(GOLD, "postprocess_golden.npz"))
    R, C, dpi, pw, ph, iw, ih = (int(v) for v in g["c2_meta"])
    thr, nms = (float(v) for v in g["c2_thr"])
    cfg = {"NUM_CLASSES": C, "CHOSEN_CLASSES": {i: "obj%d" % i for i in range(C)},
           "EVALUATION": {"SCORE_THRESH": thr, "NMS": nms, "DETECTIONS_PER_IMAGE": dpi, "IOU_THRESHOLDS": [0.5]}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))

    class Dataset:
        def __init__(self, gts):
            self.gts = gts

        def get_groundtruth(self, i):
            return self.gts[i]

    odx.set_backend(OracleBackend(np.float64))
    try:
        ev = dropin.load("AccuracyEvaluator").AccuracyEvaluator(path, str(tmp_path))
        # image 0: ground truth = the golden kept detections themselves -> every class present scores AP 1
        kept = BoxList(torch.from_numpy(g["c2_boxes"]), (iw, ih))
        kept.add_field("labels", torch.from_numpy(g["c2_labels"]))
        # predictions: already decoded per-class boxes (R, C, 4) + scores (R, C), as RegionRefiner.predict leaves them
        from odx.utils import decode_boxes_detector
        props = torch.from_numpy(g["c2_props"]) * torch.tensor([iw / pw, ih / ph, iw / pw, ih / ph])
        dec = decode_boxes_detector(BoxList(props, (iw, ih)), torch.from_numpy(g["c2_deltas"])[:, :4 * C])
        pred = BoxList(dec.reshape(R, C, 4), (iw, ih))
        pred.add_field("scores", torch.from_numpy(g["c2_scores"])[:, :C])
        with __import__("contextlib").redirect_stdout(__import__("io").StringIO()):
            res = ev.evaluate(Dataset([kept]), [pred])
    finally:
        odx.set_backend(None)
    present = sorted(set(g["c2_labels"].tolist()))
    assert all(abs(res["ap"][c] - 1.0) < 1e-12 for c in present) and abs(res["map"] - 1.0) < 1e-12
    assert "Detection mAP50: 1.0000" in open(os.path.join(str(tmp_path), "result.txt")).read()


@pytest.mark.gpu
def test_all_class_nms_equals_the_class_by_class_loop():
    """filter_results on the MI355X suppresses every class with one launch pair (odx_nms_batched_f32); result = the
    class-by-class loop's (threshold, stable sort, odx_nms_f32 per class) — same boxes, scores, labels, in the same
    order — with ragged classes: one class without any row above the threshold, one with a single row, ties in the scores."""
    import odx
    from odx import postprocess as pp
    odx.set_backend(None)
    be = odx.get_backend()
    g = torch.Generator().manual_seed(12)
    R, C = 333, 9
    xy = torch.rand((R, 2), generator=g) * torch.tensor([500.0, 300.0])
    wh = 30 + torch.rand((R, 2), generator=g) * 150
    props = torch.cat((xy, xy + wh), dim=1).cuda()
    deltas = (torch.randn((R, 4 * (C + 1)), generator=g) * 0.15).cuda()
    scores = (torch.randn((R, C + 1), generator=g) * 0.5 - 0.6)
    scores[:, 3] = -5.0                                   # no row of class 3 passes
    scores[:, 5] = -5.0
    scores[17, 5] = 0.4                                   # exactly one row of class 5 passes
    scores[40:60, 7] = 0.25                               # ties
    scores = scores.cuda()
    for thr, per_img in ((-1.0, 100), (-0.5, 0), (-2.0, 40)):
        got = pp.postprocess_detections(scores, deltas, props, (640, 480), thr, 0.3, per_img)

        class Loop:                                       # the same backend without the batched entry point
            def __getattr__(self, name):
                if name == "nms_batched":
                    raise AttributeError(name)
                return getattr(be, name)
        odx.set_backend(Loop())
        try:
            ref = pp.postprocess_detections(scores, deltas, props, (640, 480), thr, 0.3, per_img)
        finally:
            odx.set_backend(None)
        assert len(ref["scores"]) > 0
        for key in ("boxes", "scores", "labels"):
            assert torch.equal(got[key], ref[key]), (thr, key)
