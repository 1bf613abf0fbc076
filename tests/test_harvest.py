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

# Every reference-pinned comparison below runs twice: on host tensors (here, `-m "not gpu"`) and with the inputs, the
# harvesters' buffers and all their sort / gather / scatter work on HIP tensors (`-m gpu`, on the MI355X).  The random
# draws come from the global CPU generator in both arms (as in the reference), so the golden vectors are the same.
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def T(a, device):
    return torch.from_numpy(np.asarray(a)).to(device)


def N(t):
    return t.detach().cpu().numpy()


def _run(shuffle, device="cpu"):
    D, C, ITER, BS, NIMG = (int(G[k]) for k in ("D", "C", "ITER", "BS", "NIMG"))
    h = DetectorHarvester(D, C, ITER, BS, NIMG, shuffle_negatives=shuffle, device=device)
    torch.manual_seed(123)
    for im in range(NIMG):
        h.add_image(T(G["x_%d" % im], device), T(G["prop_%d" % im], device), T(G["gt_%d" % im], device),
                    G["labels_%d" % im].tolist(), [320, 240])
    return h


@pytest.mark.parametrize("device", DEVICES)
def test_fill_mode_matches_reference(device):
    h = _run(False, device)
    C, ITER = int(G["C"]), int(G["ITER"])
    assert h.still_to_complete == G["fill_still_to_complete"].tolist()
    negatives, positives, COXY = h.finalize()
    assert COXY["X"].device.type == device
    assert np.array_equal(N(COXY["C"]), G["fill_C"])
    assert np.array_equal(N(COXY["X"]), G["fill_X"])
    assert np.allclose(N(COXY["Y"]), G["fill_Y"], atol=1e-6)
    assert COXY["O"] is None
    for c in range(C):
        assert np.array_equal(N(positives[c]), G["fill_pos_%d" % c])
        assert len(negatives[c]) == ITER
        for b in range(ITER):
            assert np.array_equal(N(negatives[c][b]), G["fill_neg_%d_%d" % (c, b)]), (c, b)
    h.add_test_image(T(G["x_0"], device), T(G["prop_0"], device), len(G["labels_0"]), [320, 240])
    tb = h.test_boxes[0]
    assert np.array_equal(tb["boxes"], G["test_boxes"]) and np.array_equal(tb["feat"], G["test_feat"])
    assert np.array_equal(tb["gt"], G["test_gt"]) and tb["img_size"].tolist() == [320, 240]


@pytest.mark.parametrize("device", DEVICES)
def test_shuffle_mode_matches_reference_before_the_final_permutation(device):
    h = _run(True, device)
    C = int(G["C"])
    for c in range(C):
        total = N(torch.cat([g.view() for g in h._neg[c]]))
        assert np.array_equal(total, G["shuf_neg_%d" % c])
        assert np.array_equal(N(h._pos[c].view()), G["shuf_pos_%d" % c])
    assert np.array_equal(N(h._X.view()), G["shuf_X"])
    negatives, _, _ = h.finalize()
    assert all(len(n) == int(G["ITER"]) for n in negatives)
    assert sum(len(b) for b in negatives[0]) == min(len(G["shuf_neg_0"]), int(G["ITER"]) * int(G["BS"]))


@pytest.mark.parametrize("device", DEVICES)
def test_iou_matches_reference_and_oracle(device):
    gt, prop = T(G["iou_gt"], device), T(G["iou_prop"], device)
    got = N(box_iou_plus1(gt[None], prop)[0])
    assert np.allclose(got, G["iou_out"], atol=1e-6)
    assert np.allclose(roi_ref.compute_overlap(G["iou_gt"], G["iou_prop"]), G["iou_out"], atol=1e-6)


@pytest.mark.parametrize("device", DEVICES)
def test_add_new_class_and_empty_images(device):
    h = DetectorHarvester(4, 1, 2, 3, 2, device=device)
    h.add_new_class()
    assert h.num_classes == 2 and h.still_to_complete == [0, 1]
    torch.manual_seed(0)
    h.add_image(torch.randn(5, 4).to(device), (torch.rand(5, 4) * 50).to(device), torch.zeros(0, 4).to(device), [], [100, 100])
    negatives, positives, COXY = h.finalize()
    assert COXY["X"].shape == (0, 4) and positives[0].shape == (0, 4)
    assert sum(len(b) for b in negatives[1]) == 3


# ------------------------------------------------------------------ on-line RPN harvesting (A12)
R = np.load(os.path.join(os.path.dirname(__file__), "golden", "rpn_harvest_golden.npz"))


def _run_rpn(shuffle, device="cpu"):
    from odx.harvest import RPNHarvester
    D, A, H, W, ITER, BS, NIMG = (int(R[k]) for k in ("D", "A", "H", "W", "ITER", "BS", "NIMG"))
    h = RPNHarvester(D, A, ITER, BS, NIMG, shuffle_negatives=shuffle, device=device)
    anchors = T(R["anchors"], device)
    torch.manual_seed(321)
    for im in range(NIMG):
        h.add_image(T(R["t_%d" % im], device), anchors, (W * 16, H * 16), T(R["gt_%d" % im], device))
    return h


@pytest.mark.parametrize("device", DEVICES)
def test_rpn_fill_mode_matches_reference(device):
    """Against the reference's own RPNModule.forward (rpn_getProposals.py) run on the same inputs."""
    from odx.extract import cell_anchors, grid_anchors
    assert np.array_equal(N(grid_anchors(int(R["H"]), int(R["W"]), 16, cell_anchors(16).to(device))), R["anchors"])
    h = _run_rpn(False, device)
    A, ITER = int(R["A"]), int(R["ITER"])
    assert h.anchors_ids == R["fill_anchors_ids"].tolist()
    assert h.still_to_complete == R["fill_still_to_complete"].tolist()
    negatives, positives, COXY = h.finalize()
    assert COXY["X"].device.type == device
    assert np.array_equal(N(COXY["C"]), R["fill_C"]) and np.array_equal(N(COXY["X"]), R["fill_X"])
    assert np.allclose(N(COXY["Y"]), R["fill_Y"], atol=1e-6)
    for c in range(A):
        assert np.array_equal(N(positives[c]), R["fill_pos_%d" % c]), c
        for b in range(ITER):
            assert np.array_equal(N(negatives[c][b]), R["fill_neg_%d_%d" % (c, b)]), (c, b)
    assert sum(len(p) for p in positives) > 0 and len(COXY["X"]) == sum(len(p) for p in positives)


@pytest.mark.parametrize("device", DEVICES)
def test_rpn_shuffle_mode_matches_reference_before_the_final_permutation(device):
    h = _run_rpn(True, device)
    for c in range(int(R["A"])):
        assert np.array_equal(N(torch.cat([g.view() for g in h._neg[c]])), R["shuf_neg_%d" % c]), c
        assert np.array_equal(N(h._pos[c].view()), R["shuf_pos_%d" % c])
    assert np.array_equal(N(h._X.view()), R["shuf_X"])


# ------------------------------------------------------------------ on-line segmentation harvesting (A13)
@pytest.mark.parametrize("device", DEVICES)
def test_mask_harvest_matches_reference(device):
    """Pixel sampling / bookkeeping against the reference's own ROIMaskHead.forward
    (mask_head_getProposals.py) on the same activations, masks and RNG seed."""
    from odx.harvest import MaskHarvester, project_masks_on_boxes
    Mg = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_harvest_golden.npz"))
    D, C, S = int(Mg["D"]), int(Mg["C"]), int(Mg["S"])
    h = MaskHarvester(D, C, batch_size=60, sampling_factor=0.3, device=device)
    torch.manual_seed(77)
    for im in range(3):
        act = T(Mg["act_%d" % im], device)
        mg = project_masks_on_boxes(T(Mg["masks_%d" % im], device), T(Mg["boxes_%d" % im], device), S)
        assert mg.device.type == device
        assert mg.shape == (act.shape[0], S, S) and set(np.unique(N(mg))) <= {0.0, 1.0}
        h.add_image(act, mg, Mg["labels_%d" % im].tolist())
    negatives, positives = h.finalize()
    for c in range(C):
        assert np.allclose(N(positives[c]), Mg["pos_%d" % c], atol=1e-6), c
        assert np.allclose(N(negatives[c]), Mg["neg_%d" % c], atol=1e-6), c
    assert sum(len(p) for p in positives) > 0 and sum(len(n) for n in negatives) > 0


@pytest.mark.parametrize("device", DEVICES)
def test_project_masks_on_boxes_basics(device):
    from odx.harvest import project_masks_on_boxes
    m = torch.zeros(2, 40, 40, dtype=torch.uint8, device=device)
    m[0, 10:30, 10:30] = 1
    m[1, :, :20] = 1
    out = project_masks_on_boxes(m, torch.tensor([[10.0, 10, 30, 30], [0.0, 0, 40, 40]], device=device), 14)
    assert out.shape == (2, 14, 14) and bool((out[0] == 1).all())
    assert bool((out[1][:, :6] == 1).all()) and bool((out[1][:, 8:] == 0).all())
    assert project_masks_on_boxes(torch.zeros(0, 8, 8), torch.zeros(0, 4), 14).shape == (0,)


def _project_one_by_one(masks, boxes, M):
    """The per-object formulation (crop, F.interpolate, truncate) the batched project_masks_on_boxes must reproduce."""
    out = []
    H, W = masks.shape[1], masks.shape[2]
    for m, b in zip(masks, boxes):
        x1, y1, x2, y2 = [int(round(float(v))) for v in b]
        x1, y1 = min(max(x1, 0), W - 1), min(max(y1, 0), H - 1)
        x2, y2 = min(max(x2, 0), W - 1), min(max(y2, 0), H - 1)
        x2, y2 = max(x2, x1 + 1), max(y2, y1 + 1)
        crop = m[y1:y2, x1:x2].float()[None, None]
        r = torch.nn.functional.interpolate(crop, size=(M, M), mode="bilinear", align_corners=False)[0, 0]
        out.append((r + 1e-6).to(torch.uint8).float())
    return torch.stack(out)


@pytest.mark.parametrize("device", DEVICES)
def test_batched_mask_projection_equals_the_per_object_resize(device):
    from odx.harvest import project_masks_on_boxes
    g = torch.Generator().manual_seed(12)
    H, W, G = 97, 131, 40
    masks = (torch.rand((G, H, W), generator=g) > 0.35).to(torch.uint8)
    for k in range(G):                                    # blobs, so that the crops are not pure noise
        cy, cx = int(torch.randint(H, (1,), generator=g)), int(torch.randint(W, (1,), generator=g))
        masks[k, max(cy - 20, 0):cy + 20, max(cx - 25, 0):cx + 25] = 1
    xy = torch.rand((G, 2), generator=g) * torch.tensor([W * 0.8, H * 0.8]) - 5.0          # some boxes start outside the image
    wh = torch.cat((torch.rand((G // 2, 2), generator=g) * 90 + 1, torch.rand((G - G // 2, 2), generator=g) * 6))   # up- and down-scaling
    boxes = torch.cat((xy, xy + wh), 1)
    for M in (14, 7, 28):
        got = project_masks_on_boxes(masks.to(device), boxes.to(device), M).cpu()
        ref = _project_one_by_one(masks, boxes, M)
        assert got.shape == ref.shape and torch.equal(got, ref), (M, int((got != ref).sum()))


@pytest.mark.parametrize("device", DEVICES)
def test_rpn_positives_with_one_host_read_equal_the_box_by_box_walk(device):
    """RPNHarvester.add_image decides the positives from ONE block of counts read from the device (round 5); the reference
    walks the ground-truth boxes one by one with `g in positive_gts` tests on tensors (rpn_getProposals.py:383-400), which is
    what this class did until round 4.  Random scenes — several boxes, boxes sharing a coordinate with another box (the
    reference's any-element-equal test then skips them), duplicated boxes, boxes without any anchor over the threshold —
    through both: the same positives in the same order (regressor rows X / Y / C and the per-type positives)."""
    from odx.extract import cell_anchors, grid_anchors
    from odx.harvest import RPNHarvester, box_iou_plus1
    D, A, H, W = 6, 15, 12, 16
    anchors = grid_anchors(H, W, 16, cell_anchors(16)).to(device)
    g = torch.Generator().manual_seed(5)
    scenes = []
    for s in range(12):
        G = 1 + s % 4
        xy = torch.rand(G, 2, generator=g) * torch.tensor([W * 16 * 0.6, H * 16 * 0.6])
        wh = 12 + torch.rand(G, 2, generator=g) * torch.tensor([W * 16 * 0.5, H * 16 * 0.5])
        gt = torch.cat((xy, xy + wh), dim=1)
        if s % 3 == 1 and G > 1:
            gt[1, 0] = gt[0, 0]                       # shares x1 with box 0
        if s % 5 == 2 and G > 2:
            gt[2] = gt[0]                             # a duplicate
        if s % 4 == 3:
            gt[0] = torch.tensor([3.0, 3.0, 9.0, 8.0])    # too small for any anchor to pass 0.7
        scenes.append((torch.randn(D, H, W, generator=g), gt))

    def walk(h, t, gt):
        """The box-by-box form (round 4's code): returns the positives' anchor indices in append order, type-sorted."""
        ious = torch.squeeze(box_iou_plus1(gt, h.anchors))
        if gt.shape[0] > 1:
            ious, idx = torch.max(ious, dim=0)
            assoc = gt[idx]
        else:
            ious = ious.reshape(-1)
            assoc = gt[0].expand(h.anchors.shape[0], 4)
        pos = torch.nonzero(ious > h.pos_iou_thresh).reshape(-1)
        pos_gt = assoc[pos]
        for gb in gt:
            if bool((gb[None, :] == pos_gt).any()):
                continue
            mine = (assoc == gb[None, :]).all(dim=1)
            if bool(mine.any()):
                best = ious[mine].max()
                pos = torch.cat((pos, torch.nonzero(mine & (ious == best)).reshape(-1)))
                pos_gt = assoc[pos]
        pcls = h.cls[pos]
        return pos[torch.argsort(pcls, stable=True)]

    h = RPNHarvester(D, A, 2, 40, len(scenes), device=device)
    torch.manual_seed(1)
    want_rows, n_extra = [], 0
    for t, gt in scenes:
        t, gt = t.to(device), gt.to(device)
        before = h._X.n
        h.add_image(t, anchors, (W * 16, H * 16), gt)
        sel = walk(h, t, gt)
        n_extra += int((box_iou_plus1(gt, h.anchors).max(dim=0)[0][sel] <= h.pos_iou_thresh).sum())
        want = t[:, h.rows[sel], h.cols[sel]].t().reshape(-1, D)
        got = h._X.view()[before:]
        assert got.shape == want.shape and torch.equal(got, want)
        want_rows.append(h.cls[sel])
    assert n_extra > 0                                  # the scenes do exercise the added best anchors
    assert torch.equal(h._C.view().reshape(-1), torch.cat(want_rows).to(torch.float32))
    assert sum(p.n for p in h._pos) == h._X.n


@pytest.mark.gpu
def test_staged_uploads_deliver_the_same_tensors(monkeypatch):
    """harvest.to_device through the page-locked staging block (odx.options staged_uploads = True): every tensor arrives as a plain copy
    would deliver it — dtypes, shapes, more bytes than both halves of the block hold (the halves are re-entered behind their
    fences), copies issued from two streams."""
    from odx import harvest
    import odx
    monkeypatch.setattr(odx.options.current(), "staged_uploads", True)
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    side = torch.cuda.Stream()
    kept = []
    for i in range(900):
        n = int(torch.randint(1, 3000, (1,), generator=g))
        host = [torch.randint(0, 1 << 40, (n,), generator=g), torch.randn((n // 7 + 1, 7), generator=g),
                torch.randint(0, 2, (n,), generator=g).bool()][i % 3]
        if i % 5 == 0:
            with torch.cuda.stream(side):
                d = harvest.to_device(host, dev)
            torch.cuda.current_stream().wait_stream(side)
        else:
            d = harvest.to_device(host, dev)
        kept.append((host, d))
    torch.cuda.synchronize()
    assert sum(h.numel() * h.element_size() for h, _ in kept) > 4 * harvest._Staging.HALF
    for h, d in kept:
        assert d.is_cuda and d.dtype == h.dtype and d.shape == h.shape and torch.equal(d.cpu(), h)


@pytest.mark.gpu
@pytest.mark.parametrize("G", [1, 2, 5, 17])
def test_labelling_kernels_equal_the_tensor_statements(G):
    """odx_rpn_label_f32 / odx_det_label_f32 / odx_box_targets_f32 against the tensor statements they replace in the harvesters'
    prepare / commit (the CPU path: box_iou_plus1, maxima, associations, flags, counts), bit for bit — overlaps are compared with
    thresholds, so the overlaps have to be the same bits; duplicate ground-truth boxes, boxes sharing single coordinates, boxes
    partly outside the image, thresholds hit exactly."""
    import odx
    from odx import harvest
    be = odx.get_backend()
    g = torch.Generator().manual_seed(G)
    W, H, A, C = 800.0, 600.0, 15, 30
    xy = torch.rand((G, 2), generator=g) * torch.tensor([W * 0.7, H * 0.7])
    gt = torch.cat((xy, xy + 30 + torch.rand((G, 2), generator=g) * 250), dim=1).round()
    if G >= 2:
        gt[1] = gt[0]                                        # a duplicate box
    if G >= 5:
        gt[3, 0] = gt[2, 0]                                  # boxes sharing one coordinate
        gt[4, 2:] += 400                                     # partly outside the image
    n = 5000
    axy = torch.rand((n, 2), generator=g) * torch.tensor([W, H]) - 40
    anchors = torch.cat((axy, axy + 16 + torch.rand((n, 2), generator=g) * 300), dim=1).round()
    anchors[:G] = gt                                          # overlap exactly 1
    cls = torch.randint(0, A, (n,), generator=g)
    gt_d, an_d, cls_d = gt.cuda(), anchors.cuda(), cls.cuda()
    # ---- RPN
    iou_all = harvest.box_iou_plus1(gt_d, an_d)
    ious, idx = torch.max(iou_all, dim=0)
    assoc = gt_d[idx]
    thr_n, thr_p = float(ious[n // 2]), float(ious[n // 3])       # thresholds that some overlap equals exactly
    neg_mask, over = ious < thr_n, ious > thr_p
    onehot = cls_d[:, None] == torch.arange(A, device="cuda")[None, :]
    mine = (assoc[None, :, :] == gt_d[:, None, :]).all(dim=2)
    best = torch.where(mine, ious[None, :], torch.full_like(ious, -1.0)[None, :]).max(dim=1)[0]
    extra = mine & (ious[None, :] == best[:, None])
    share = (assoc[None, :, :] == gt_d[:, None, :]).any(dim=2)
    want = torch.cat(((neg_mask[:, None] & onehot).sum(0), (share & over[None, :]).sum(1), mine.any(dim=1).long(), (over[:, None] & onehot).sum(0),
                      (extra[:, :, None] & onehot[None, :, :]).sum(1).reshape(-1)))
    k_ious, k_assoc, k_neg, k_over, k_extra, k_cnt = be.rpn_label(gt_d, an_d, cls_d, A, thr_n, thr_p)
    assert torch.equal(k_ious, ious) and torch.equal(k_assoc, assoc) and torch.equal(k_neg, neg_mask) and torch.equal(k_over, over)
    assert torch.equal(k_extra, extra) and torch.equal(k_cnt.long(), want)
    # ---- detector
    labels = [1 + int(v) for v in torch.randint(0, C, (G,), generator=g)]
    R = 700
    pxy = torch.rand((R, 2), generator=g) * torch.tensor([W, H]) - 30
    props = torch.cat((pxy, pxy + 10 + torch.rand((R, 2), generator=g) * 300), dim=1)
    props[:G] = gt
    prop = harvest.clamp_boxes_(props.clone().cuda(), (W, H))
    gtc = harvest.clamp_boxes_(gt_d.clone(), (W, H))
    iou = harvest.box_iou_plus1(gtc, prop)
    overlap = torch.zeros((R, C), device="cuda")
    for j in range(G):
        overlap[:, labels[j] - 1] = torch.max(overlap[:, labels[j] - 1], iou[j])
    bestv, _ = iou.max(dim=0)
    first = (iou == bestv[None, :]).float().argmax(dim=0)
    asc = torch.where(bestv > 0, first, torch.full((R,), -1, dtype=torch.int64, device="cuda"))
    lab = torch.tensor([l - 1 for l in labels], device="cuda")
    reg_min, thr = float(overlap[:, labels[0] - 1][R // 2]), 0.3
    sel = (overlap[:, lab].t() > reg_min) & (asc[None, :] == torch.arange(G, device="cuda")[:, None])
    in_image = sorted({l - 1 for l in labels})
    cmask = overlap[:, in_image] < thr
    up = torch.tensor([l - 1 for l in labels] + in_image, dtype=torch.int32, device="cuda")
    k_prop, k_ov, k_sel, k_cm, k_cnt = be.det_label(gt_d, up[:G], props.cuda(), C, (W, H), reg_min, thr, up[G:])
    assert torch.equal(k_prop, prop) and torch.equal(k_ov, overlap) and torch.equal(k_sel, sel) and torch.equal(k_cm, cmask)
    assert torch.equal(k_cnt.long(), torch.cat((sel.sum(1), cmask.sum(0))))
    # ---- targets
    ex, tg = prop[G:G + 300], prop[:G][torch.randint(0, G, (300,), generator=g).cuda()]
    sw, sh = ex[:, 2] - ex[:, 0] + 1, ex[:, 3] - ex[:, 1] + 1
    sx, sy = ex[:, 0] + 0.5 * sw, ex[:, 1] + 0.5 * sh
    gw, gh = tg[:, 2] - tg[:, 0] + 1, tg[:, 3] - tg[:, 1] + 1
    gx, gy = tg[:, 0] + 0.5 * gw, tg[:, 1] + 0.5 * gh
    wantt = torch.stack(((gx - sx) / sw, (gy - sy) / sh, torch.log(gw / sw), torch.log(gh / sh)), dim=1)
    gott = be.box_targets(ex, tg)
    # (the logarithm is the device library's logf here and torch's own kernel there: a unit in the last place apart)
    # (a box pushed out of the image clamps to zero width: log of 0 / of a negative ratio — the same non-numbers on both sides)
    fin = torch.isfinite(wantt[:, 2:])
    assert torch.equal(gott[:, :2], wantt[:, :2]) and torch.equal(torch.isfinite(gott[:, 2:]), fin)
    assert float((gott[:, 2:][fin] - wantt[:, 2:][fin]).abs().max()) <= 1e-6


def _walk(fill, cb, per_batch, bs, k, n, phantom):
    """The batch walk as the reference's loops state it, one class (box_head_getProposals.py:226-290 with phantom rows,
    rpn_getProposals.py:283-331 without): returns (rows landed per batch, their source offsets, full batches passed)."""
    B = len(fill)
    take, lo, skipped, taken = [0] * B, [0] * B, 0, 0
    for b in range(cb, B):
        if fill[b] >= bs:
            skipped += 1
            continue
        if phantom:
            end = taken + min(per_batch, bs - fill[b], k - taken)
            a, z = min(taken, n), min(end, n)
        else:
            end = taken + min(per_batch, bs - fill[b], k - taken, n - taken)
            a, z = taken, end
        take[b], lo[b] = z - a, a
        taken = end
        if taken == k:
            break
    return take, lo, skipped


def test_fill_plan_equals_the_batch_walk_on_random_states():
    """harvest.fill_plan (all classes of an image at once, array arithmetic) against the class-by-class batch walk it replaces,
    on random fill states: full batches in the middle, classes at their last batch, fewer rows than asked for (phantom rows in
    the detector's form, an exhausted walk in the on-line RPN's), per_batch 1 and larger than a batch's room."""
    from odx.harvest import fill_plan
    rng = np.random.default_rng(123)
    for trial in range(400):
        nc, B = int(rng.integers(1, 9)), int(rng.integers(1, 12))
        bs = int(rng.integers(1, 40))
        k = int(rng.integers(1, 60))
        per_batch = -(-k // B) if rng.random() < 0.7 else int(rng.integers(1, k + 1))
        fill = rng.integers(0, bs + 1, (nc, B))
        fill[rng.random((nc, B)) < 0.3] = bs                                  # plenty of full batches
        cb = rng.integers(0, B, nc)
        n = np.where(rng.random(nc) < 0.3, rng.integers(0, k + 1, nc), k)       # some classes with fewer rows than k (0 included)
        for phantom in (True, False):
            take, lo, skipped = fill_plan(fill.astype(np.int64), cb.astype(np.int64), per_batch, bs, k, n.astype(np.int64), phantom)
            for c in range(nc):
                wt, wl, ws = _walk(fill[c].tolist(), int(cb[c]), per_batch, bs, k, int(n[c]), phantom)
                assert take[c].tolist() == wt, (trial, phantom, c, take[c].tolist(), wt)
                assert skipped[c] == ws, (trial, phantom, c)
                assert all(lo[c][b] == wl[b] for b in range(B) if wt[b]), (trial, phantom, c)
