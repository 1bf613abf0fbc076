"""Test-time detection post-processing and VOC-style evaluation ("next" row f3).

masks            MaskPostProcessor + Masker (mrcnn_modified/modeling/roi_heads/mask_head/inference.py:27-62,119-191):
                 sigmoid of the pixel scores, channel of the predicted label, pasted into the image on the GPU;
                 segmentation AP = the same matching with mask IoU, `difficult` not consulted (icw_eval.py:404-518).
post-processing  OnlineDetectionPostProcessor.forward / filter_results
                 (src/modules/accuracy-evaluator/OnlineDetectionPostProcessor.py:12-79): decode the
                 per-class box deltas against the proposals (+1 widths, clamp), clip, keep scores
                 above SCORE_THRESH (-2 in the shipped configs), per-class NMS (HIP kernel), then at
                 most DETECTIONS_PER_IMAGE detections over all classes by a k-th-value threshold.
evaluation       eval_detection_icw: precision / recall per class with the PASCAL-VOC greedy matching
                 on "+1" integer boxes, duplicates counted as false positives, `difficult` boxes
                 ignored; AP as VOC07 11-point interpolation or area under the monotone envelope;
                 mAP = nanmean (mrcnn_modified/data/datasets/evaluation/icubworld/icw_eval.py:227-403)
                 and the RPN's average recall (mrcnn_modified/modeling/rpn/average_recall.py:5-10).
"""
from collections import defaultdict

import numpy as np
import torch

from . import backend as _backend
from .utils import decode_boxes_detector, mask_iou


class _Boxes:
    def __init__(self, bbox, size):
        self.bbox, self.size = bbox, size


def postprocess_detections(cls_scores, bbox_pred, proposals, img_size, score_thresh=-2.0, nms_thresh=0.3,
                           detections_per_img=100, proposals_size=None):
    """cls_scores (R, C+1), bbox_pred (R, 4 (C+1)), proposals (R, 4) given in a frame of
    proposals_size = (width, height) and rescaled to img_size when the two differ
    ->  dict(boxes (K, 4), scores (K,), labels (K,)); None when there is no foreground class."""
    if proposals_size is not None and tuple(proposals_size) != tuple(img_size):
        rw, rh = img_size[0] / proposals_size[0], img_size[1] / proposals_size[1]
        proposals = proposals * proposals.new_tensor([rw, rh, rw, rh])
    boxes = decode_boxes_detector(_Boxes(proposals, img_size), bbox_pred)
    return filter_results(boxes, cls_scores, img_size, score_thresh, nms_thresh, detections_per_img)


def filter_results(boxes, cls_scores, img_size, score_thresh=-2.0, nms_thresh=0.3, detections_per_img=100):
    """Clip, threshold, per-class NMS and global top-k of already decoded boxes (R, 4 (C+1)) — or
    class-agnostic (R, 4), repeated per class as OnlineDetectionPostProcessor_standalone.py:51-58 does."""
    be = _backend.get_backend()
    num_classes = cls_scores.shape[1]
    if boxes.shape[1] == 4:
        boxes = boxes.repeat(1, num_classes)
    boxes = boxes.reshape(-1, num_classes, 4).clone()
    boxes[..., 0].clamp_(0, img_size[0] - 1)
    boxes[..., 2].clamp_(0, img_size[0] - 1)
    boxes[..., 1].clamp_(0, img_size[1] - 1)
    boxes[..., 3].clamp_(0, img_size[1] - 1)
    keep_all = cls_scores > score_thresh
    if num_classes < 2:
        return None
    if hasattr(be, "nms_batched") and boxes.shape[0] > 0:
        # all foreground classes at once: the rows of a class that pass the threshold sorted by descending score (stable,
        # ties in row order — what the class-by-class loop below does with nonzero() + a stable argsort), ONE launch pair
        # for the suppression of every class, ONE nonzero() for the survivors (class-major, rank order = the order the loop
        # concatenates in).  The loop costs ~10 launches and two host round trips per class: 5-6 ms per image at 30 classes.
        fg = keep_all[:, 1:].t()                                                      # (C, R)
        sc = torch.where(fg, cls_scores[:, 1:].t(), cls_scores.new_full((), float("-inf")))
        order = torch.argsort(sc, dim=1, descending=True, stable=True)
        sc = torch.gather(sc, 1, order)
        bs = torch.gather(boxes[:, 1:].permute(1, 0, 2), 1, order.unsqueeze(2).expand(-1, -1, 4)).contiguous()
        keep = be.nms_batched(bs, fg.sum(1), nms_thresh)
        cls, rank = keep.nonzero(as_tuple=True)
        b, s, l = bs[cls, rank], sc[cls, rank], cls + 1
    else:
        out_b, out_s, out_l = [], [], []
        for j in range(1, num_classes):                       # 0 is the background slot
            inds = keep_all[:, j].nonzero().squeeze(1)
            sj, bj = cls_scores[inds, j], boxes[inds, j]
            k = be.nms(bj, sj, nms_thresh) if inds.numel() else inds
            out_b.append(bj[k])
            out_s.append(sj[k])
            out_l.append(torch.full((k.numel(),), j, dtype=torch.int64, device=sj.device))
        b, s, l = torch.cat(out_b), torch.cat(out_s), torch.cat(out_l)
    n = s.numel()
    if n > detections_per_img > 0:
        thresh, _ = torch.kthvalue(s.cpu(), n - detections_per_img + 1)
        keep = torch.nonzero(s >= thresh.item()).squeeze(1)
        b, s, l = b[keep], s[keep], l[keep]
    return {"boxes": b, "scores": s, "labels": l}


class OnlineDetectionPostProcessor:
    """Same constructor / forward contract as the reference class (OnlineDetectionPostProcessor.py:11-33);
    returns a BoxList with `scores` and `labels` fields."""

    def __init__(self, score_thresh=-2.0, nms=0.3, detections_per_img=100, **_ignored):
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img

    def forward(self, x, proposals, num_classes, img_size):
        from .boxlist import get_boxlist_class
        cls_scores, bbox_pred = x
        p = proposals[0]
        res = postprocess_detections(cls_scores[:, :num_classes], bbox_pred[:, :4 * num_classes], p.bbox, img_size, self.score_thresh,
                                     self.nms, self.detections_per_img, proposals_size=p.size)
        if res is None:
            return None
        out = get_boxlist_class()(res["boxes"], tuple(img_size), mode="xyxy")
        out.add_field("scores", res["scores"])
        out.add_field("labels", res["labels"])
        return out

    __call__ = forward


def select_class_masks(x, labels):
    """x (R, C+1, S, S) pixel scores -> (R, S, S) probabilities of each detection's own label (inference.py:38-45)."""
    idx = torch.arange(x.shape[0], device=x.device)
    return x.sigmoid()[idx, labels.to(x.device)]


def paste_masks(mask_prob, boxes, img_size, thresh=0.5, padding=1):
    """Masker: (R, S, S) probabilities + (R, 4) boxes -> (R, height, width) bool, img_size = (width, height)."""
    return _backend.get_backend().paste_masks(mask_prob, boxes, img_size[1], img_size[0], thresh, padding)


def _iou_plus1(a, b):
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt + 1, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (aa[:, None] + ab[None, :] - inter)


def detection_prec_rec(preds, gts, iou_thresh=0.5, key="boxes"):
    """preds: per image dict(boxes, labels, scores); gts: per image dict(boxes, labels[, difficult]).
    key="masks": pasted (K, H, W) masks instead of boxes, mask IoU, `difficult` ignored (segmentation AP).
    Returns (prec, rec): lists indexed by class id (None where undefined)."""
    n_pos, score, match = defaultdict(int), defaultdict(list), defaultdict(list)
    seg = key == "masks"
    for p, g in zip(preds, gts):
        pb, pl, ps = (np.asarray(p[k]) for k in (key, "labels", "scores"))
        gb, gl = np.asarray(g[key]), np.asarray(g["labels"])
        gd = np.zeros(len(gl), dtype=bool) if seg else np.asarray(g.get("difficult", np.zeros(len(gl), dtype=bool))).astype(bool)
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            pm = pl == l
            order = ps[pm].argsort()[::-1]
            pbl, psl = pb[pm][order], ps[pm][order]
            gm = gl == l
            gbl, gdl = gb[gm], gd[gm]
            n_pos[l] += int(np.logical_not(gdl).sum())
            score[l].extend(psl)
            if len(pbl) == 0:
                continue
            if len(gbl) == 0:
                match[l].extend((0,) * pbl.shape[0])
                continue
            if seg:
                with np.errstate(divide="ignore", invalid="ignore"):
                    iou = mask_iou(pbl.astype(bool), np.rint(gbl).astype(bool))
            else:
                a, b = pbl.copy(), gbl.copy()      # arithmetic stays in the boxes' own dtype (f32 from the heads)
                a[:, 2:] += 1          # VOC evaluates integer-typed boxes
                b[:, 2:] += 1
                iou = _iou_plus1(a, b)
            idx = iou.argmax(axis=1)
            idx[iou.max(axis=1) < iou_thresh] = -1
            taken = np.zeros(len(gbl), dtype=bool)
            for gi in idx:
                if gi < 0:
                    match[l].append(0)
                    continue
                match[l].append(-1 if gdl[gi] else (0 if taken[gi] else 1))
                taken[gi] = True
    if not n_pos:
        return [None], [None]
    n_cls = max(n_pos) + 1
    prec, rec = [None] * n_cls, [None] * n_cls
    for l in n_pos:
        m = np.array(match[l], dtype=np.int8)[np.array(score[l]).argsort()[::-1]]
        tp, fp = np.cumsum(m == 1), np.cumsum(m == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec[l] = tp / (fp + tp)
        if n_pos[l] > 0:
            rec[l] = tp / n_pos[l]
    return prec, rec


def average_precision(prec, rec, use_07_metric=True):
    ap = np.empty(len(prec))
    for l in range(len(prec)):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        p = np.nan_to_num(prec[l])
        if use_07_metric:
            ap[l] = sum((p[rec[l] >= t].max() if np.any(rec[l] >= t) else 0.0) for t in np.arange(0.0, 1.1, 0.1)) / 11
        else:
            mpre = np.concatenate(([0], p, [0]))
            mrec = np.concatenate(([0], rec[l], [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap


def eval_detection(preds, gts, iou_thresh=0.5, use_07_metric=True, key="boxes"):
    prec, rec = detection_prec_rec(preds, gts, iou_thresh, key)
    ap = average_precision(prec, rec, use_07_metric)
    return {"ap": ap, "map": np.nanmean(ap)}


def average_recall(best_iou_per_gt):
    """2 * mean(max(IoU_best - 0.5, 0)) over the ground-truth boxes of an image (average_recall.py:5-10)."""
    v = np.asarray(best_iou_per_gt, dtype=np.float64)
    return float(2.0 * np.mean(np.maximum(v - 0.5, 0.0))) if v.size else 0.0
