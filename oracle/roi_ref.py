"""CPU restatement (numpy, plain loops) of RoIAlign forward, greedy NMS and the reference's IoU —
TEST ORACLE ONLY.

PARITY UNPINNED for RoIAlign / NMS: the reference reaches them through maskrcnn_benchmark's
compiled extension (facebookresearch/maskrcnn-benchmark, unpinned HEAD, INSTALLATION_GUIDE.md:61-63;
not vendored, not installable offline).  Restated from the published operator definitions:
RoIAlign of Mask R-CNN in maskrcnn_benchmark's legacy form (no half-pixel shift, roi size
clamped to >= 1, adaptive sampling grid ceil(roi / bins) when sampling_ratio == 0, samples
outside [-1, size] contribute 0); greedy NMS with +1 areas and a strict > threshold.
compute_overlap IS pinned: it restates mrcnn_modified/utils/evaluations.py:4-18 which runs here
(tests/golden/make_golden.py -> harvest_golden.npz).
"""
import math

import numpy as np


def _bilinear(plane, y, x):
    H, W = plane.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return 0.0
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * plane[yl, xl] + hy * lx * plane[yl, xh] + ly * hx * plane[yh, xl] + ly * lx * plane[yh, xh]


def roi_align(feat, rois, spatial_scale, output_size, sampling_ratio=0):
    feat = np.asarray(feat, dtype=np.float64)
    rois = np.asarray(rois, dtype=np.float64)
    N, C, H, W = feat.shape
    PH, PW = output_size
    out = np.zeros((rois.shape[0], C, PH, PW))
    for r, roi in enumerate(rois):
        b = int(roi[0])
        x1, y1, x2, y2 = (np.float32(v) * np.float32(spatial_scale) for v in roi[1:])
        rw, rh = max(float(x2 - x1), 1.0), max(float(y2 - y1), 1.0)
        bw, bh = rw / PW, rh / PH
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / PH))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / PW))
        for ph in range(PH):
            for pw in range(PW):
                for iy in range(gh):
                    y = float(y1) + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = float(x1) + pw * bw + (ix + 0.5) * bw / gw
                        for c in range(C):
                            out[r, c, ph, pw] += _bilinear(feat[b, c], y, x)
                out[r, :, ph, pw] /= gh * gw
    return out


def fpn_levels(boxes, scales, canonical_scale=224.0, canonical_level=4.0, eps=1e-6):
    """maskrcnn_benchmark.modeling.poolers.LevelMapper (reached from FPN2MLPFeatureExtractor's Pooler,
    mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py:61-68; PARITY UNPINNED like roi_align): the
    pyramid level (0-based) each box is pooled from,
        floor(4 + log2(sqrt(area) / 224 + 1e-6)) clamped to [k_min, k_max], minus k_min,
    area with the +1 pixel convention (BoxList.area, mode xyxy), k_min / k_max = -log2 of the first / last scale.
    Evaluated in f32 like the torch ops it restates."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    f = np.float32
    area = (b[:, 2] - b[:, 0] + f(1)) * (b[:, 3] - b[:, 1] + f(1))
    s = np.sqrt(area).astype(np.float32)
    k_min = -np.log2(np.float32(scales[0]))
    k_max = -np.log2(np.float32(scales[-1]))
    lv = np.floor(f(canonical_level) + np.log2(s / f(canonical_scale) + f(eps)))
    return (np.clip(lv, k_min, k_max) - k_min).astype(np.int64)


def roi_align_fpn(feats, rois, scales, output_size, sampling_ratio=2):
    """Pooler.forward over several levels: every RoI through roi_align on the map / scale of its level."""
    rois = np.asarray(rois, dtype=np.float64)
    lv = fpn_levels(rois[:, 1:5], scales)
    C = np.asarray(feats[0]).shape[1]
    out = np.zeros((rois.shape[0], C, output_size[0], output_size[1]))
    for l, (f, sc) in enumerate(zip(feats, scales)):
        idx = np.flatnonzero(lv == l)
        if len(idx):
            out[idx] = roi_align(f, rois[idx], sc, output_size, sampling_ratio)
    return out


def iou_plus1(a, b):
    w = max(min(a[2], b[2]) - max(a[0], b[0]) + 1.0, 0.0)
    h = max(min(a[3], b[3]) - max(a[1], b[1]) + 1.0, 0.0)
    inter = w * h
    return inter / ((a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0) + (b[2] - b[0] + 1.0) * (b[3] - b[1] + 1.0) - inter)


def nms(boxes, scores, thr):
    """Indices kept, in descending score order (stable)."""
    boxes = np.asarray(boxes, dtype=np.float64)
    order = np.argsort(-np.asarray(scores), kind="stable")
    keep, removed = [], np.zeros(len(order), bool)
    for a, i in enumerate(order):
        if removed[a]:
            continue
        keep.append(int(i))
        for b in range(a + 1, len(order)):
            if not removed[b] and iou_plus1(boxes[i], boxes[order[b]]) > thr:
                removed[b] = True
    return np.array(keep, dtype=np.int64)


def compute_overlap(gt, prop):
    """IoU of one ground-truth box against proposals, +1 convention, 0 where they do not touch
    (mrcnn_modified/utils/evaluations.py:4-18)."""
    gt = np.asarray(gt, dtype=np.float64)
    prop = np.asarray(prop, dtype=np.float64)
    xmin, ymin = np.maximum(gt[0], prop[:, 0]), np.maximum(gt[1], prop[:, 1])
    xmax, ymax = np.minimum(gt[2], prop[:, 2]), np.minimum(gt[3], prop[:, 3])
    inter = (xmax - xmin + 1) * (ymax - ymin + 1)
    ga = (gt[2] - gt[0] + 1) * (gt[3] - gt[1] + 1)
    pa = (prop[:, 2] - prop[:, 0] + 1) * (prop[:, 3] - prop[:, 1] + 1)
    ov = inter / (ga + pa - inter)
    ov = np.where(xmax - xmin + 1 > 0, ov, 0.0)
    return np.where(ymax - ymin + 1 > 0, ov, 0.0)


def paste_mask(mask, box, im_h, im_w, thresh=0.5, padding=1):
    """paste_mask_in_image (mrcnn_modified/modeling/roi_heads/mask_head/inference.py:119-159) in plain loops:
    zero-pad, expand the box by (S + 2 pad) / S, truncate to int, bilinear resize (align_corners=False, f32), threshold."""
    f = np.float32
    S = mask.shape[-1]
    Sp = S + 2 * padding
    pm = np.zeros((Sp, Sp), dtype=f)
    pm[padding:Sp - padding, padding:Sp - padding] = mask
    scale = f(float(Sp) / S)
    b = np.asarray(box, dtype=f)
    w_half, h_half = (b[2] - b[0]) * f(0.5) * scale, (b[3] - b[1]) * f(0.5) * scale
    x_c, y_c = (b[2] + b[0]) * f(0.5), (b[3] + b[1]) * f(0.5)
    bx0, bx2, by0, by2 = int(x_c - w_half), int(x_c + w_half), int(y_c - h_half), int(y_c + h_half)
    w, h = max(bx2 - bx0 + 1, 1), max(by2 - by0 + 1, 1)
    out = np.zeros((im_h, im_w), dtype=bool)
    sx, sy = f(Sp) / f(w), f(Sp) / f(h)
    for y in range(max(by0, 0), min(by2 + 1, im_h)):
        fy = max(sy * (f(y - by0) + f(0.5)) - f(0.5), f(0))
        iy0 = int(fy)
        iy1 = iy0 + (1 if iy0 < Sp - 1 else 0)
        ly1 = f(fy - f(iy0))
        ly0 = f(1) - ly1
        for x in range(max(bx0, 0), min(bx2 + 1, im_w)):
            fx = max(sx * (f(x - bx0) + f(0.5)) - f(0.5), f(0))
            ix0 = int(fx)
            ix1 = ix0 + (1 if ix0 < Sp - 1 else 0)
            lx1 = f(fx - f(ix0))
            lx0 = f(1) - lx1
            v = ly0 * (lx0 * pm[iy0, ix0] + lx1 * pm[iy0, ix1]) + ly1 * (lx0 * pm[iy1, ix0] + lx1 * pm[iy1, ix1])
            out[y, x] = v > thresh
    return out
