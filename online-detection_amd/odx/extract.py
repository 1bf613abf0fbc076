"""Feature-extraction forward (A11 / A12 front end): Mask R-CNN R-50-C4 trunk -> RPN -> RoIAlign ->
conv5 head -> pooled RoI features, one image at a time, feeding odx.harvest at train time and the
on-line heads (odx.heads) at test time.

What the reference does here lives almost entirely in maskrcnn_benchmark (build_backbone, RPN
post-processing, Pooler/ROIAlign, nms; not vendored, unpinned HEAD) driven by
mrcnn_modified/modeling/detector/generalized_rcnn_getProposals.py:39-100,
mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py:13-52 and
mrcnn_modified/engine/feature_proposal_extractor.py:228-281.  This module restates that graph
with the architecture constants of the shipped configs (R-50-C4, FrozenBatchNorm, stride-in-1x1
bottlenecks, 15 anchors = 3 ratios x 5 sizes at stride 16, RPN pre/post-NMS 6000/300 @ 0.7,
RoIAlign 14 x 14 @ 1/16 with adaptive sampling, conv5 head -> D = 2048):
  * convolutions / pooling are PyTorch-ROCm ops (MIOpen) — plumbing per the project scope;
  * RoIAlign and NMS are the hand-written HIP kernels of csrc/roi.hip;
  * the RPN / box heads switch to the on-line FALKON + RLS heads when models are supplied.
No pretrained weights or datasets exist in this environment: weights are random (seeded) unless
a state dict is given, and images come from the caller.  PARITY UNPINNED (structure only).
"""
import contextlib
import os
import sys
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import backend as _backend
from . import options as _options
from .harvest import DetectorHarvester, MaskHarvester, RPNHarvester, project_masks_on_boxes, to_device
from .heads import OnlineBoxPredictor, OnlineMaskPredictor, OnlineRPNHead  # noqa: F401


# ---------------------------------------------------------------------------- anchors
def cell_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0)):
    """Detectron anchor windows around one stride x stride cell: ratios enumerated first, sizes
    second (rpn/anchor_generator.py:216-243 — the order `classifier = ind % 15` decodes)."""
    base = float(stride)
    ctr = 0.5 * (base - 1.0)
    out = []
    for r in aspect_ratios:
        w0 = round(math.sqrt(base * base / r))
        h0 = round(w0 * r)
        for s in sizes:
            k = s / base
            w, h = w0 * k, h0 * k
            out.append([ctr - 0.5 * (w - 1), ctr - 0.5 * (h - 1), ctr + 0.5 * (w - 1), ctr + 0.5 * (h - 1)])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(H, W, stride, cells):
    """(H * W * A, 4): location-major (y, then x), anchor-type-minor (anchor_generator.py:73-95)."""
    sx = torch.arange(0, W * stride, stride, dtype=torch.float32, device=cells.device)
    sy = torch.arange(0, H * stride, stride, dtype=torch.float32, device=cells.device)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cells.view(1, -1, 4)).reshape(-1, 4)


DELTA_CLAMP = math.log(1000.0 / 16)          # BoxCoder's bbox_xform_clip


def decode_deltas(deltas, boxes, clamp=DELTA_CLAMP):
    """BoxCoder(weights=(1, 1, 1, 1)).decode with the +1 width convention."""
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    dx, dy = deltas[:, 0], deltas[:, 1]
    dw, dh = deltas[:, 2].clamp(max=clamp), deltas[:, 3].clamp(max=clamp)
    pcx, pcy = dx * w + cx, dy * h + cy
    pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw - 1, pcy + 0.5 * ph - 1), dim=1)


def _nms_takes_max_keep(be):
    """Whether the backend's nms has the early-stopping contract (max_keep / sorted_desc) — asked of its signature, so
    that a TypeError raised INSIDE a backend's nms is never mistaken for a missing keyword."""
    import inspect
    try:
        params = inspect.signature(be.nms).parameters
    except (TypeError, ValueError):
        return False
    return "max_keep" in params and "sorted_desc" in params


def rpn_proposals(objectness, box_regression, anchors, img_size, pre_nms_top_n=6000, post_nms_top_n=300,
                  nms_thresh=0.7, min_size=0):
    """RPNPostProcessor.forward_for_single_feature_map (rpn/inference.py:76-123) for one image:
    sigmoid, top-k, decode, clip, drop small boxes, NMS (HIP kernel), keep post_nms_top_n.
    objectness (1, A, H, W), box_regression (1, 4A, H, W); img_size = (width, height)."""
    be = _backend.get_backend()
    _, A, H, W = objectness.shape
    obj = objectness.permute(0, 2, 3, 1).reshape(-1).sigmoid()
    reg = box_regression.view(1, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(-1, 4)
    k = min(pre_nms_top_n, obj.numel())
    score, idx = obj.topk(k, sorted=True)
    boxes = decode_deltas(reg[idx], anchors.to(reg.device)[idx])
    boxes[:, 0].clamp_(0, img_size[0] - 1)
    boxes[:, 2].clamp_(0, img_size[0] - 1)
    boxes[:, 1].clamp_(0, img_size[1] - 1)
    boxes[:, 3].clamp_(0, img_size[1] - 1)
    if min_size > 0:            # (sides are >= 0 + ... by construction: with the shipped min_size = 0 nothing is dropped)
        ws, hs = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
        ok = (ws >= min_size) & (hs >= min_size)
        boxes, score = boxes[ok], score[ok]
    # top-k hands the candidates over in descending score order, and only the first post_nms_top_n survivors are used:
    # no second sort, and the suppression stops at the last one it needs (a few hundred of the 6000 candidates in)
    if _nms_takes_max_keep(be):
        keep = be.nms(boxes, score, nms_thresh, max_keep=post_nms_top_n, sorted_desc=True)
    else:                        # a backend with the plain contract (tests' oracle backend)
        keep = be.nms(boxes, score, nms_thresh)[:post_nms_top_n]
    return boxes[keep], score[keep]


def rpn_proposals_batch(objectness, box_regression, anchors, img_size, pre_nms_top_n=6000, post_nms_top_n=300, nms_thresh=0.7):
    """rpn_proposals for B images of ONE size at once (the per-image rule of rpn/inference.py:76-123 applied to every image of
    the batch): one top-k over (B, H W A), one decode / clip over all selected candidates, ONE suppression launch pair for the
    B candidate sets (odx_nms_batched_first_f32: a set's walk stops at its post_nms_top_n-th survivor) and ONE host
    synchronisation for the B survivor counts.  Returns [(boxes_b, scores_b)] — per image what rpn_proposals returns."""
    be = _backend.get_backend()
    B, A, H, W = objectness.shape
    k = min(pre_nms_top_n, A * H * W)
    if hasattr(be, "rpn_topk_decode") and objectness.is_cuda and k <= 8192:
        # top-k, sorting, delta gather, decoding and clipping of the whole batch as ONE launch (odx_rpn_topk_decode_f32)
        boxes, score, _ = be.rpn_topk_decode(objectness, box_regression, anchors, k, img_size, DELTA_CLAMP)
        counts = torch.full((B,), k, dtype=torch.int32, device=boxes.device)
        keep = be.nms_batched(boxes, counts, nms_thresh, max_keep=post_nms_top_n)
        n = keep.sum(dim=1).tolist()                      # the batch's one host synchronisation
        return list(zip(boxes[keep].split(n), score[keep].split(n)))
    obj = objectness.permute(0, 2, 3, 1).reshape(B, -1).sigmoid()
    reg = box_regression.view(B, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(B, -1, 4)
    score, idx = obj.topk(k, dim=1, sorted=True)
    sel = reg.gather(1, idx.unsqueeze(2).expand(B, k, 4)).reshape(B * k, 4)
    boxes = decode_deltas(sel, anchors.to(reg.device)[idx.reshape(-1)])
    boxes[:, 0].clamp_(0, img_size[0] - 1)
    boxes[:, 2].clamp_(0, img_size[0] - 1)
    boxes[:, 1].clamp_(0, img_size[1] - 1)
    boxes[:, 3].clamp_(0, img_size[1] - 1)
    boxes = boxes.view(B, k, 4)
    counts = torch.full((B,), k, dtype=torch.int32, device=boxes.device)
    keep = be.nms_batched(boxes, counts, nms_thresh, max_keep=post_nms_top_n)
    n = keep.sum(dim=1).tolist()                          # the batch's one host synchronisation
    kb, ks = boxes[keep].split(n), score[keep].split(n)   # image after image, descending score inside an image
    return list(zip(kb, ks))


# ---------------------------------------------------------------------------- image pre-processing
PIXEL_MEAN_BGR255 = (102.9801, 115.9465, 122.7717)     # cfg.INPUT.PIXEL_MEAN of the shipped configs (Detectron BGR-255)


def preprocess_image(image, min_size=600, pixel_mean=PIXEL_MEAN_BGR255, pixel_std=(1.0, 1.0, 1.0), to_bgr255=True):
    """The reference's test/harvest-time transform (engine/feature_proposal_extractor.py:86-113: ToPILImage ->
    Resize(MIN_SIZE_TEST) -> ToTensor -> x 255 (TO_BGR255) -> Normalize(PIXEL_MEAN, PIXEL_STD)) as one op sequence on the
    image's device.  image: (H, W, 3) uint8 in BGR order, as the reference hands it over (inference.py:231-234).
    Resize(int) brings the SHORTER side to min_size keeping the aspect ratio (the longer one truncated, torchvision's
    rule) with PIL's bilinear filter, i.e. a triangle filter whose support grows with the down-scaling factor
    (antialias=True below is that filter) and a result rounded back to 8 bits.  Returns (1, 3, H', W') f32 and the
    (width, height) it was resized to.  PARITY UNPINNED by the reference (PIL / torchvision are not importable here);
    tests/test_extract.py restates the filter in numpy."""
    if image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("preprocess_image: expected an (H, W, 3) image, got %s" % (tuple(image.shape),))
    H, W = int(image.shape[0]), int(image.shape[1])
    if H <= W:
        nh, nw = int(min_size), int(min_size * W / H)
    else:
        nh, nw = int(min_size * H / W), int(min_size)
    x = image.permute(2, 0, 1).unsqueeze(0).to(torch.float32)
    if (nh, nw) != (H, W):
        x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)
        x = torch.floor(x + 0.5).clamp_(0, 255)                     # PIL stores the resized image as 8-bit
    if not to_bgr255:
        x = x[:, [2, 1, 0]] / 255.0
    mean = torch.tensor(pixel_mean, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(pixel_std, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std, (nw, nh)


# ---------------------------------------------------------------------------- reference checkpoints
_REF_KEY_RULES = (("backbone.body.stem.", "backbone."), ("backbone.body.", "backbone."),
                  ("rpn.head.conv.", "rpn_conv."), ("rpn.head.cls_logits.", "rpn_logits."), ("rpn.head.bbox_pred.", "rpn_deltas."),
                  ("roi_heads.box.feature_extractor.head.", "head."), ("roi_heads.mask.predictor.conv5_mask.", "conv5_mask."))
_REF_KEY_IGNORED = ("roi_heads.box.predictor.", "roi_heads.mask.predictor.mask_fcn_logits.", "roi_heads.mask.feature_extractor.",
                    "rpn.anchor_generator.")


def remap_reference_state_dict(state_dict):
    """Parameter names of a maskrcnn_benchmark / Detectron R-50-C4 Mask R-CNN checkpoint (the reference's
    `feature_extractor_*.pth` and `e2e_mask_rcnn_R-50-C4_1x`, loaded by DetectronCheckpointer at
    extract_features_detector.py:150-170) -> this module's names.  Accepts the bare state dict or the checkpoint dict
    with its "model" entry, with or without DistributedDataParallel's "module." prefix.  Returns (mapped, ignored, unknown):
    `ignored` are entries this pipeline replaces by the on-line heads (the SGD-trained class / box / mask predictors, the
    shared mask feature extractor, anchor buffers); `unknown` are names no rule covers — callers should treat a non-empty
    list as an error.  The names are those of maskrcnn_benchmark's ResNet / RPNHead / ResNet50Conv5ROIFeatureExtractor /
    MaskRCNNC4Predictor modules (not vendored here: restated, PARITY UNPINNED)."""
    sd = state_dict.get("model", state_dict) if isinstance(state_dict, dict) else state_dict
    mapped, ignored, unknown = {}, [], []
    for key, val in sd.items():
        k = key[len("module."):] if key.startswith("module.") else key
        if any(k.startswith(p) for p in _REF_KEY_IGNORED):
            ignored.append(key)
            continue
        for src, dst in _REF_KEY_RULES:
            if k.startswith(src):
                mapped[dst + k[len(src):].replace(".downsample.", ".down.")] = val
                break
        else:
            unknown.append(key)
    return mapped, ignored, unknown


def load_reference_checkpoint(model, state_dict, strict=True):
    """Load a reference-named checkpoint into an OnlineDetectionModel; raises when names are left over on either side."""
    mapped, ignored, unknown = remap_reference_state_dict(state_dict)
    res = model.load_state_dict(mapped, strict=False)
    if strict and (unknown or res.missing_keys or res.unexpected_keys):
        raise KeyError("reference checkpoint does not match the model: unknown %s, missing %s, unexpected %s"
                       % (unknown[:5], list(res.missing_keys)[:5], list(res.unexpected_keys)[:5]))
    return ignored


# ---------------------------------------------------------------------------- network
class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def forward(self, x):
        scale = self.weight * self.running_var.rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class _FoldedBN(nn.Module):
    """Frozen batch norm is an affine map per channel: y = conv(x) * scale + shift with constants fixed at load time, so
    it is folded into the convolution it follows (w * scale, bias = shift) — one MIOpen call per layer instead of the
    convolution plus six elementwise kernels over the whole activation that evaluating rsqrt / scale / shift per
    forward costs (53 layers per image).  The folded tensors are derived data: rebuilt after any move (`_apply`) or
    state-dict load; parameters and buffers keep the reference's names and values."""

    def __init__(self):
        super().__init__()
        self._folded = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._folded.clear())

    def _apply(self, fn, *a, **kw):
        self._folded.clear()
        return super()._apply(fn, *a, **kw)

    def _fold(self, key, conv, bn, x):
        """(weight, bias) of conv followed by the frozen batch norm, in the dtype the convolution will run in: under
        autocast the folded tensors are kept in the autocast dtype too — autocast's own cast of an f32 weight is redone
        on every call (53 weight tensors per image; the bf16 trunk was slower than the f32 one for it)."""
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
        elif x.dtype in (torch.bfloat16, torch.float16):
            dt = x.dtype            # a trunk run natively in a 16-bit dtype (no autocast context: its per-op dispatch is host time)
        else:
            dt = conv.weight.dtype
        wb = self._folded.get((key, dt))
        if wb is None:
            with torch.no_grad():
                scale = bn.weight * bn.running_var.rsqrt()
                wb = ((conv.weight * scale.view(-1, 1, 1, 1)).to(dt).contiguous(), (bn.bias - bn.running_mean * scale).to(dt).contiguous())
            self._folded[(key, dt)] = wb
        return wb

    def conv_bn(self, name_conv, name_bn, x):
        conv, bn = getattr(self, name_conv), getattr(self, name_bn)
        wb = self._fold(name_conv, conv, bn, x)
        return F.conv2d(x, wb[0], wb[1], conv.stride, conv.padding)

    def conv_bn_act(self, name_conv, name_bn, x, residual=None):
        """relu(conv_bn(x) (+ residual)).  On the GPU (no autocast, no gradient): the convolution without its bias and
        bias + residual + ReLU as ONE in-place pass (HipBackend.bias_act_) — the library runs the bias as a kernel of its own
        and the addition and the ReLU are two more, ~110 of the C4 trunk's 275 launches per image; same additions in the
        same order, so the same bits."""
        conv, bn = getattr(self, name_conv), getattr(self, name_bn)
        wb = self._fold(name_conv, conv, bn, x)
        if (_fused_epilogue(x, wb[0]) and (residual is None or (residual.dtype == x.dtype and residual.is_contiguous()))):
            y = F.conv2d(x, wb[0], None, conv.stride, conv.padding)
            if y.is_contiguous():
                return _backend.get_backend().bias_act_(y, wb[1], residual, relu=True)
            y = y + wb[1].view(1, -1, 1, 1)
        else:
            y = F.conv2d(x, wb[0], wb[1], conv.stride, conv.padding)
        return F.relu(y if residual is None else y + residual)


def _stem_rows(net, x):
    """The stem of a ResNet trunk (net.conv1 / net.bn1, 3 x 3 / 2 max pooling) for the row-GEMM stages: the 7 x 7 convolution by the
    convolution library WITHOUT its bias, then bias + ReLU + pooling + the change to NHWC rows + the maximum the first packing
    needs as ONE pass (HipBackend.stem_pool_rows) -> what _stages_rows / _stages_rows16 take for `y`: (rows, (B, h, w), meta).
    Where that pass does not apply (the CPU, gradients, operand types that differ): the pooled (B, C, h, w) map as before."""
    be = _backend.get_backend()
    dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
    if dt in (torch.bfloat16, torch.float16) and x.dtype != dt and x.is_cuda and not torch.is_grad_enabled():
        x = x.to(dt)                                     # (autocast's own cast of the image, made once here)
    conv, bn = net.conv1, net.bn1
    wb = net._fold("conv1", conv, bn, x)
    if _fused_epilogue(x, wb[0]) and hasattr(be, "stem_pool_rows"):
        y = F.conv2d(x, wb[0], None, conv.stride, conv.padding)
        if y.is_contiguous():
            return be.stem_pool_rows(y, wb[1])
    return F.max_pool2d(net.conv_bn_act("conv1", "bn1", x), 3, 2, 1)


def _fused_epilogue(x, w):
    """The trunk's bias / residual / ReLU epilogue runs as one HIP pass (HipBackend.bias_act_): GPU maps in the dtype the
    convolution computes in (f32, or a trunk run natively in bf16 / f16; under autocast the operands differ and the library
    path stays), no gradient."""
    return (x.is_cuda and x.dtype == w.dtype and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and not torch.is_grad_enabled())


class Bottleneck(_FoldedBN):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.down = None
        if cin != cout or stride != 1:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), FrozenBatchNorm2d(cout))
        self.conv1, self.bn1 = nn.Conv2d(cin, mid, 1, stride, bias=False), FrozenBatchNorm2d(mid)   # stride in the 1x1
        self.conv2, self.bn2 = nn.Conv2d(mid, mid, 3, 1, 1, bias=False), FrozenBatchNorm2d(mid)
        self.conv3, self.bn3 = nn.Conv2d(mid, cout, 1, bias=False), FrozenBatchNorm2d(cout)

    def forward(self, x):
        if self.down is None:
            idn = x
        else:
            wb = self._fold("down", self.down[0], self.down[1], x)
            if _fused_epilogue(x, wb[0]):
                idn = F.conv2d(x, wb[0], None, self.down[0].stride, self.down[0].padding)
                idn = (_backend.get_backend().bias_act_(idn, wb[1], None, relu=False) if idn.is_contiguous()
                       else (idn + wb[1].view(1, -1, 1, 1)).contiguous())
            else:
                idn = F.conv2d(x, wb[0], wb[1], self.down[0].stride, self.down[0].padding)
        y = self.conv_bn_act("conv1", "bn1", x)
        y = self.conv_bn_act("conv2", "bn2", y)
        return self.conv_bn_act("conv3", "bn3", y, residual=idn)


def _bottleneck_rows_b16(be, blk, x, R, H, W):
    """_bottleneck_rows for a forward run in bf16 / f16 (compute_dtype): x is a backend.Rows16; the folded weights rounded once
    to the 16-bit type, every product on the matrix cores with ONE MFMA term (odx_gemm_b16), sums, bias and identity in f32,
    one rounding per layer output; the 9-tap gather in the 16-bit type (odx_taps3x3_16).  No vendor GEMM."""
    dt = x.buf.dtype

    def wrows(key, conv, bn, taps=False):
        c = blk._folded.get((key + "/b16", dt))
        if c is None:
            w, b = blk._fold(key, conv, bn, x.buf)
            w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1) if taps else w.reshape(w.shape[0], -1)     # (out, ky kx in)
            c = blk._folded[(key + "/b16", dt)] = (be.rows16(w.to(dt).contiguous(), dt), b.float().contiguous())
        return c
    if blk.down is None:
        idn = x
    else:
        wd, bd = wrows("down", blk.down[0], blk.down[1])
        idn = be.gemm_b16(x, wd, bias=bd)
    w1, b1 = wrows("conv1", blk.conv1, blk.bn1)
    y = be.gemm_b16(x, w1, bias=b1, relu=True, zero_row=True)
    w2, b2 = wrows("conv2", blk.conv2, blk.bn2, taps=True)
    y = be.conv3x3_rows16(y, R, H, W, w2, bias=b2, relu=True)             # (taps gathered inside the product where it can)
    w3, b3 = wrows("conv3", blk.conv3, blk.bn3)
    return be.gemm_b16(y, w3, bias=b3, residual=idn, relu=True)


def _bottleneck_rows_h2(be, blk, x, R, H, W, x_meta=None, want_max=False, pack_out=False):
    """The f32 form of _bottleneck_rows on the split-f16 tile cores (odx_gemm_h2_f32: f32 accuracy on the f16 matrix cores,
    243-317 TF on these shapes against 95-117 TF for the f32 MFMA GEMMs) as a CHAIN of layers: folded weights packed once per
    layer; bias / identity / ReLU in the GEMM epilogues; and every layer writes its output as the next layer's packed operand
    straight from its accumulators (odx_gemm_h2_chain_f32) — the two inner activations exist in packed form only, the 3 x 3
    convolution gathers its taps inside its operand loads where the library can (odx_gemm_h2_taps_f32) or from the packed rows,
    and only the block's input is ever packed by a pass of its own (when it did not come out of such a layer).
    x: the block's input — f32 rows (x_meta: the meta words its producer left, its maximum in [1]) or the backend.PackedRows a
    previous block returned with pack_out (its f32 rows serve the identity branch).  Returns the output rows; with want_max
    (out, meta); with pack_out a PackedRows holding both forms, for the next block."""
    from .backend import PackedRows

    def wpack(key, conv, bn, taps=False):
        c = blk._folded.get((key + "/h2", torch.float32))
        if c is None:
            w, b = blk._fold(key, conv, bn, x.X if isinstance(x, PackedRows) else x)
            w = (w.permute(0, 2, 3, 1).reshape(w.shape[0], -1) if taps else w.reshape(w.shape[0], -1)).float().contiguous()   # (out, ky kx in)
            c = blk._folded[(key + "/h2", torch.float32)] = (be.packed(w), b.float().contiguous(), be.weight_bounds(w, b))
        return c
    if isinstance(x, PackedRows):
        xp, xf, xm = x, x.X, x.meta
    else:
        xp = be.packed(x, meta=x_meta)
        xf, xm = x, xp.meta
    if blk.down is None:
        idn, im = xf, xm
    else:
        wd, bd, _ = wpack("down", blk.down[0], blk.down[1])
        idn, im = be.gemm_h2(xp, wd, bias=bd, with_max=True)
    w1, b1, g1 = wpack("conv1", blk.conv1, blk.bn1)
    y = be.chain_gemm(xp, w1, bias=b1, relu=True, bounds=g1, f32_out=False, zero_row=True)
    w2, b2, g2 = wpack("conv2", blk.conv2, blk.bn2, taps=True)
    y = be.chain_conv3x3(y, R, H, W, w2, bias=b2, relu=True, bounds=g2, f32_out=False)
    w3, b3, g3 = wpack("conv3", blk.conv3, blk.bn3)
    if pack_out:
        return be.chain_gemm(y, w3, bias=b3, residual=idn, residual_meta=im, relu=True, bounds=g3, f32_out=True)
    return be.gemm_h2(y, w3, bias=b3, residual=idn, relu=True, with_max=want_max)


def _bottleneck_rows(blk, x, R, H, W):
    """One Bottleneck on activations kept as a (R * H * W, C) row matrix (NHWC): the 1 x 1 convolutions ARE matrix products
    over those rows and the 3 x 3 one is a product over a 9-tap gather of them, so the block is three (four with the
    projection) GEMMs with the folded batch-norm bias added by the GEMM, instead of library convolutions on 7 x 7 maps.  A
    stride (always in the 1 x 1 convolutions here, STRIDE_IN_1X1) is applied by the caller: x holds the rows of the
    positions that survive it.  On the GPU the products run on this library's tile cores (f32: _bottleneck_rows_h2; a 16-bit
    forward: _bottleneck_rows_b16 through Conv5Head.forward_rows); the plain-torch form below is the CPU statement of the
    same block (tests' oracle backend)."""
    if x.is_cuda:
        be = _backend.get_backend()
        if x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda") and hasattr(be, "gemm_h2"):
            return _bottleneck_rows_h2(be, blk, x, R, H, W)
        if hasattr(be, "gemm_b16"):
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
            if dt in (torch.bfloat16, torch.float16):
                return _bottleneck_rows_b16(be, blk, be.rows16(x, dt), R, H, W).dense
        raise RuntimeError("conv5 head: no tile core for rows of type %s on this backend" % x.dtype)
    dt = x.dtype

    def mat(key, conv, bn, taps=False):
        c = blk._folded.get((key + "/rows", dt))
        if c is None:
            w, b = blk._fold(key, conv, bn, x)
            w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1) if taps else w.reshape(w.shape[0], -1)     # (out, ky kx in)
            c = blk._folded[(key + "/rows", dt)] = (w.t().contiguous(), b)
        return c
    if blk.down is None:
        idn = x
    else:
        wd, bd = mat("down", blk.down[0], blk.down[1])
        idn = x @ wd + bd
    w1, b1 = mat("conv1", blk.conv1, blk.bn1)
    y = F.relu(x @ w1 + b1)
    mid = y.shape[1]
    yp = F.pad(y.view(R, H, W, mid), (0, 0, 1, 1, 1, 1))
    cols = torch.cat([yp[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], dim=3).view(R * H * W, 9 * mid)
    w2, b2 = mat("conv2", blk.conv2, blk.bn2, taps=True)
    y = F.relu(cols @ w2 + b2)
    w3, b3 = mat("conv3", blk.conv3, blk.bn3)
    return F.relu(y @ w3 + b3 + idn)


def _stages_rows_form(stages):
    """True when these ResNet stages can run as row GEMMs: strides only in the 1 x 1 convolutions that open a block
    (STRIDE_IN_1X1, the projection beside them with the same stride), 3 x 3 convolutions of stride 1 / padding 1, channels in
    eights."""
    for stage in stages:
        for blk in stage:
            st = blk.conv1.stride[0]
            if (blk.conv2.stride[0] != 1 or tuple(blk.conv2.padding) != (1, 1) or blk.conv3.stride[0] != 1
                    or (blk.down is not None and blk.down[0].stride[0] != st) or (blk.down is None and st != 1)
                    or blk.conv1.in_channels % 8 or blk.conv2.in_channels % 8 or blk.conv3.out_channels % 8):
                return False
    return True


def _stages_rows(be, y, stages, pack_last=False):
    """The stages of a ResNet trunk as one chain of row GEMMs (_bottleneck_rows_h2) behind the stem's output y (B, C, H, W):
    y becomes NHWC rows once; a block's stride is applied to its input rows (the positions a strided 1 x 1 convolution
    reads).  Returns, per stage, (out, (B, H, W)): out = the stage's last block's output — backend.PackedRows (f32 rows .X and
    the packed operand) for every stage but the last, whose output is plain f32 rows unless pack_last."""
    if isinstance(y, tuple):                                    # (_stem_rows: already rows, with the maximum of the matrix)
        x, (B, H, W), meta = y
        outs = []
    else:
        B, C, H, W = y.shape
        x, meta, outs = y.permute(0, 2, 3, 1).reshape(B * H * W, C), None, []
    blocks = [(si, blk) for si, stage in enumerate(stages) for blk in stage]
    for k, (si, blk) in enumerate(blocks):
        st = blk.conv1.stride[0]
        if st > 1:                                              # the rows a strided 1 x 1 convolution reads, packed anew
            # (a COPY of the producer's meta words: packing the cut rows writes its own scale into them, and the stage's packed
            # output — the pyramid's lateral convolution reads it — keeps the scale it was written with)
            rows, meta = (x.X, x.meta.clone()) if hasattr(x, "P") else (x, meta)
            cut = rows.view(B, H, W, -1)[:, ::st, ::st]
            H, W = cut.shape[1], cut.shape[2]
            x = cut.reshape(B * H * W, -1)                      # (the maximum of the whole matrix bounds that of the cut rows)
        last = k + 1 == len(blocks)
        x = _bottleneck_rows_h2(be, blk, x, B, H, W, x_meta=meta, pack_out=pack_last or not last)
        if last or blocks[k + 1][0] != si:
            outs.append((x, (B, H, W)))
    return outs


def _stages_rows16(be, y, stages):
    """_stages_rows for a forward run in bf16 / f16: y (B, C, H, W) the stem's output in the 16-bit type; every bottleneck on
    16-bit rows (_bottleneck_rows_b16: one MFMA term per product, f32 sums, one rounding per layer — no packing, no scales).
    Returns, per stage, (backend.Rows16, (B, H, W))."""
    if isinstance(y, tuple):                                    # (_stem_rows: Rows16 already)
        x, (B, H, W), _ = y
    else:
        B, C, H, W = y.shape
        x = be.rows16(y.permute(0, 2, 3, 1).reshape(B * H * W, C), y.dtype)
    outs = []
    blocks = [(si, blk) for si, stage in enumerate(stages) for blk in stage]
    for k, (si, blk) in enumerate(blocks):
        st = blk.conv1.stride[0]
        if st > 1:
            cut = x.buf.view(B, H, W, -1)[:, ::st, ::st]
            H, W = cut.shape[1], cut.shape[2]
            x = type(x)(cut.reshape(B * H * W, -1), x.K)
        x = _bottleneck_rows_b16(be, blk, x, B, H, W)
        if k + 1 == len(blocks) or blocks[k + 1][0] != si:
            outs.append((x, (B, H, W)))
    return outs


def _stage(cin, mid, cout, blocks, stride):
    layers = [Bottleneck(cin, mid, cout, stride)] + [Bottleneck(cout, mid, cout, 1) for _ in range(blocks - 1)]
    return nn.Sequential(*layers)


class ResNet50C4(_FoldedBN):
    out_channels = 1024

    def __init__(self, width=64):
        super().__init__()
        w = width
        self.conv1, self.bn1 = nn.Conv2d(3, w, 7, 2, 3, bias=False), FrozenBatchNorm2d(w)
        self.layer1 = _stage(w, w, 4 * w, 3, 1)
        self.layer2 = _stage(4 * w, 2 * w, 8 * w, 4, 2)
        self.layer3 = _stage(8 * w, 4 * w, 16 * w, 6, 2)
        self.out_channels = 16 * w

    def forward(self, x):
        x = F.max_pool2d(self.conv_bn_act("conv1", "bn1", x), 3, 2, 1)
        return self.layer3(self.layer2(self.layer1(x)))

    def rows_form(self):
        return _stages_rows_form((self.layer1, self.layer2, self.layer3))

    def forward_rows16(self, x):
        """forward_rows for a forward run in bf16 / f16 (x: the image, under the caller's autocast): the stem by the convolution
        library in the 16-bit type, the three stages on 16-bit rows (_stages_rows16).  Returns (backend.Rows16, (B, h, w))."""
        be = _backend.get_backend()
        return _stages_rows16(be, _stem_rows(self, x), (self.layer1, self.layer2, self.layer3))[-1]

    def forward_rows(self, x):
        """The trunk on this library's tile cores (f32 on the GPU): the stem (7 x 7 convolution, max pooling) by the convolution
        library, its output turned ONCE into NHWC rows, and every bottleneck of the three stages as row GEMMs on the split-f16
        matrix-core path (_stages_rows / _bottleneck_rows_h2: f32 accuracy, bias / identity / ReLU in the GEMM epilogues, the
        3 x 3 convolutions over a 9-tap gather, every layer writing the next one's packed operand) — 60 GFLOP per 600 x 800
        image that the library ran as f32 Winograd / GEMM convolutions plus an epilogue pass per layer.  Returns (rows
        (B h w, C) f32, (B, h, w)); the map as the callers know it is rows.view(B, h, w, C).permute(0, 3, 1, 2) — a
        channels-last view, no copy."""
        be = _backend.get_backend()
        return _stages_rows(be, _stem_rows(self, x), (self.layer1, self.layer2, self.layer3))[-1]


class Conv5Head(nn.Module):
    """ResNet stage 5 on 14 x 14 RoI crops -> 7 x 7 x 2048 (ResNet50Conv5ROIFeatureExtractor.head)."""

    def __init__(self, cin=1024):
        super().__init__()
        self.layer4 = _stage(cin, cin // 2, 2 * cin, 3, 2)
        self.out_channels = 2 * cin

    def forward_conv(self, x):
        """The stage as convolutions (what `forward` computes, kept as the reference of its test)."""
        return self.layer4(x)

    def rows_form(self):
        """The stride of the first block's 1 x 1 convolutions when the stage can run as row GEMMs (strides only there), else 0."""
        blocks = list(self.layer4)
        if any(b.conv1.stride[0] != 1 for b in blocks[1:]) or any(b.conv2.stride[0] != 1 for b in blocks):
            return 0
        return int(blocks[0].conv1.stride[0])

    def forward_rows(self, rows, R, H, W):
        """rows (R * H * W, C): the input at the positions the first block's strided 1 x 1 convolutions read (NHWC)."""
        if rows.is_cuda:
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else rows.dtype
            be = _backend.get_backend()
            if dt in (torch.bfloat16, torch.float16) and hasattr(be, "gemm_b16"):
                x = be.rows16(rows, dt)                              # the one cast of the stage's input
                for blk in self.layer4:
                    x = _bottleneck_rows_b16(be, blk, x, R, H, W)    # 16-bit activations from block to block, pad columns zero
                return x.dense.reshape(R, H, W, -1).permute(0, 3, 1, 2)
            if dt == torch.float32 and hasattr(be, "gemm_h2"):
                blocks = list(self.layer4)                           # (a chain: each block hands the next one its packed output)
                for k, blk in enumerate(blocks):
                    rows = _bottleneck_rows_h2(be, blk, rows, R, H, W, pack_out=k + 1 < len(blocks))
                return rows.view(R, H, W, -1).permute(0, 3, 1, 2)
        for blk in self.layer4:
            rows = _bottleneck_rows(blk, rows, R, H, W)
        return rows.view(R, H, W, -1).permute(0, 3, 1, 2)          # (R, 2048, H, W) as a view of the NHWC rows

    def forward(self, x):
        st = self.rows_form()
        if st == 0 or x.shape[0] == 0:
            return self.layer4(x)
        xs = x[:, :, ::st, ::st]                                   # the positions a stride-`st` 1 x 1 convolution reads
        R, C, H, W = xs.shape
        return self.forward_rows(xs.permute(0, 2, 3, 1).reshape(R * H * W, C), R, H, W)


class GraphedCall:
    """fn(x) replayed from a captured HIP graph, one graph per input (shape, dtype).

    At batch 1 the frozen trunk is 120-170 dependent launches of 5-60 us: the host queues them barely faster than the GPU runs
    them, and every gap is idle chip (R-50-C4 at 600 x 800: 2.06 ms launch by launch, 1.26 ms replayed; same kernels, same
    order).  The graph owns its input and output buffers: the caller's tensor is copied in, the results are handed out as
    copies, so nothing the caller holds is overwritten by the next replay.  Shapes are data here (aspect ratios differ between
    images): at most `max_graphs` are kept, least recently used first out.  Whoever changes the weights behind fn calls
    clear().  A capture that fails (an operation the runtime cannot capture) turns the wrapper into a plain call for good;
    odx.options trunk_graph=False does the same for every wrapper made afterwards."""

    import threading as _threading
    _tls = _threading.local()                  # .forbid = True: no capture from this thread (a thread that shares the device with
    # another thread's launches — the harvest loop's forward thread; captures are made on the caller's thread beforehand)

    def __init__(self, fn, max_graphs=6):
        import threading
        self.fn, self.max_graphs = fn, max_graphs
        self.graphs = {}
        self.captures = {}                     # key -> times captured (see __call__)
        self.seen = {}                         # shape -> calls so far: a shape is captured at its SECOND call (a stream of images
        # that all differ in size — capture costs three forwards — then simply runs launch by launch)
        self.enabled = bool(_options.load().trunk_graph)
        # (re-entrant: fn itself may call clear() — a weight pack remade inside the forward drops the graphs that point at the
        # old one, OnlineDetectionModel._wpack)
        self.lock = threading.RLock()          # the extractor's forward thread and the caller's may both come through here

    def clear(self):
        with self.lock:
            self.graphs.clear()                 # (the shapes seen so far stay seen: the next call of one captures again)
            self.captures.clear()

    def __getstate__(self):                     # a copied / pickled model starts without graphs (they are tied to this process's buffers)
        return {"fn": self.fn, "max_graphs": self.max_graphs, "enabled": self.enabled}

    def __setstate__(self, d):
        import threading
        self.fn, self.max_graphs, self.enabled = d["fn"], d["max_graphs"], d["enabled"]
        self.graphs, self.seen, self.captures, self.lock = {}, {}, {}, threading.RLock()

    def _capture(self, xs, gstream):
        with torch.cuda.stream(gstream):
            static_in = tuple(x.clone() for x in xs)
            side = torch.cuda.Stream()
            side.wait_stream(gstream)
            with torch.cuda.stream(side):       # the library picks its algorithms and the caches fill outside the capture
                for _ in range(2):
                    self.fn(*static_in)
            gstream.wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            be = _backend.get_backend()
            reset = getattr(be, "reset_capture_pools", None)          # memory pooled during a capture belongs to that graph alone
            if reset is not None:
                reset()
            try:
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):     # (the extractor's other thread keeps allocating)
                    static_out = self.fn(*static_in)
            finally:
                if reset is not None:
                    reset()
        return graph, static_in, static_out, gstream

    def __call__(self, x, *more, key_extra=None, capture=True):
        """fn(x, *more): every argument a tensor of fixed shape per key (copied into the graph's own inputs); key_extra:
        whatever else the captured work depends on (hashable); capture=False: replay a graph that exists, never make one in
        this call.

        A graph is ALWAYS launched on one stream of its own (made with the capture), whatever stream the caller is on: the caller's
        stream and the graph's are joined by events before the inputs are copied in and behind the copies out.  Launching one
        graph from changing streams is not safe on this runtime: the first launch from another stream with new inputs returned
        the previous inputs' results (or garbage — a survivor count read as 10^5 was the harvest loop's memory fault), on the
        legacy default stream and on side streams alike (round 5; the matrix that showed it: capture on A, replay on A / B /
        A ...)."""
        xs = (x,) + tuple(more)
        if not (self.enabled and x.is_cuda and not torch.is_grad_enabled()):
            return self.fn(*xs)
        key = (tuple((tuple(t.shape), t.dtype) for t in xs), x.device.index, torch.is_autocast_enabled("cuda"),
               torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None, key_extra)
        with self.lock:
            entry = self.graphs.pop(key, None)
            if entry is None:
                n = self.seen.get(key, 0)
                if not capture or getattr(self._tls, "forbid", False):
                    return self.fn(*xs)
                if n < 1:
                    if len(self.seen) > 256:
                        self.seen.clear()
                    self.seen[key] = n + 1
                    return self.fn(*xs)
                if self.captures.get(key, 0) >= 2:    # captured, evicted, captured, evicted: this stream of shapes cycles through
                    return self.fn(*xs)               # more of them than are kept — a capture costs ~4 plain calls, stop paying it
                self.captures[key] = self.captures.get(key, 0) + 1
                cur = torch.cuda.current_stream()
                # the graph's launch stream: the LAST of the process's measured side streams (odx/streams.py: on a hardware queue
                # of its own, away from the chains' streams, which take the first ones) — a plain pool stream made here, behind
                # whatever the process ran before, could share the caller's queue, and a group's forward then queues in front of
                # the harvest's small kernels (4.5 instead of 3.6 ms per image behind the headline job)
                from . import streams as _streams
                own = _streams.of_default(3, x.device)
                gstream = own[-1] if own and own[-1].cuda_stream != cur.cuda_stream else torch.cuda.Stream()
                gstream.wait_stream(cur)
                try:
                    entry = self._capture(xs, gstream)
                except Exception as e:          # noqa: BLE001 — whatever the runtime refuses: the plain call is always right
                    self.enabled = False
                    print("odx: HIP graph capture of %s failed (%s: %s); running it launch by launch" % (
                        getattr(self.fn, "__name__", "the forward"), type(e).__name__, e), file=sys.stderr)
                    return self.fn(*xs)
                cur.wait_stream(gstream)
                while len(self.graphs) >= self.max_graphs:
                    self.graphs.pop(next(iter(self.graphs)))
            self.graphs[key] = entry             # (re-inserted last: most recently used)
            graph, static_in, static_out, gstream = entry
            cur = torch.cuda.current_stream()
            gstream.wait_stream(cur)             # the caller's inputs are ready; (earlier users' copies out are on gstream itself)
            with torch.cuda.stream(gstream):
                for dst, src in zip(static_in, xs):
                    dst.copy_(src)
                graph.replay()
                out = static_out.clone() if torch.is_tensor(static_out) else tuple(None if t is None else t.clone() for t in static_out)
            cur.wait_stream(gstream)
            for t in ((out,) if torch.is_tensor(out) else out):
                if t is not None:
                    t.record_stream(cur)         # allocated on the graph's stream, used (and later freed) by the caller's
            return out


class OnlineDetectionModel(nn.Module):
    """backbone -> RPN -> (gt boxes prepended) -> RoIAlign -> conv5 head -> avg-pooled features."""

    def __init__(self, width=64, num_anchors=15, pre_nms_top_n=6000, post_nms_top_n=300, rpn_nms=0.7, resolution=14,
                 seed=0, mask_dim=256, compute_dtype=None):
        """compute_dtype: None = f32 convolutions; torch.bfloat16 = the convolutions under bf16 autocast (BASELINE
        config 2's forward; 126 instead of 89 images/s at 600 x 800 with 300 RoIs).  Everything handed on — the C4 map the
        RoIAlign kernel reads, RoI features, RPN rows, mask pixel rows — is f32 either way."""
        super().__init__()
        self.compute_dtype = compute_dtype
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.backbone = ResNet50C4(width)
        C = self.backbone.out_channels
        self.rpn_conv = nn.Conv2d(C, C, 3, 1, 1)
        self.rpn_logits = nn.Conv2d(C, num_anchors, 1)
        self.rpn_deltas = nn.Conv2d(C, 4 * num_anchors, 1)
        for l in (self.rpn_conv, self.rpn_logits, self.rpn_deltas):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        self.head = Conv5Head(C)
        # mask branch: ConvTranspose 2C -> mask_dim, k2 s2 (roi_mask_predictors.py:24): 7x7 -> 14x14 pixel rows
        self.conv5_mask = nn.ConvTranspose2d(2 * C, mask_dim, 2, 2, 0)
        nn.init.kaiming_normal_(self.conv5_mask.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.conv5_mask.bias, 0)
        torch.random.set_rng_state(g)
        self.stride = 16
        self.cells = cell_anchors(self.stride)
        self.pre_nms_top_n, self.post_nms_top_n, self.rpn_nms, self.resolution = pre_nms_top_n, post_nms_top_n, rpn_nms, resolution
        self.online_rpn = None          # odx.heads.OnlineRPNHead once FALKON RPN models exist
        self.online_box = None          # odx.heads.OnlineBoxPredictor
        self.online_mask = None         # odx.heads.OnlineMaskPredictor
        self.mask_dim = mask_dim
        # stride-16 positions (images x h x w) from which the f32 trunk and RPN head run as row GEMMs (_rows_path)
        self.rows_min_positions = int(_options.load().rows_min_positions)
        self._trunk_graphs = GraphedCall(self._c4_eager)
        # The whole group forward from ONE HIP graph (forward_group); odx.options group_graph=False turns it off.  What it took on this
        # runtime (round 5): (i) with the proposal stage as tensor operations (a library top-k, gather, advanced indexing) the
        # graph's second or third replay ended in a GPU memory fault — tools/group_graph_bisect.py localised it to that stage,
        # which is three kernels of this library now (odx_rpn_topk_decode_f32, odx_nms_batched_first_f32, odx_nms_compact_f32);
        # (ii) a graph launched from CHANGING streams returned the previous inputs' results or zeros — GraphedCall launches
        # every graph on one stream of its own; (iii) one sequence still produced all-zero features once — explained later in
        # the round: hipMemsetAsync captured into a graph is not reliably ordered against the kernel nodes around it (the same
        # fault showed as all-zero trunk maps under the threaded harvest loop), and the suppression's two
        # clears were memsets.  Every clear that can be captured is a kernel now; tools/group_graph_soak.py (1500 group
        # forwards on changing images and calling streams with another graph and other work in between, 40 harvest passes,
        # every result checked against the launch-by-launch forward: no difference) is what the default rests on.
        self._group_graphs = GraphedCall(self._group_static, max_graphs=4)
        if not _options.load().group_graph:
            self._group_graphs.enabled = False
        self.register_load_state_dict_post_hook(OnlineDetectionModel._drop_graphs)

    @staticmethod
    def _drop_graphs(module, incompatible):
        module._trunk_graphs.clear()
        module._group_graphs.clear()
        module.__dict__.pop("_mask_pack", None)
        module.__dict__.pop("_wpacks", None)

    def _apply(self, fn, *a, **kw):
        if "_trunk_graphs" in self.__dict__:
            self._trunk_graphs.clear()          # (.to / .cuda / .half: the captured graphs point at the old tensors)
            self._group_graphs.clear()
            self.__dict__.pop("_mask_pack", None)
            self.__dict__.pop("_wpacks", None)
        return super()._apply(fn, *a, **kw)

    @property
    def feat_dim(self):
        return self.head.out_channels

    def _amp(self):
        if self.compute_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast("cuda", dtype=self.compute_dtype)

    def _rows_path(self, x):
        """The f32 forward on the GPU runs as row GEMMs on this library's tile cores from the stem's output on (trunk stages,
        RPN head; the conv5 head always did): odx.options trunk="conv" keeps the convolution library for the trunk and the RPN head."""
        if not (x.is_cuda and self.compute_dtype is None and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and not torch.is_autocast_enabled("cuda") and _options.current().trunk != "conv"):
            return False
        # below two images of 600 x 800 the stage-3 / RPN products (1900 rows per image) leave most of the chip idle on
        # 128 x 128 tiles and the convolution library's kernels are faster (forward 3.85 against 4.34 ms at one image; 3.30
        # against 3.20 ms per image at two, 2.97 against 2.06 at eight)
        positions = x.shape[0] * x.shape[2] * x.shape[3] if x.shape[1] != 3 else x.shape[0] * (-(-x.shape[2] // self.stride)) * (-(-x.shape[3] // self.stride))
        if positions < self.rows_min_positions:
            return False
        return hasattr(_backend.get_backend(), "gemm_h2") and self.backbone.rows_form()

    def _wpack(self, name, params, make):
        """A packed GEMM operand made from weights, remade when one of them is replaced or written in place."""
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.setdefault("_wpacks", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            if hit is not None:
                self._drop_captured()       # a captured forward points at the pack this one replaces: no graph may outlive it
            hit = cache[name] = (key, make())
        return hit[1]

    def _weights_key(self):
        """(storage, in-place version) of EVERY parameter and buffer of the model: part of both graph caches' keys.  A captured
        forward bakes in the trunk, the RPN head's and the conv5 head's packed weights and the mask pack; an optimiser step or a
        `mul_` / `copy_` into any of them under no_grad — none of which passes through load_state_dict or _apply — must not be
        answered from the old capture (round-5 advisor finding: the keys used to cover two trunk weights only).  An edit
        through `.data` moves no version counter and cannot be seen here or by any cache: call refresh_weights() after one."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def refresh_weights(self):
        """Drop everything derived from the weights — folded batch norms, packed GEMM operands, captured graphs — after an
        in-place edit of a FROZEN weight (the trunk and the conv5 head fold their norms once, at load / .to() /
        load_state_dict, and are not version-checked per call; the RPN head's, the mask branch's and the on-line heads'
        operands are)."""
        for m in self.modules():
            if isinstance(m, _FoldedBN):
                m._folded.clear()
        self.__dict__.pop("_mask_pack", None)
        self.__dict__.pop("_wpacks", None)
        self._drop_captured()

    def _drop_captured(self):
        for name in ("_trunk_graphs", "_group_graphs"):
            g = self.__dict__.get(name)
            if g is not None and not torch.cuda.is_current_stream_capturing():
                g.clear()

    def _rows16_path(self, x):
        """_rows_path for a forward run in bf16 / f16 (compute_dtype): trunk stages and RPN head on 16-bit rows (odx_gemm_b16)."""
        if not (x.is_cuda and self.compute_dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled()
                and _options.current().trunk != "conv"):
            return False
        positions = x.shape[0] * x.shape[2] * x.shape[3] if x.shape[1] != 3 else x.shape[0] * (-(-x.shape[2] // self.stride)) * (-(-x.shape[3] // self.stride))
        if positions < self.rows_min_positions:
            return False
        return hasattr(_backend.get_backend(), "conv3x3_rows16") and self.backbone.rows_form()

    def _c4_eager(self, image):
        if self._rows_path(image):
            rows, (B, h, w) = self.backbone.forward_rows(image)
            return rows.view(B, h, w, -1).permute(0, 3, 1, 2)          # (B, C, h, w) as a channels-last view of the NHWC rows
        if self._rows16_path(image):
            with self._amp():
                x16, (B, h, w) = self.backbone.forward_rows16(image)
            return x16.dense.float().view(B, h, w, -1).permute(0, 3, 1, 2)      # handed on in f32, as every map of this model
        with self._amp():
            return self.backbone(image).float()

    @torch.no_grad()
    def c4(self, image):
        """(1, C, H/16, W/16) f32 trunk features; on the GPU replayed from a HIP graph per image size (GraphedCall)."""
        # (every weight's storage and in-place version counter is part of the key: a graph replays the tensors it was captured
        # with — an optimiser step or a copy_ into a weight must not be answered from the old capture)
        return self._trunk_graphs(image, key_extra=(self.compute_dtype, self._weights_key()))

    def update_model(self, models_rpn=None, models_detection=None, models_segmentation=None):
        """Swap trained on-line models into the running pipeline, each a dict {'classifiers', 'regressors', 'stats'}
        (segmentation: no regressors) — the demo's OnlineSegmentationDemo.update_model
        (mrcnn_modified/demo/predictor_online_segmentation.py:404-425).  A head is created on first use and its
        batched-inference caches are dropped whenever its models change, so a class added with one more FALKON + RLS
        fit (append to the lists, call this) is live on the next image without touching the other classes."""
        from . import heads
        for attr, cls, models in (("online_rpn", heads.OnlineRPNHead, models_rpn), ("online_box", heads.OnlineBoxPredictor, models_detection),
                                  ("online_mask", heads.OnlineMaskPredictor, models_segmentation)):
            if not models:
                continue
            if getattr(self, attr) is None:
                setattr(self, attr, cls())
            getattr(self, attr).set_models(models["classifiers"], models.get("regressors"), models["stats"])

    def grid_anchors(self, h, w, device):
        """grid_anchors(h, w, stride, cells) on `device`, kept per map size: the cell anchors are a host tensor, and a host ->
        device copy per image waits for whatever the GPU is running (harvest.to_device)."""
        cache = self.__dict__.setdefault("_anchor_cache", {})
        key = (int(h), int(w), str(device))
        hit = cache.get(key)
        if hit is None:
            if len(cache) > 16:
                cache.clear()
            hit = cache[key] = grid_anchors(h, w, self.stride, self.cells.to(device))
        return hit

    def rpn_activation(self, c4):
        cv = self.rpn_conv
        if (self._rows_path(c4) and c4.shape[1] % 8 == 0 and tuple(cv.kernel_size) == (3, 3) and tuple(cv.stride) == (1, 1)
                and tuple(cv.padding) == (1, 1)):
            # RPNHead.conv (rpn.py:164-170) as ONE product over the 9-tap gather of the map's NHWC rows (a view when the trunk
            # ran as row GEMMs), bias + ReLU in its epilogue: 36 of the forward's 96 trunk-side GFLOP at 600 x 800
            be = _backend.get_backend()
            B, C, h, w = c4.shape
            rows = c4.permute(0, 2, 3, 1).reshape(B * h * w, C)
            wp, b = self._wpack("rpn_conv", (cv.weight, cv.bias), lambda: (
                be.packed(cv.weight.detach().float().permute(0, 2, 3, 1).reshape(cv.out_channels, -1).contiguous()),
                cv.bias.detach().float().contiguous()))
            t = be.conv3x3_rows(rows, B, h, w, wp, bias=b, relu=True)
            return t.view(B, h, w, -1).permute(0, 3, 1, 2)
        if (self._rows16_path(c4) and c4.shape[1] % 8 == 0 and tuple(cv.kernel_size) == (3, 3) and tuple(cv.stride) == (1, 1)
                and tuple(cv.padding) == (1, 1)):
            be, dt = _backend.get_backend(), self.compute_dtype
            B, C, h, w = c4.shape
            x16 = be.rows16(c4.permute(0, 2, 3, 1).reshape(B * h * w, C), dt, zero_row=True)
            wp, b = self._wpack(("rpn_conv", dt), (cv.weight, cv.bias), lambda: (
                be.rows16(cv.weight.detach().permute(0, 2, 3, 1).reshape(cv.out_channels, -1).to(dt).contiguous(), dt),
                cv.bias.detach().float().contiguous()))
            t16 = be.conv3x3_rows16(x16, B, h, w, wp, bias=b, relu=True)
            return t16.dense.float().view(B, h, w, -1).permute(0, 3, 1, 2)
        with self._amp():
            return F.relu(self.rpn_conv(c4)).float()

    def rpn_outputs(self, t):
        """(objectness logits (B, A, h, w), box deltas (B, 4 A, h, w)) of the RPN's two 1 x 1 convolutions, f32 contiguous."""
        lg, dl = self.rpn_logits, self.rpn_deltas
        if self._rows_path(t) and tuple(lg.kernel_size) == (1, 1) and tuple(dl.kernel_size) == (1, 1):
            be = _backend.get_backend()                 # both as ONE product over the rows (A + 4 A output columns)
            B, C, h, w = t.shape
            wp, b = self._wpack("rpn_out", (lg.weight, lg.bias, dl.weight, dl.bias), lambda: (
                be.packed(torch.cat((lg.weight.detach().float().reshape(lg.out_channels, -1), dl.weight.detach().float().reshape(dl.out_channels, -1)), dim=0).contiguous()),
                torch.cat((lg.bias.detach().float(), dl.bias.detach().float())).contiguous()))
            o = be.gemm_h2(be.packed(t.permute(0, 2, 3, 1).reshape(B * h * w, C)), wp, bias=b).view(B, h, w, -1)
            A = lg.out_channels
            return o[..., :A].permute(0, 3, 1, 2).contiguous(), o[..., A:].permute(0, 3, 1, 2).contiguous()
        if self._rows16_path(t) and tuple(lg.kernel_size) == (1, 1) and tuple(dl.kernel_size) == (1, 1):
            be, dt = _backend.get_backend(), self.compute_dtype
            B, C, h, w = t.shape
            wp, b = self._wpack(("rpn_out", dt), (lg.weight, lg.bias, dl.weight, dl.bias), lambda: (
                be.rows16(torch.cat((lg.weight.detach().reshape(lg.out_channels, -1), dl.weight.detach().reshape(dl.out_channels, -1)), dim=0).to(dt).contiguous(), dt),
                torch.cat((lg.bias.detach().float(), dl.bias.detach().float())).contiguous()))
            o = be.gemm_b16(be.rows16(t.permute(0, 2, 3, 1).reshape(B * h * w, C), dt), wp, bias=b, out_f32=True).view(B, h, w, -1)
            A = lg.out_channels
            return o[..., :A].permute(0, 3, 1, 2).contiguous(), o[..., A:].permute(0, 3, 1, 2).contiguous()
        with self._amp():
            return self.rpn_logits(t).float(), self.rpn_deltas(t).float()

    @torch.no_grad()
    def proposals(self, c4, img_size, t=None):
        """t: rpn_activation(c4) when the caller already has it (the on-line RPN harvest reads the same map)."""
        if t is None:
            t = self.rpn_activation(c4)
        if self.online_rpn is not None:
            logits, deltas = self.online_rpn(t)
        else:
            logits, deltas = self.rpn_outputs(t)
        anchors = self.grid_anchors(c4.shape[2], c4.shape[3], c4.device)
        return rpn_proposals(logits, deltas, anchors, img_size, self.pre_nms_top_n, self.post_nms_top_n, self.rpn_nms)

    @torch.no_grad()
    def proposals_batch(self, c4, img_size, t=None):
        """proposals() for the trunk features of B images of one size: [(boxes_b, scores_b)].  One RPN-head pass, one top-k,
        one suppression launch pair and one host synchronisation for the batch (rpn_proposals_batch); with an on-line RPN head
        (its scoring is written for one image) or a backend without the batched suppression: image after image."""
        be = _backend.get_backend()
        B = c4.shape[0]
        if B == 1 or self.online_rpn is not None or not (c4.is_cuda and hasattr(be, "nms_batched") and _nms_takes_max_keep(be)):
            return [self.proposals(c4[b:b + 1], img_size, None if t is None else t[b:b + 1]) for b in range(B)]
        if t is None:
            t = self.rpn_activation(c4)
        logits, deltas = self.rpn_outputs(t)
        anchors = self.grid_anchors(c4.shape[2], c4.shape[3], c4.device)
        return rpn_proposals_batch(logits, deltas, anchors, img_size, self.pre_nms_top_n, self.post_nms_top_n, self.rpn_nms)

    def _group_static(self, images, gt_slots, anchors):
        """The forward of B images of one size with NOTHING data-dependent in its shapes, so that it can be replayed from ONE HIP
        graph (forward_group): every image has `gpad` ground-truth slots (gt_slots (B, gpad, 4); unused ones hold a dummy box)
        followed by post_nms_top_n proposal slots — the first survivors of the suppression in score order, compacted by a kernel
        instead of a masked select; when fewer survive, the tail slots hold a dummy box and the count n[b] says where they start.  Returns (slots (B, gpad + P, 4), n (B,), features (B (gpad + P), D), RPN activation
        (B, C, h, w), mask activation of the ground-truth slots (B gpad, mask_dim, r, r))."""
        be = _backend.get_backend()
        B, gpad = gt_slots.shape[0], gt_slots.shape[1]
        img_size = (images.shape[3], images.shape[2])
        c4 = self._c4_eager(images)
        t = self.rpn_activation(c4)
        logits, deltas = self.rpn_outputs(t)
        _, A, H, W = logits.shape            # (anchors: an argument — made from host constants, a copy a capture does not allow)
        if anchors.shape[0] != H * W * A:
            raise ValueError("forward_group: %d anchors for a %d x %d map of %d types" % (anchors.shape[0], H, W, A))
        # the proposal stage with no library sort / top-k in it (their replay from a captured graph faulted: tools/
        # group_graph_bisect.py) and no data-dependent shape: the k best candidates sorted, decoded and clipped by one launch,
        # the suppression of all images by one launch pair, the first P survivors compacted into their slots by one more
        k = min(self.pre_nms_top_n, H * W * A)
        cand, _, _ = be.rpn_topk_decode(logits, deltas, anchors, k, img_size, DELTA_CLAMP)
        keep = be.nms_batched(cand, torch.full((B,), k, dtype=torch.int32, device=cand.device), self.rpn_nms, max_keep=self.post_nms_top_n,
                              as_bool=False)
        P = min(self.post_nms_top_n, k)
        props, nkept = be.nms_compact(cand, keep, P)
        slots = torch.cat((gt_slots, props), dim=1)
        bidx = torch.arange(B, device=cand.device).repeat_interleave(gpad + P)
        maps = self.roi_head_maps(c4, slots.reshape(-1, 4), batch_idx=bidx)
        feats = maps.mean(dim=(2, 3))
        act = self.mask_activation(maps.view(B, gpad + P, *maps.shape[1:])[:, :gpad].reshape(B * gpad, *maps.shape[1:])) if gpad else None
        return slots, nkept, feats, t, act

    @torch.no_grad()
    def forward_group(self, images, gt_boxes_list, capture=True):
        """forward_batch for a group of B images, replayed from ONE HIP graph per (shape, ground-truth slots): trunk, RPN head,
        top-k, decoding, suppression, RoIAlign, conv5 head, mask activation — ~300 launches per image launch by launch, one
        graph launch once captured.  Returns a list of per-image dicts: boxes (ground truth first), feats, t (RPN activation),
        act (mask activation of the ground-truth rows).  One host synchronisation (the survivor counts).  begin / finish: the
        same in two halves — begin queues the group's forward and returns at once, finish waits for it —, so that ONE host
        thread can harvest group k while the GPU runs group k + 1 (OnlineFeatureExtractor.train)."""
        return self.forward_group_finish(self.forward_group_begin(images, gt_boxes_list, capture))

    @torch.no_grad()
    def forward_group_begin(self, images, gt_boxes_list, capture=True):
        B = images.shape[0]
        dev = images.device
        G = [0 if g is None else int(len(g)) for g in gt_boxes_list]
        gpad = (max(G) + 3) // 4 * 4
        gt_slots = torch.zeros((B, gpad, 4), dtype=torch.float32, device=dev)
        if gpad:
            gt_slots[:, :, 2:] = 15.0                       # unused slots: a harmless 16 x 16 box
            for b, g in enumerate(gt_boxes_list):
                if G[b]:
                    gt_slots[b, :G[b]] = g.to(dev).float()
        h, w = -(-images.shape[2] // self.stride), -(-images.shape[3] // self.stride)          # the C4 map: ceil(size / 16)
        anchors = self.grid_anchors(h, w, dev)
        out = self._group_graphs(images, gt_slots, anchors, key_extra=(self.pre_nms_top_n, self.post_nms_top_n, self.rpn_nms,
                                                                       self.compute_dtype, self.resolution, self._weights_key()),
                                 capture=capture)
        return (out, G, gpad)

    @torch.no_grad()
    def forward_group_finish(self, handle):
        """The per-image results of a forward_group_begin: waits for that group's forward (the survivor counts are read on the
        host — the group's one synchronisation; call it on the stream the group was begun on)."""
        (slots, n, feats, t, act), G, gpad = handle
        B, dev = slots.shape[0], slots.device
        n = n.tolist()
        P = slots.shape[1] - gpad
        sel = torch.tensor([b * (gpad + P) + j for b in range(B) for j in list(range(G[b])) + list(range(gpad, gpad + n[b]))],
                           dtype=torch.int64)
        sel = to_device(sel, dev)                           # (staged through page-locked memory: the host list is a temporary)
        boxes_all, feats_all = slots.view(-1, 4).index_select(0, sel), feats.index_select(0, sel)
        out, at = [], 0
        for b in range(B):
            r = G[b] + n[b]
            out.append({"boxes": boxes_all[at:at + r], "feats": feats_all[at:at + r], "t": t[b],
                        "act": None if act is None or not G[b] else act[b * gpad:b * gpad + G[b]]})
            at += r
        return out

    @torch.no_grad()
    def roi_head_maps(self, c4, boxes, batch_idx=None):
        """(R, D, r/2, r/2): RoIAlign r x r @ 1/stride (HIP kernel) -> conv5 head.  batch_idx (R,): the image of c4 (N > 1)
        each box is pooled from — the RoIs of a batch of images go through the head as ONE row matrix."""
        be = _backend.get_backend()
        first = (torch.zeros((boxes.shape[0], 1), device=boxes.device) if batch_idx is None
                 else batch_idx.to(device=boxes.device, dtype=boxes.dtype).view(-1, 1))
        rois = torch.cat((first, boxes), dim=1)
        st = self.head.rows_form() if hasattr(self.head, "rows_form") else 0
        if st > 0 and hasattr(be, "roi_align_rows") and boxes.shape[0] > 0:
            # the head's first 1 x 1 convolutions read every st-th position of the crop: RoIAlign forms those bins only,
            # directly as the rows the head's GEMMs consume (no 14 x 14 crop, no NCHW -> NHWC copy)
            rows, (R, OH, OW) = be.roi_align_rows(c4, rois, 1.0 / self.stride, (self.resolution, self.resolution), 0, step=st)
            with self._amp():
                return self.head.forward_rows(rows, R, OH, OW).float()
        crops = be.roi_align(c4, rois, 1.0 / self.stride, (self.resolution, self.resolution), 0)
        with self._amp():
            return self.head(crops).float()

    @torch.no_grad()
    def roi_features(self, c4, boxes):
        """(R, D): conv5 head maps, global average pooled."""
        return self.roi_head_maps(c4, boxes).mean(dim=(2, 3))

    @torch.no_grad()
    def mask_activation(self, head_maps):
        """(R, mask_dim, r, r) = relu(conv5_mask(head maps))  (roi_mask_predictors.py:38).  conv5_mask is a 2 x 2, stride-2
        transposed convolution: every input position feeds its own 2 x 2 output block, i.e. out[(r, i, j), (oc, di, dj)] =
        in[(r, i, j), :] . W[:, oc, di, dj] — ONE product over the head's NHWC rows.  On the GPU in f32 it runs on the split-f16
        tile core with bias + ReLU in the GEMM's epilogue (f32 accuracy; also what keeps this step inside a captured forward:
        the library's transposed convolution brought a workspace of its own into the graph)."""
        be = _backend.get_backend() if head_maps.is_cuda else None
        cm = self.conv5_mask
        if (be is not None and hasattr(be, "gemm_h2") and self.compute_dtype is None and head_maps.dtype == torch.float32
                and head_maps.shape[0] > 0 and tuple(cm.kernel_size) == (2, 2) and tuple(cm.stride) == (2, 2) and tuple(cm.padding) == (0, 0)
                and tuple(cm.output_padding) == (0, 0) and cm.groups == 1):
            R, Cin, H, W = head_maps.shape
            rows = head_maps.permute(0, 2, 3, 1).reshape(R * H * W, Cin)          # a view when the maps are the head's NHWC rows
            key = (cm.weight.data_ptr(), cm.weight._version, cm.bias._version)
            hit = self.__dict__.get("_mask_pack")
            if hit is None or hit[0] != key:
                if hit is not None:
                    self._drop_captured()                                                         # (graphs that point at the old pack)
                wt = cm.weight.detach().float().reshape(Cin, -1).t().contiguous()                 # (Cout * 4, Cin): column (oc, di, dj)
                hit = self.__dict__["_mask_pack"] = (key, be.packed(wt), cm.bias.detach().float().repeat_interleave(4).contiguous())
            out = be.gemm_h2(be.packed(rows), hit[1], bias=hit[2], relu=True)               # (R H W, Cout * 4)
            Cout = cm.out_channels
            return out.view(R, H, W, Cout, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(R, Cout, 2 * H, 2 * W)
        with self._amp():
            return F.relu(self.conv5_mask(head_maps.contiguous())).float()      # (the head hands out an NHWC-strided view)

    @torch.no_grad()
    def forward(self, image, gt_boxes=None):
        """image (1, 3, H, W) already normalised / resized; returns (boxes (R, 4), feats (R, D), c4)
        with the ground-truth boxes prepended to the proposals (generalized_rcnn_getProposals.py:90-96)."""
        c4 = self.c4(image)
        img_size = (image.shape[3], image.shape[2])
        boxes, _ = self.proposals(c4, img_size)
        if gt_boxes is not None and len(gt_boxes):
            boxes = torch.cat((gt_boxes.to(boxes.device).float(), boxes), dim=0)
        return boxes, self.roi_features(c4, boxes), c4


def _model_anchors(m, h, w, dev):
    """The anchors of an (h, w) map of model m on dev: the model's cached ones when it keeps them (OnlineDetectionModel.grid_anchors)."""
    if hasattr(m, "grid_anchors"):
        return m.grid_anchors(h, w, dev)
    return grid_anchors(h, w, m.stride, m.cells.to(dev))


def forward_batch(model, images, gt_boxes_list=None, want_rpn_activation=False):
    """OnlineDetectionModel.forward for B pre-processed images of ONE size, images (B, 3, H, W): one trunk call (replayed from
    a HIP graph per shape), one proposal stage (proposals_batch) and ONE pass of the RoI head over all images' RoIs — at batch
    1 the trunk is ~120 latency-bound launches and the head's GEMMs see 300 x 49 rows where the 256 x 256 tile core wants
    tens of thousands (feature_proposal_extractor.py:228-281 walks one image per iteration; SURVEY 2.1 leaves batching to the
    build).  Returns ([(boxes_b, feats_b)], trunk, maps, offsets): per image what forward() returns, the trunk features, the
    head maps of all RoIs (image after image) and the row offset of each image in them; with want_rpn_activation also the
    (B, C, h, w) RPN activation the proposal stage computed."""
    B = images.shape[0]
    trunk = model.c4(images)
    img_size = (images.shape[3], images.shape[2])
    tr0 = trunk if torch.is_tensor(trunk) else trunk[0]
    act = None
    if hasattr(model, "proposals_batch"):
        if want_rpn_activation:              # computed once: the proposal stage and the caller's on-line RPN harvest read the same map
            act = model.rpn_activation(trunk)
        props = model.proposals_batch(trunk, img_size, t=act)
    else:
        props = [model.proposals(model.trunk_slice(trunk, b) if hasattr(model, "trunk_slice") else trunk[b:b + 1], img_size) for b in range(B)]
    boxes_list = []
    for b in range(B):
        bx = props[b][0]
        gt = None if gt_boxes_list is None else gt_boxes_list[b]
        if gt is not None and len(gt):
            bx = torch.cat((gt.to(bx.device).float(), bx), dim=0)
        boxes_list.append(bx)
    counts = [int(bx.shape[0]) for bx in boxes_list]
    # (the image index of every RoI from B fills: an upload of the counts waits for whatever the GPU is running, and
    # repeat_interleave with a tensor of repeats reads their sum back — two more stalls of the host right behind the proposal
    # stage's own read, while the GPU has nothing queued)
    bidx = torch.cat([torch.full((c,), b, dtype=torch.int64, device=tr0.device) for b, c in enumerate(counts)])
    maps = model.roi_head_maps(trunk, torch.cat(boxes_list, dim=0), batch_idx=bidx)
    feats = maps.mean(dim=(2, 3))
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    out = [(boxes_list[b], feats[offs[b]:offs[b + 1]]) for b in range(B)], trunk, maps, offs
    return out + (act,) if want_rpn_activation else out


def detect(model, image, orig_size=None, score_thresh=-2.0, nms_thresh=0.3, detections_per_img=100, with_masks=False):
    """Test-time forward of one pre-processed image (1, 3, H, W) with the model's on-line heads
    (GeneralizedRCNN.forward at test time: RPN proposals -> RoI features -> OnlineBoxPredictor -> post-processing in the
    ORIGINAL image frame, generalized_rcnn.py + OnlineDetectionPostProcessor.py:12-79; masks of the kept detections
    through OnlineMaskPredictor + Masker when with_masks).  orig_size = (width, height) of the image before resizing.
    Returns dict(boxes, scores, labels[, masks]) or None, and the proposals."""
    from .postprocess import paste_masks, postprocess_detections, select_class_masks
    if model.online_box is None:
        raise RuntimeError("detect: the model has no on-line box predictor (update_model(models_detection=...))")
    c4 = model.c4(image)
    img_size = (image.shape[3], image.shape[2])
    orig_size = tuple(orig_size) if orig_size is not None else img_size
    boxes, _ = model.proposals(c4, img_size)
    maps = model.roi_head_maps(c4, boxes)
    scores, deltas = model.online_box(maps.mean(dim=(2, 3)))
    res = postprocess_detections(scores, deltas, boxes, orig_size, score_thresh, nms_thresh, detections_per_img,
                                 proposals_size=img_size)
    if res is not None and with_masks and model.online_mask is not None and len(res["boxes"]):
        back = res["boxes"] * res["boxes"].new_tensor([img_size[0] / orig_size[0], img_size[1] / orig_size[1]] * 2)
        pix = model.online_mask(model.mask_activation(model.roi_head_maps(c4, back)))
        res["masks"] = paste_masks(select_class_masks(pix, res["labels"]), res["boxes"], orig_size)
    return res, boxes


def detect_batch(model, images, orig_sizes=None, score_thresh=-2.0, nms_thresh=0.3, detections_per_img=100, with_masks=False):
    """detect() for B pre-processed images of ONE size, images (B, 3, H, W): [(result, proposals)] per image.  One forward for the
    group (forward_batch: trunk, proposal stage and RoI head once — from two 600 x 800 images on the f32 trunk and heads are
    one chain of row GEMMs), ONE pass of the on-line box predictor over all images' RoI features (the FALKON scoring and the RLS
    regressors act row by row), then the reference's post-processing per image in its own original frame
    (OnlineDetectionPostProcessor.py:12-79).  The reference's test loop walks one image per iteration
    (engine/inference.py:268-357); the evaluator drop-in groups consecutive images of one size."""
    from .postprocess import paste_masks, postprocess_detections, select_class_masks
    if model.online_box is None:
        raise RuntimeError("detect_batch: the model has no on-line box predictor (update_model(models_detection=...))")
    B = images.shape[0]
    img_size = (images.shape[3], images.shape[2])
    sizes = [tuple(s) if s is not None else img_size for s in (orig_sizes if orig_sizes is not None else [None] * B)]
    if B == 1:
        return [detect(model, images, sizes[0], score_thresh, nms_thresh, detections_per_img, with_masks)]
    per, trunk, _, offs = forward_batch(model, images)
    scores, deltas = model.online_box(torch.cat([p[1] for p in per], dim=0))
    out = []
    for b in range(B):
        boxes = per[b][0]
        res = postprocess_detections(scores[offs[b]:offs[b + 1]], deltas[offs[b]:offs[b + 1]], boxes, sizes[b], score_thresh, nms_thresh,
                                     detections_per_img, proposals_size=img_size)
        if res is not None and with_masks and model.online_mask is not None and len(res["boxes"]):
            tb = model.trunk_slice(trunk, b) if hasattr(model, "trunk_slice") else trunk[b:b + 1]
            back = res["boxes"] * res["boxes"].new_tensor([img_size[0] / sizes[b][0], img_size[1] / sizes[b][1]] * 2)
            pix = model.online_mask(model.mask_activation(model.roi_head_maps(tb, back)))
            res["masks"] = paste_masks(select_class_masks(pix, res["labels"]), res["boxes"], sizes[b])
        out.append((res, boxes))
    return out


class DetectorFeatureExtractor:
    """The per-image harvest loop of FeatureExtractorDetector.train
    (feature_extractor_detector/extract_features_detector.py:96-292 with
    engine/feature_proposal_extractor.py:228-281): images are walked one at a time; with several
    ranks each takes every world-th image and keeps the rows it harvested (already row-sharded
    for the trainers)."""

    def __init__(self, model, num_classes, iterations=10, batch_size=2000, neg_iou_thresh=0.3, reg_min_overlap=0.6,
                 shuffle_negatives=False, rank=0, world=1):
        self.model, self.rank, self.world = model, rank, world
        self.kw = dict(num_classes=num_classes, iterations=iterations, batch_size=batch_size, neg_iou_thresh=neg_iou_thresh,
                       reg_min_overlap=reg_min_overlap, shuffle_negatives=shuffle_negatives)

    def train(self, samples, use_only_gt_positives=True):
        """samples: sequence of (image (1,3,H,W), gt_boxes (G,4), gt_labels list[int] 1..C)."""
        samples = list(samples)[self.rank::self.world]
        dev = next(self.model.parameters()).device
        hv = DetectorHarvester(self.model.feat_dim, num_images=max(len(samples), 1), device=dev, **self.kw)
        for image, gt_boxes, gt_labels in samples:
            image = image.to(dev)
            boxes, feats, _ = self.model(image, gt_boxes)
            hv.add_image(feats, boxes, gt_boxes.to(dev), list(gt_labels), [image.shape[3], image.shape[2]])
        return hv.finalize(use_only_gt_positives)

    def test(self, samples):
        samples = list(samples)[self.rank::self.world]
        dev = next(self.model.parameters()).device
        hv = DetectorHarvester(self.model.feat_dim, num_images=max(len(samples), 1), device=dev, **self.kw)
        for image, gt_boxes, gt_labels in samples:
            image = image.to(dev)
            boxes, feats, _ = self.model(image, gt_boxes)
            hv.add_test_image(feats, boxes, len(gt_labels), [image.shape[3], image.shape[2]])
        return hv.test_boxes


def _unpack(sample):
    image, gt_boxes, gt_labels = sample[0], sample[1], sample[2]
    masks = sample[3] if len(sample) > 3 else None
    return image, gt_boxes, gt_labels, masks


class OnlineFeatureExtractor:
    """One pass over the images that harvests everything the "Ours" pipeline trains on
    (FeatureExtractorRPNDetector.train, feature_extractor_RPN_detector/extract_features_rpn_detector.py:105-369):
    on-line RPN rows per anchor type, detector rows per class (on the pretrained RPN's proposals with
    the ground truth prepended) and, optionally, segmentation pixel rows per class.
    `parts` selects what is harvested: any of "rpn", "detector", "mask"."""

    def __init__(self, model, num_classes, parts=("rpn", "detector"), det=None, rpn=None, mask=None, rank=0, world=1,
                 pipeline=True, trunk_batch=8, num_images=None):
        self.model, self.C, self.parts, self.rank, self.world = model, num_classes, tuple(parts), rank, world
        # num_images: the length of the WHOLE image stream when `train` is handed a part of it (the reference sizes its
        # per-image quota of negatives from len(dataset), box_head_getProposals.py:67: ceil(BATCH_SIZE x ITERATIONS / images));
        # default: the images train() is given
        self.num_images = num_images
        self.pipeline = pipeline        # on a GPU: forward of the next image on a second thread / stream while this one is harvested
        # on a GPU: consecutive images of one size share ONE forward — trunk, proposal stage and RoI head each run once for the
        # group (forward_batch; 1 = one image per call).  With > 1 a harvested row depends, in its last bits, on the image's
        # neighbours in the list and on the rank sharding (the convolution library picks its algorithm per batch size) and
        # differs in rounding from detect() / forward(), which see one image: trunk_batch = 1 (cfg_options['trunk_batch'] of
        # the facade) is the per-image bit-reproducible setting
        self.trunk_batch = trunk_batch
        self.det_kw = dict(iterations=10, batch_size=2000, neg_iou_thresh=0.3, reg_min_overlap=0.6, shuffle_negatives=False)
        self.rpn_kw = dict(iterations=10, batch_size=2000, neg_iou_thresh=0.3, pos_iou_thresh=0.7, shuffle_negatives=False)
        self.mask_kw = dict(batch_size=20000, sampling_factor=0.3)
        self.det_kw.update(det or {})
        self.rpn_kw.update(rpn or {})
        self.mask_kw.update(mask or {})

    def train(self, samples, use_only_gt_positives=True, save_dir=None):
        """`save_dir`: spill to the reference's on-disk feature cache (odx/storage.py) instead of returning the rows
        (SAVE_FEATURES_DETECTOR / SAVE_FEATURES_RPN: the reference's train() returns nothing in that mode)."""
        samples = list(samples)[self.rank::self.world]
        m = self.model
        dev = next(m.parameters()).device
        n = max(len(samples) if self.num_images is None else int(self.num_images), 1)
        hv_det = DetectorHarvester(m.feat_dim, self.C, num_images=n, device=dev, **self.det_kw) if "detector" in self.parts else None
        hv_rpn = RPNHarvester(m.backbone.out_channels, m.cells.shape[0], num_images=n, device=dev, **self.rpn_kw) if "rpn" in self.parts else None
        hv_mask = MaskHarvester(m.mask_dim, self.C, device=dev, **self.mask_kw) if "mask" in self.parts else None

        def forward_one(sample, c4=None):
            """Everything of an image that does not depend on the harvesters' state: trunk, RPN activation, proposals
            (+ ground truth), head maps, mask activation.  c4: the image's trunk features when they were computed with a
            neighbour's (forward_items)."""
            image, gt_boxes, gt_labels, masks = _unpack(sample)
            image, gt_boxes = image.to(dev), gt_boxes.to(dev).float()
            img_size = (image.shape[3], image.shape[2])
            item = {"gt_boxes": gt_boxes, "gt_labels": list(gt_labels), "img_size": img_size}
            with torch.no_grad():
                if c4 is None:
                    c4 = m.c4(image)
                t_act = None
                if hv_rpn is not None and len(gt_boxes):
                    item["anchors"] = _model_anchors(m, c4.shape[2], c4.shape[3], dev)
                    t_act = m.rpn_activation(c4)
                    item["t"] = t_act[0]
                if hv_det is None and hv_mask is None:
                    return item
                # (the RPN activation is computed once: the harvest and the proposal stage read the same map)
                boxes, _ = m.proposals(c4, img_size, t_act) if (t_act is not None and not hasattr(m, "trunk_slice")) else m.proposals(c4, img_size)
                if len(gt_boxes):
                    boxes = torch.cat((gt_boxes, boxes), dim=0)
                maps = m.roi_head_maps(c4, boxes)
                item["boxes"] = boxes
                if hv_det is not None:
                    item["feats"] = maps.mean(dim=(2, 3))
                if hv_mask is not None and masks is not None and len(gt_labels):
                    item["act"] = m.mask_activation(maps[:len(gt_labels)])
                    item["mg"] = project_masks_on_boxes(masks.to(dev), gt_boxes, item["act"].shape[2])
            return item

        def group_graph_ok(group):
            return (dev.type == "cuda" and (hv_det is not None or hv_mask is not None) and len(group) > 1
                    and len(group) == max(1, int(self.trunk_batch)) and hasattr(m, "forward_group_begin") and m.online_rpn is None
                    and getattr(m._group_graphs, "enabled", False))

        def group_begin(group):
            """Queue the forward of a full group as ONE HIP-graph launch (OnlineDetectionModel.forward_group_begin) and return
            at once; None for a group that goes launch by launch (forward_group does all of it then)."""
            if not group_graph_ok(group):
                return None
            unp = [_unpack(smp) for smp in group]
            with torch.no_grad():
                return m.forward_group_begin(torch.cat([u[0].to(dev) for u in unp], dim=0), [u[1].to(dev).float() for u in unp])

        def forward_group(group, graphed=None):
            """forward_one for `len(group)` > 1 images of one size through ONE forward (forward_batch: trunk, proposal stage
            and RoI head each once for the group; `graphed`: the group's group_begin, whose results are collected here); the
            per-image items come out in the group's order."""
            unp = [_unpack(smp) for smp in group]
            # (a graphed group's images were concatenated — and copied into the graph — by group_begin: not again here)
            images = torch.cat([u[0].to(dev) for u in unp], dim=0) if graphed is None else None
            gts = [u[1].to(dev).float() for u in unp]
            img_size = (unp[0][0].shape[3], unp[0][0].shape[2])
            items = [{"gt_boxes": gts[j], "gt_labels": list(unp[j][2]), "img_size": img_size} for j in range(len(group))]
            gts_host = None
            if hv_mask is not None and any(u[3] is not None and len(u[2]) for u in unp):
                # the crop windows of the mask projection are decided on the host: the group's boxes in ONE read (none at all
                # for boxes handed in on the host), not one per image
                if all(not torch.as_tensor(u[1]).is_cuda for u in unp):
                    gts_host = [torch.as_tensor(u[1]).float().reshape(-1, 4).tolist() for u in unp]
                else:
                    flat, at, gts_host = torch.cat([g.reshape(-1, 4) for g in gts]).tolist(), 0, []
                    for g in gts:
                        gts_host.append(flat[at:at + len(g)])
                        at += len(g)
            need_heads = hv_det is not None or hv_mask is not None
            want_t = hv_rpn is not None and any(len(g) for g in gts)
            if graphed is not None:
                # a full group: the whole forward replayed from one HIP graph (no host work but the copies in and out)
                res = m.forward_group_finish(graphed)
                anchors = _model_anchors(m, res[0]["t"].shape[1], res[0]["t"].shape[2], dev) if want_t else None
                for j, (it, r) in enumerate(zip(items, res)):
                    it["boxes"] = r["boxes"]
                    if hv_det is not None:
                        it["feats"] = r["feats"]
                    if want_t and len(gts[j]):
                        it["anchors"], it["t"] = anchors, r["t"]
                    masks = unp[j][3]
                    if hv_mask is not None and masks is not None and len(it["gt_labels"]) and r["act"] is not None:
                        it["act"] = r["act"]
                        it["mg"] = project_masks_on_boxes(masks.to(dev), gts[j], it["act"].shape[2], boxes_host=None if gts_host is None else gts_host[j])
                return items
            with torch.no_grad():
                ts = None
                if need_heads:
                    res = forward_batch(m, images, gts, want_rpn_activation=want_t)
                    per, c4s, maps, offs = res[:4]
                    ts = res[4] if want_t else None
                else:
                    c4s = m.c4(images)
                if want_t:
                    if ts is None:
                        ts = m.rpn_activation(c4s)
                    anchors = _model_anchors(m, c4s.shape[2], c4s.shape[3], dev)
                    for j, it in enumerate(items):
                        if len(gts[j]):
                            it["anchors"], it["t"] = anchors, ts[j]
                if need_heads:
                    for j, it in enumerate(items):
                        it["boxes"] = per[j][0]
                        if hv_det is not None:
                            it["feats"] = per[j][1]
                        masks = unp[j][3]
                        if hv_mask is not None and masks is not None and len(it["gt_labels"]):
                            it["act"] = m.mask_activation(maps[offs[j]:offs[j] + len(it["gt_labels"])])
                            it["mg"] = project_masks_on_boxes(masks.to(dev), gts[j], it["act"].shape[2], boxes_host=None if gts_host is None else gts_host[j])
            return items

        def forward_items(seq):
            """forward_one for every sample, in order; consecutive images of one size go through the forward together,
            `trunk_batch` at a time (forward_group).  A network whose trunk hands out several maps (the pyramid of odx/fpn.py)
            shares the trunk call only and runs everything behind it per image."""
            k = max(1, int(self.trunk_batch)) if dev.type == "cuda" else 1
            # (forward_batch: one trunk call and ONE pass of the RoI head for the group; the proposal stage in one go for the C4
            # network, image after image on the pyramid of odx/fpn.py, whose levels it already suppresses with one launch pair)
            whole = ((hasattr(m, "proposals_batch") and not hasattr(m, "trunk_slice"))
                     or (hasattr(m, "trunk_slice") and hv_rpn is None and hv_mask is None))
            i = 0
            while i < len(seq):
                group = [seq[i]]
                if k > 1:
                    shape = tuple(_unpack(seq[i])[0].shape)
                    while len(group) < k and i + len(group) < len(seq) and tuple(_unpack(seq[i + len(group)])[0].shape) == shape:
                        group.append(seq[i + len(group)])
                # ("_end": the last image of its group — the consumer harvests a group's images together, harvest_group)
                if len(group) == 1:
                    item = forward_one(group[0])
                    item["_end"] = True
                    yield item
                elif whole:
                    # (a full group from one HIP graph — but only on the caller's own thread: the two-thread loop below keeps to
                    # plain launches, a capture or replay beside another thread's launches faulted on this runtime)
                    items = forward_group(group, None if in_thread[0] else group_begin(group))
                    items[-1]["_end"] = True
                    for item in items:
                        yield item
                else:
                    with torch.no_grad():
                        c4s = m.c4(torch.cat([_unpack(smp)[0].to(dev) for smp in group], dim=0))
                    cut = getattr(m, "trunk_slice", None)          # (a pyramid is sliced level by level: odx/fpn.py)
                    for j, smp in enumerate(group):
                        item = forward_one(smp, cut(c4s, j) if cut is not None else c4s[j:j + 1])
                        item["_end"] = j + 1 == len(group)
                        yield item
                i += len(group)

        def harvest_one(item):
            """The stateful part, in image order: RNG draws, batch bookkeeping, rows into the buffers."""
            if "t" in item:
                hv_rpn.add_image(item["t"], item["anchors"], item["img_size"], item["gt_boxes"])
            if "feats" in item:
                hv_det.add_image(item["feats"], item["boxes"], item["gt_boxes"], item["gt_labels"], list(item["img_size"]))
            if "act" in item:
                hv_mask.add_image(item["act"], item["mg"], item["gt_labels"])

        def harvest_group(items):
            """harvest_one for the images of a group with ONE host read: every harvester's device work in front of its host
            decisions for all images (prepare: stateless), the counts of all of them in one copy, then the stateful parts image
            by image in the order harvest_one has (commit: RNG draws, batch bookkeeping, rows into the buffers) — same rows,
            same draws.  Image by image each of the three harvesters reads its own counts, and every such read waits for its
            few small kernels to get their turn beside the next group's forward (~0.5 ms each, four per image)."""
            if len(items) < 2 or dev.type != "cuda":
                for item in items:
                    harvest_one(item)
                return
            ctxs, blocks = [], []
            for item in items:
                c = {}
                if "t" in item:
                    c["rpn"] = hv_rpn.prepare(item["t"], item["anchors"], item["img_size"], item["gt_boxes"])
                if "feats" in item:
                    c["det"] = hv_det.prepare(item["feats"], item["boxes"], item["gt_boxes"], item["gt_labels"], list(item["img_size"]))
                if "act" in item:
                    c["mask"] = hv_mask.prepare(item["act"], item["mg"], item["gt_labels"])
                ctxs.append(c)
                blocks.extend(x["block"] for x in c.values() if x is not None and x["block"] is not None)
            host = torch.cat([b.reshape(-1).to(torch.float64) for b in blocks]).tolist() if blocks else []     # counts: exact in f64
            at = [0]

            def take(x, as_int):
                if x is None or x["block"] is None:
                    return []
                n = x["block"].numel()
                vals = host[at[0]:at[0] + n]
                at[0] += n
                return [int(v) for v in vals] if as_int else vals
            got = [{k: take(c[k], k != "rpn") for k in ("rpn", "det", "mask") if k in c} for c in ctxs]      # (the order blocks were listed in)
            for c, g in zip(ctxs, got):
                if "rpn" in c:
                    hv_rpn.commit(c["rpn"], g["rpn"])
                if "det" in c:
                    hv_det.commit(c["det"], g["det"])
                if c.get("mask") is not None:
                    hv_mask.commit(c["mask"], g["mask"])

        def split_groups(seq):
            k = max(1, int(self.trunk_batch)) if dev.type == "cuda" else 1
            out, i = [], 0
            while i < len(seq):
                group = [seq[i]]
                if k > 1:
                    shape = tuple(_unpack(seq[i])[0].shape)
                    while len(group) < k and i + len(group) < len(seq) and tuple(_unpack(seq[i + len(group)])[0].shape) == shape:
                        group.append(seq[i + len(group)])
                out.append(group)
                i += len(group)
            return out

        groups = split_groups(samples)
        in_thread = [False]
        if dev.type == "cuda" and self.pipeline and len(groups) > 1 and any(group_graph_ok(g) for g in groups):
            # ONE host thread, two streams: the forward of group k + 1 is queued (a single HIP-graph launch for a full group:
            # no host work to speak of) on the forward stream BEFORE group k is harvested on the caller's stream, so the GPU
            # runs it under the harvest's host work; the host waits for a group only when it needs its survivor counts.  (Two
            # threads — below, for networks without a graphed group forward — share the interpreter: with ~300 launches per
            # image to queue, the forward thread and the harvesting thread took turns and the loop was bound by their sum.)
            main = torch.cuda.current_stream()
            from . import streams as _streams
            own = _streams.of_default(3, dev)
            fwd = own[-2] if len(own) >= 2 and own[-2].cuda_stream != main.cuda_stream else torch.cuda.Stream()
            fwd.wait_stream(main)
            # (the harvest's own kernels — hundreds of tiny launches with a host read every few of them — on a high-priority stream
            # of their own: measured, no gain: 5.4 against 5.2 ms per image)
            harv = main

            def begin(group):
                with torch.cuda.stream(fwd):
                    return (group, group_begin(group))

            def finish(state):
                group, graphed = state
                with torch.cuda.stream(fwd):
                    items = (forward_group(group, graphed) if graphed is not None else
                             ([forward_one(group[0])] if len(group) == 1 else forward_group(group)))
                    ev = torch.cuda.Event()
                    ev.record(fwd)
                return items, ev

            state = begin(groups[0])
            for gi in range(len(groups)):
                items, ev = finish(state)                               # waits for group gi (its survivor counts)
                if gi + 1 < len(groups):
                    state = begin(groups[gi + 1])                       # group gi + 1 runs on the GPU ...
                harv.wait_event(ev)
                with torch.cuda.stream(harv):
                    for item in items:                                  # ... while group gi is harvested here
                        for v in item.values():
                            if torch.is_tensor(v) and v.is_cuda:
                                v.record_stream(harv)
                    harvest_group(items)
            fwd.synchronize()
        elif dev.type == "cuda" and self.pipeline and len(samples) > 1:
            import queue
            import threading
            in_thread[0] = True
            # HIP graphs of the trunk are captured HERE, on the caller's thread with nothing else in flight — for the group shapes
            # this list starts with — and only replayed by the forward thread below (GraphedCall._tls.forbid): a capture beside
            # another thread's launches on the legacy stream fails ("dependency created on uncaptured work in another stream")
            tg = getattr(m, "_trunk_graphs", None)
            if tg is not None and tg.enabled:
                done_shapes = set()
                with torch.no_grad():
                    for grp in groups:
                        shape = (len(grp),) + tuple(_unpack(grp[0])[0].shape[1:])
                        if shape in done_shapes:
                            continue
                        if len(done_shapes) >= 3:
                            break
                        done_shapes.add(shape)
                        ims = torch.cat([_unpack(smp)[0].to(dev) for smp in grp], dim=0)
                        for _ in range(2):                       # a shape's first call runs launch by launch, its second captures
                            m.c4(ims)
                torch.cuda.synchronize()
            main = torch.cuda.current_stream()
            from . import streams as _streams
            own = _streams.of_default(3, dev)
            fwd = own[-2] if len(own) >= 2 and own[-2].cuda_stream != main.cuda_stream else torch.cuda.Stream()
            fwd.wait_stream(main)
            q = queue.Queue(maxsize=2 * max(1, int(self.trunk_batch)) + 2)      # two groups ahead: the next group's forward starts
            # while the current group's items are still being harvested
            stop = threading.Event()

            def put(x):
                while not stop.is_set():
                    try:
                        q.put(x, timeout=0.1)
                        return True
                    except queue.Full:
                        pass
                return False

            def producer():
                try:
                    torch.cuda.set_device(dev)
                    GraphedCall._tls.forbid = True
                    with torch.cuda.stream(fwd):
                        for item in forward_items(samples):
                            if stop.is_set():
                                return
                            ev = torch.cuda.Event()
                            ev.record(fwd)
                            if not put((item, ev)):
                                return
                    put((None, None))
                except BaseException as e:      # noqa: BLE001 — handed to the consumer thread
                    put((e, None))

            th = threading.Thread(target=producer, daemon=True)
            th.start()
            try:
                pending = []
                while True:
                    item, ev = q.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    main.wait_event(ev)
                    for v in item.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
                    pending.append(item)
                    if item.get("_end", True):                       # the group is complete: one host read for its images
                        harvest_group(pending)
                        pending = []
                harvest_group(pending)
            finally:
                stop.set()                      # a failed harvest must not leave the forward thread blocked on the queue
                th.join()
                fwd.synchronize()
        else:
            for item in forward_items(samples):
                harvest_one(item)
        out = {}
        if save_dir:
            from . import storage
            if hv_rpn is not None:
                storage.save_rpn_features(hv_rpn, save_dir)
            if hv_det is not None:
                storage.save_detector_features(hv_det, save_dir, use_only_gt_positives, hv_mask)
            return out
        if hv_rpn is not None:
            out["rpn"] = hv_rpn.finalize()
        if hv_det is not None:
            out["detector"] = hv_det.finalize(use_only_gt_positives)
        if hv_mask is not None:
            out["mask"] = hv_mask.finalize()
        return out
