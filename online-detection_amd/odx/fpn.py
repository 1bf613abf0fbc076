"""The R-50-FPN variant of the feature forward: ResNet-50 trunk with a feature pyramid, RPN over five levels, multi-level
RoIAlign and the two-layer MLP head — the "ResNet50-FPN + RoIAlign" forward BASELINE.json's north star and config 2 name
(D = 1024 RoI features).

In the reference this variant exists only as selectable registry entries of maskrcnn_benchmark — none of its YAMLs picks
it, all shipped configs run R-50-C4 (odx/extract.py):
    FPN2MLPFeatureExtractor        mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py:55-87
    RPNPostProcessor over levels   mrcnn_modified/modeling/rpn/inference.py:76-175 (forward_for_single_feature_map,
                                   select_over_all_levels)
    the knobs                      mrcnn_modified/config/defaults.py:107-111 (FPN), :130-167 (RPN incl. FPN_POST_NMS_TOP_N_*),
                                   :214-219 (POOLER_*, MLP_HEAD_DIM = 1024)
with the trunk / pyramid / Pooler code living in maskrcnn_benchmark itself (not vendored).  PARITY UNPINNED (as the C4
forward): restated from the published R-50-FPN definition — lateral 1 x 1 and output 3 x 3 convolutions of 256 channels on
C2..C5, nearest-neighbour top-down pathway, P6 by a stride-2 subsampling of P5, one RPN head shared by the levels with one
anchor size per level (32 .. 512) x three aspect ratios, strides 4 .. 64; LevelMapper + RoIAlign 7 x 7, sampling_ratio 2,
scales 1/4 .. 1/32; fc6 (256 * 49 -> 1024) + ReLU, fc7 (1024 -> 1024) + ReLU.

What runs where: the convolutions are PyTorch-ROCm library calls (frozen batch norm folded in); the pooling is ONE
hand-written HIP launch for all RoIs and levels (odx_roi_align_fpn_f32); the proposals' suppression is odx_nms_first_f32
per level; fc6 / fc7 run on the split-f16 tile cores with bias + ReLU in the GEMM epilogue (odx_gemm_h2_f32: f32 accuracy
on the f16 matrix cores).  The model offers the interface of extract.OnlineDetectionModel (c4 / proposals / roi_head_maps /
roi_features / forward / feat_dim / online_box / update_model), so the detector harvester, the on-line box head and
detect() take it unchanged; the on-line RPN and mask heads of the reference are defined on the C4 network only.
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import backend as _backend
from .extract import (DELTA_CLAMP, FrozenBatchNorm2d, _FoldedBN, _nms_takes_max_keep, _stage, _stages_rows, _stages_rows16, _stem_rows,
                      _stages_rows_form, cell_anchors, decode_deltas, grid_anchors)



def _trunk_rows():
    """odx.options trunk != "conv": the trunk stages, pyramid and RPN head as row GEMMs on the library's tile cores."""
    from . import options as _options
    return _options.current().trunk != "conv"

class ResNet50Stages(_FoldedBN):
    """ResNet-50 stem + res2..res5, returning all four stage outputs (C2..C5: 256, 512, 1024, 2048 channels at width 64)."""

    def __init__(self, width=64):
        super().__init__()
        w = width
        self.conv1, self.bn1 = nn.Conv2d(3, w, 7, 2, 3, bias=False), FrozenBatchNorm2d(w)
        self.layer1 = _stage(w, w, 4 * w, 3, 1)
        self.layer2 = _stage(4 * w, 2 * w, 8 * w, 4, 2)
        self.layer3 = _stage(8 * w, 4 * w, 16 * w, 6, 2)
        self.layer4 = _stage(16 * w, 8 * w, 32 * w, 3, 2)
        self.channels = (4 * w, 8 * w, 16 * w, 32 * w)

    def forward(self, x):
        x = F.max_pool2d(self.conv_bn_act("conv1", "bn1", x), 3, 2, 1)
        c2 = self.layer1(x)
        c3 = self.layer2(c2)
        c4 = self.layer3(c3)
        return c2, c3, c4, self.layer4(c4)

    def rows_form(self):
        return _stages_rows_form((self.layer1, self.layer2, self.layer3, self.layer4))

    def forward_rows16(self, x):
        """forward_rows for a trunk run natively in bf16 / f16 (x in that type): [(backend.Rows16 of C2 .. C5, (B, h, w))]."""
        return _stages_rows16(_backend.get_backend(), _stem_rows(self, x), (self.layer1, self.layer2, self.layer3, self.layer4))

    def forward_rows(self, x):
        """The four stages as one chain of row GEMMs on the split-f16 tile cores behind the library's stem (extract._stages_rows):
        [(backend.PackedRows of C2 .. C5 — f32 rows and packed operand —, (B, h, w))]."""
        return _stages_rows(_backend.get_backend(), _stem_rows(self, x), (self.layer1, self.layer2, self.layer3, self.layer4), pack_last=True)


class FeaturePyramid(nn.Module):
    """FPN on C2..C5: inner (lateral 1 x 1) and layer (output 3 x 3) blocks of `out_channels`, top-down by nearest-neighbour
    upsampling, plus P6 = the stride-2 subsampling of P5 (LastLevelMaxPool: max_pool2d(kernel 1, stride 2))."""

    def __init__(self, in_channels, out_channels=256):
        super().__init__()
        self.inner = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in in_channels])
        self.layer = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, 1, 1) for _ in in_channels])
        for m in list(self.inner) + list(self.layer):
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)
        self.out_channels = out_channels

    def _apply(self, fn, *a, **kw):
        self._cast = {}
        self.__dict__.pop("_rows_pack", None)
        self.__dict__.pop("_rows16_pack", None)
        return super()._apply(fn, *a, **kw)

    def _wb(self, conv, dtype):
        """conv's weight / bias in `dtype` (cached per parameter version: a 16-bit trunk does not cast them per call)."""
        if dtype == conv.weight.dtype:
            return conv.weight, conv.bias
        cache = self.__dict__.setdefault("_cast", {})
        key = (id(conv), dtype, conv.weight.data_ptr(), conv.weight._version, conv.bias._version)
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            hit = cache[id(conv)] = (key, conv.weight.detach().to(dtype), conv.bias.detach().to(dtype))
        return hit[1], hit[2]

    def _conv(self, conv, x):
        if torch.is_autocast_enabled("cuda") and x.is_cuda:
            return conv(x)
        w, b = self._wb(conv, x.dtype)
        return F.conv2d(x, w, b, conv.stride, conv.padding)

    def forward(self, cs):
        last = self._conv(self.inner[-1], cs[-1])
        outs = [self._conv(self.layer[-1], last)]
        for k in range(len(cs) - 2, -1, -1):
            lat = self._conv(self.inner[k], cs[k])
            last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            outs.insert(0, self._conv(self.layer[k], last))
        outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
        return tuple(outs)                                   # P2, P3, P4, P5, P6

    def _rows_weights(self, be):
        """Packed operands of the eight convolutions (lateral 1 x 1: (out, in); output 3 x 3: (out, ky kx in)), remade when a
        weight is replaced or written in place."""
        key = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in list(self.inner) + list(self.layer))
        hit = self.__dict__.get("_rows_pack")
        if hit is None or hit[0] != key:
            inner = [(be.packed(c.weight.detach().float().reshape(c.out_channels, -1).contiguous()), c.bias.detach().float().contiguous())
                     for c in self.inner]
            layer = [(be.packed(c.weight.detach().float().permute(0, 2, 3, 1).reshape(c.out_channels, -1).contiguous()),
                      c.bias.detach().float().contiguous()) for c in self.layer]
            hit = self.__dict__["_rows_pack"] = (key, inner, layer)
        return hit[1], hit[2]

    def _rows16_weights(self, be, dt):
        key = (dt,) + tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in list(self.inner) + list(self.layer))
        hit = self.__dict__.get("_rows16_pack")
        if hit is None or hit[0] != key:
            inner = [(be.rows16(c.weight.detach().reshape(c.out_channels, -1).to(dt).contiguous(), dt), c.bias.detach().float().contiguous())
                     for c in self.inner]
            layer = [(be.rows16(c.weight.detach().permute(0, 2, 3, 1).reshape(c.out_channels, -1).to(dt).contiguous(), dt),
                      c.bias.detach().float().contiguous()) for c in self.layer]
            hit = self.__dict__["_rows16_pack"] = (key, inner, layer)
        return hit[1], hit[2]

    def forward_rows16(self, cs):
        """forward_rows for a pyramid in bf16 / f16: the stages' 16-bit rows (ResNet50Stages.forward_rows16) through odx_gemm_b16 /
        conv3x3_rows16, the top-down sum in the 16-bit type (as the library's 16-bit pyramid adds two 16-bit maps); the five
        levels as 16-bit (B, C, h, w) channels-last views of the products' rows."""
        be = _backend.get_backend()
        dt = cs[0][0].buf.dtype
        inner, layer = self._rows16_weights(be, dt)
        outs, last, dims = [], None, None
        for k in range(len(cs) - 1, -1, -1):
            x, (B, H, W) = cs[k]
            lat = be.gemm_b16(x, inner[k][0], bias=inner[k][1], zero_row=True)
            if last is not None:
                _, Hp, Wp = dims
                if hasattr(be, "upsample_add_rows16") and lat.K % 4 == 0 and lat.zero_row:
                    be.upsample_add_rows16(lat, last, B, H, W, Hp, Wp)                 # (in place: lat keeps its zero row)
                else:
                    hi = (torch.arange(H, device=lat.buf.device) * Hp) // H
                    wi = (torch.arange(W, device=lat.buf.device) * Wp) // W
                    up = last.dense.reshape(B, Hp, Wp, -1)[:, hi][:, :, wi]
                    lat = be.rows16((lat.dense.reshape(B, H, W, -1) + up).reshape(B * H * W, -1), dt, zero_row=True)
            last, dims = lat, (B, H, W)
            o = be.conv3x3_rows16(lat, B, H, W, layer[k][0], bias=layer[k][1])
            outs.insert(0, o.dense.reshape(B, H, W, -1).permute(0, 3, 1, 2))          # channels-last VIEWS of the rows, as forward_rows
        outs.append(outs[-1][:, :, ::2, ::2].contiguous(memory_format=torch.channels_last))
        return tuple(outs)

    def forward_rows(self, cs):
        """forward() on the stage outputs of ResNet50Stages.forward_rows — every convolution of the pyramid a product of the
        split-f16 tile cores over NHWC rows: the lateral 1 x 1 convolutions read the stages' packed operands as the trunk's last
        products wrote them, the nearest-neighbour upsampling of the top-down path is a row gather, the 3 x 3 output convolutions
        gather their taps inside the product's operand loads where the library serves the shape (HipBackend.conv3x3_rows).
        Returns the five levels as (B, C, h, w) channels-last VIEWS of the products' rows: the RPN head of a group
        (proposals_batch) and the multi-level RoIAlign (odx_roi_align_fpn_nhwc_f32) read the rows as they are; whoever needs an
        NCHW map (the one-image proposal stage on the convolution library) copies."""
        be = _backend.get_backend()
        inner, layer = self._rows_weights(be)
        outs, last, dims = [], None, None
        for k in range(len(cs) - 1, -1, -1):
            x, (B, H, W) = cs[k]
            lat = be.gemm_h2(x, inner[k][0], bias=inner[k][1])                        # (B H W, C) f32
            meta = None
            if last is not None:
                _, Hp, Wp = dims                                                      # F.interpolate(..., mode="nearest") to (H, W)
                if hasattr(be, "upsample_add_rows") and lat.shape[1] % 4 == 0:
                    # one pass, in place, the sum's maximum left for the packing of the output convolution's operand (two row
                    # gathers, an addition and a pass for the maximum before: 0.37 ms of the 0.9 ms P2 took for eight images)
                    meta = be.upsample_add_rows(lat, last, B, H, W, Hp, Wp)
                else:
                    hi = (torch.arange(H, device=lat.device) * Hp) // H
                    wi = (torch.arange(W, device=lat.device) * Wp) // W
                    up = last.view(B, Hp, Wp, -1)[:, hi][:, :, wi]
                    lat = (lat.view(B, H, W, -1) + up).reshape(B * H * W, -1)
            last, dims = lat, (B, H, W)
            o = be.conv3x3_rows(lat, B, H, W, layer[k][0], bias=layer[k][1], meta=meta)
            outs.insert(0, o.view(B, H, W, -1).permute(0, 3, 1, 2))                   # (B, C, h, w) as a channels-last VIEW of the rows
        # (max pooling, kernel 1, stride 2; as rows of its own: a view of every second position is not a row matrix)
        outs.append(outs[-1][:, :, ::2, ::2].contiguous(memory_format=torch.channels_last))
        return tuple(outs)


class OnlineDetectionModelFPN(nn.Module):
    """trunk + pyramid -> RPN over P2..P6 -> (gt boxes prepended) -> multi-level RoIAlign on P2..P5 -> fc6 / fc7 features."""

    strides = (4, 8, 16, 32, 64)
    anchor_sizes = (32, 64, 128, 256, 512)

    def __init__(self, width=64, fpn_channels=256, mlp_dim=1024, pre_nms_top_n=1000, post_nms_top_n=1000, fpn_post_nms_top_n=1000,
                 rpn_nms=0.7, resolution=7, sampling_ratio=2, seed=0, compute_dtype=None):
        """Defaults = maskrcnn_benchmark's R-50-FPN test-time values (PRE / POST_NMS_TOP_N_TEST 1000 per level,
        FPN_POST_NMS_TOP_N_TEST 1000 over the levels; the reference's defaults.py:155-167 carries 6000 / 1000 / 2000)."""
        super().__init__()
        self.compute_dtype = compute_dtype
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.backbone = ResNet50Stages(width)
        self.fpn = FeaturePyramid(self.backbone.channels, fpn_channels)
        C = fpn_channels
        self.rpn_conv = nn.Conv2d(C, C, 3, 1, 1)
        self.rpn_logits = nn.Conv2d(C, 3, 1)
        self.rpn_deltas = nn.Conv2d(C, 12, 1)
        for l in (self.rpn_conv, self.rpn_logits, self.rpn_deltas):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        self.fc6 = nn.Linear(C * resolution * resolution, mlp_dim)
        self.fc7 = nn.Linear(mlp_dim, mlp_dim)
        for l in (self.fc6, self.fc7):
            nn.init.kaiming_uniform_(l.weight, a=1)
            nn.init.constant_(l.bias, 0)
        torch.random.set_rng_state(g)
        self.cells = [cell_anchors(st, sizes=(sz,)) for st, sz in zip(self.strides, self.anchor_sizes)]     # 3 anchors per level
        self.pre_nms_top_n, self.post_nms_top_n, self.fpn_post_nms_top_n = pre_nms_top_n, post_nms_top_n, fpn_post_nms_top_n
        self.rpn_nms, self.resolution, self.sampling_ratio = rpn_nms, resolution, sampling_ratio
        self.pool_scales = tuple(1.0 / s for s in self.strides[:4])
        self.mlp_dim = mlp_dim
        self.online_rpn = None
        self.online_box = None
        self.online_mask = None
        self._packed = {}
        self._anchor_cache = {}
        from . import options as _options
        self.rows_min_positions = int(_options.load().rows_min_positions)
        from .extract import GraphedCall
        self._trunk_graphs = GraphedCall(self._c4_eager)
        # load_state_dict copies the parameters in place and never goes through _apply: the packed fc weights are derived
        # data of the OLD values then (the C4 network guards the same case in _FoldedBN)
        self.register_load_state_dict_post_hook(OnlineDetectionModelFPN._drop_derived)

    @staticmethod
    def _drop_derived(module, incompatible_keys):
        module._packed.clear()
        module._anchor_cache.clear()
        module._trunk_graphs.clear()

    def _apply(self, fn, *a, **kw):
        self._packed.clear()
        self._anchor_cache.clear()                                  # packed fc weights are derived data
        if "_trunk_graphs" in self.__dict__:
            self._trunk_graphs.clear()                              # (the captured graphs point at the old tensors)
        return super()._apply(fn, *a, **kw)

    @property
    def feat_dim(self):
        return self.mlp_dim

    def _amp(self):
        if self.compute_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast("cuda", dtype=self.compute_dtype)

    # ------------------------------------------------------------------ trunk
    @torch.no_grad()
    def c4(self, image):
        """The pyramid of an image, on the GPU replayed from a HIP graph per image size (extract.GraphedCall: this forward is
        host-bound at batch 1 — trunk + pyramid are ~170 launches and the proposal stage behind them synchronises with the
        host per level, so the host never runs ahead)."""
        # every weight's (storage, in-place version) in the key: a graph replays the tensors and packs it was captured with
        wk = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        return self._trunk_graphs(image, key_extra=(self.compute_dtype, wk))

    def _c4_eager(self, image):
        """The trunk features of an image — here the pyramid (P2 .. P6) (the method keeps extract's name: the harvest loops
        call model.c4 / proposals / roi_head_maps on whatever the trunk hands out).  f32 maps; under a bf16 trunk the maps stay
        in bf16: the RPN head convolves them in bf16 anyway and the RoIAlign launch converts the four levels it pools from —
        five casts to f32 here plus five casts back in the proposal stage were most of what made the bf16 forward SLOWER
        than the f32 one (6.6 against 5.5 ms in round 3's bench line)."""
        if self._rows_path(image):
            return self.fpn.forward_rows(self.backbone.forward_rows(image))
        if self._rows16_path(image):
            return self.fpn.forward_rows16(self.backbone.forward_rows16(image.to(self.compute_dtype)))
        if self.compute_dtype is not None and image.is_cuda:
            # natively in the 16-bit dtype: the folded / cached weights are already in it, and an autocast context costs host
            # time per operation — the bf16 trunk's kernels are shorter than the f32 one's and the forward became host-bound
            return tuple(self.fpn(self.backbone(image.to(self.compute_dtype))))
        with self._amp():
            return tuple(self.fpn(self.backbone(image)))

    pyramid = c4

    def _rows16_path(self, x):
        """_rows_path for compute_dtype = bf16 / f16: trunk and pyramid on 16-bit rows (odx_gemm_b16 / odx_gemm_b16_taps)."""
        import os
        if not (x.is_cuda and self.compute_dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled()
                and _trunk_rows()):
            return False
        if x.shape[0] * (-(-x.shape[2] // 16)) * (-(-x.shape[3] // 16)) < self.rows_min_positions:
            return False
        return hasattr(_backend.get_backend(), "conv3x3_rows16") and self.backbone.rows_form() and self.fpn.out_channels % 8 == 0

    def _rows_path(self, x):
        """The f32 trunk and pyramid on the GPU as row GEMMs on the library's tile cores (as OnlineDetectionModel._rows_path: from
        rows_min_positions stride-16 positions per call on; odx.options trunk="conv" keeps the convolution library)."""
        import os
        if not (x.is_cuda and self.compute_dtype is None and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and not torch.is_autocast_enabled("cuda") and _trunk_rows()):
            return False
        if x.shape[0] * (-(-x.shape[2] // 16)) * (-(-x.shape[3] // 16)) < self.rows_min_positions:
            return False
        return hasattr(_backend.get_backend(), "gemm_h2") and self.backbone.rows_form() and self.fpn.out_channels % 8 == 0

    @staticmethod
    def trunk_slice(trunk, j):
        """Image j of a batched trunk output."""
        return tuple(p[j:j + 1] for p in trunk)

    def update_model(self, models_rpn=None, models_detection=None, models_segmentation=None):
        from . import heads
        if models_rpn or models_segmentation:
            raise NotImplementedError("the reference defines its on-line RPN and mask heads on the R-50-C4 network only")
        if models_detection:
            if self.online_box is None:
                self.online_box = heads.OnlineBoxPredictor()
            self.online_box.set_models(models_detection["classifiers"], models_detection.get("regressors"), models_detection["stats"])

    # ------------------------------------------------------------------ proposals
    @torch.no_grad()
    def proposals(self, trunk, img_size):
        """RPNPostProcessor.forward at test time over the five levels (rpn/inference.py:76-175): per level the
        pre_nms_top_n best anchors by objectness are decoded, clipped and suppressed (NMS 0.7, at most post_nms_top_n kept —
        odx_nms_first_f32 on the top-k's own order), then the fpn_post_nms_top_n best of all levels are kept in descending
        score order (select_over_all_levels, the per-image branch).  Returns (boxes, scores)."""
        be = _backend.get_backend()
        fast = _nms_takes_max_keep(be)
        # per level: the RPN head and the top-k (their sizes differ); the decoding / clipping of the selected candidates of
        # all five levels is then done ONCE on their concatenation (a level's ~20 small elementwise launches otherwise: the
        # FPN forward was launch-bound at 600 launches per image), and the suppression runs per level again
        fused = hasattr(be, "rpn_topk_decode") and trunk[0].is_cuda and self.online_rpn is None
        sel_reg, sel_anc, sel_score, counts = [], [], [], []
        lvl_boxes = []
        side = cur = None
        if fused and not torch.cuda.is_current_stream_capturing():
            # a level's selection kernel is ONE workgroup (a radix select + sort of the level's candidates: 50 .. 280 us on one of
            # the chip's 256 CUs): on a side stream it runs under the NEXT level's head convolutions instead of between them — at
            # one image per call the five selections were 0.53 of the forward's 4.8 ms (profiles/r06_summary.md)
            from . import streams as _streams
            own = _streams.distinct(1)
            if own:
                side, cur = own[0], torch.cuda.current_stream()
        for lvl, p in enumerate(trunk):
            p = p.contiguous()                   # (a level of the row-GEMM pyramid is a channels-last view: the library's route copies)
            with (contextlib.nullcontext() if p.dtype in (torch.bfloat16, torch.float16) else self._amp()):
                # (weights / biases held in the compute dtype: autocast casts an f32 parameter again on every call — 15 casts
                # per image over the five levels)
                w = self._rpn_weights(p.dtype)
                t = F.relu(F.conv2d(p, w[0], w[1], 1, 1))
                logits, deltas = F.conv2d(t, w[2], w[3]).float(), F.conv2d(t, w[4], w[5]).float()
            _, A, H, W = logits.shape
            k = min(self.pre_nms_top_n, A * H * W)
            if fused and k <= 8192:
                # a level's top-k, sorting, delta gather, decoding and clipping as ONE launch (odx_rpn_topk_decode_f32) instead of
                # ~25 tensor operations: the five levels were 125 of this forward's launches
                anchors = self._anchors(lvl, H, W, logits.device)
                if side is not None:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        b, sc, _ = be.rpn_topk_decode(logits, deltas, anchors, k, img_size, DELTA_CLAMP)
                    for t_ in (logits, deltas):
                        t_.record_stream(side)
                    for t_ in (b, sc):
                        t_.record_stream(cur)
                else:
                    b, sc, _ = be.rpn_topk_decode(logits, deltas, anchors, k, img_size, DELTA_CLAMP)
                lvl_boxes.append(b[0])
                sel_score.append(sc[0])
                counts.append(k)
                continue
            obj = logits.permute(0, 2, 3, 1).reshape(-1)
            reg = deltas.view(1, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(-1, 4)
            score, idx = obj.topk(k, sorted=True)              # (the sigmoid is monotonic: same order on the logits)
            sel_reg.append(reg[idx])
            sel_anc.append(self._anchors(lvl, H, W, reg.device)[idx])
            sel_score.append(score)
            counts.append(k)
        if side is not None:
            cur.wait_stream(side)
        if lvl_boxes and len(lvl_boxes) == len(counts):
            scores_cat, boxes_cat = torch.cat(sel_score), torch.cat(lvl_boxes)          # (already sigmoid, decoded, clipped)
        else:
            if lvl_boxes:
                raise RuntimeError("FPN proposals: levels took different routes")      # (k > 8192 on some level only: not a shipped setting)
            scores_cat = torch.cat(sel_score).sigmoid()
            boxes_cat = decode_deltas(torch.cat(sel_reg), torch.cat(sel_anc))
            boxes_cat.clamp_(min=0)
            boxes_cat[:, 0::2].clamp_(max=img_size[0] - 1)
            boxes_cat[:, 1::2].clamp_(max=img_size[1] - 1)
        if hasattr(be, "nms_batched") and boxes_cat.is_cuda and len(counts) > 1:
            # the suppression of all levels with ONE launch pair and ONE host synchronisation (odx_nms_batched_f32: independent
            # sorted box sets, sizes on the device) instead of a launch pair and a synchronisation per level — this stage is
            # what kept the host from running ahead of the GPU; "the first post_nms_top_n survivors" of a level are the same
            # boxes whether the greedy pass stops there or runs to the end
            B, Rmax = len(counts), max(counts)
            padded = torch.zeros((B, Rmax, 4), dtype=torch.float32, device=boxes_cat.device)
            at = 0
            for lvl, k in enumerate(counts):
                padded[lvl, :k] = boxes_cat[at:at + k]
                at += k
            meta = self._anchor_cache.get(("nms_meta", tuple(counts), boxes_cat.device))
            if meta is None:
                offs = np.concatenate(([0], np.cumsum(counts)[:-1]))
                meta = self._anchor_cache[("nms_meta", tuple(counts), boxes_cat.device)] = (
                    torch.tensor(counts, dtype=torch.int32, device=boxes_cat.device),
                    torch.from_numpy(offs).to(boxes_cat.device).view(B, 1) + torch.arange(Rmax, device=boxes_cat.device).view(1, Rmax))
            keep = be.nms_batched(padded, meta[0], self.rpn_nms)
            keep &= keep.cumsum(1) <= self.post_nms_top_n
            kept = meta[1].masked_select(keep)                  # level after level, descending score inside a level
        else:
            kept = []
            at = 0
            for k in counts:
                b, sc = boxes_cat[at:at + k], scores_cat[at:at + k]
                if fast:
                    keep = be.nms(b, sc, self.rpn_nms, max_keep=self.post_nms_top_n, sorted_desc=True)
                else:
                    keep = be.nms(b, sc, self.rpn_nms)[:self.post_nms_top_n]
                kept.append(keep + at)
                at += k
            kept = torch.cat(kept)
        boxes, scores = boxes_cat[kept], scores_cat[kept]
        k = min(self.fpn_post_nms_top_n, scores.numel())
        top, order = scores.topk(k, sorted=True)
        return boxes[order], top

    @torch.no_grad()
    def proposals_batch(self, trunk, img_size, t=None):
        """proposals() for the pyramids of B images of one size: [(boxes_b, scores_b)].  Per LEVEL one pass of the RPN head over
        the B maps and one odx_rpn_topk_decode_f32 launch (a workgroup per image: the five launches per image of proposals() were
        half a millisecond of each image's forward), ONE suppression launch pair over the B x 5 candidate sets and one host
        synchronisation for the group; the per-image selection over all levels as one top-k over a padded score matrix.  Same
        candidates, same suppression, same order as proposals() image after image.  Falls back to it where the fused kernels do
        not apply (an on-line RPN head, the CPU, more than 8192 candidates per level)."""
        be = _backend.get_backend()
        B = trunk[0].shape[0]
        fused = (hasattr(be, "rpn_topk_decode") and hasattr(be, "nms_batched") and trunk[0].is_cuda and self.online_rpn is None and B > 1)
        ks = []
        if fused:
            for p in trunk:
                ks.append(min(self.pre_nms_top_n, 3 * p.shape[2] * p.shape[3]))
            fused = max(ks) <= 8192 and self.rpn_logits.out_channels == 3
        if not fused:
            return [self.proposals(self.trunk_slice(trunk, b), img_size) for b in range(B)]
        dev = trunk[0].device
        L, Rmax = len(trunk), max(ks)
        cand = torch.zeros((B, L, Rmax, 4), dtype=torch.float32, device=dev)
        score = torch.full((B, L, Rmax), -1.0, dtype=torch.float32, device=dev)           # (sigmoid scores are > 0: pads sort last)
        rows_head = (self.compute_dtype is None and trunk[0].dtype == torch.float32 and not torch.is_autocast_enabled("cuda")
                     and not torch.is_grad_enabled() and hasattr(be, "conv3x3_rows") and self.rpn_conv.in_channels % 8 == 0
                     and _trunk_rows())
        dt16 = self.compute_dtype if self.compute_dtype in (torch.bfloat16, torch.float16) else None
        rows16_head = (dt16 is not None and trunk[0].dtype == dt16 and not torch.is_grad_enabled() and hasattr(be, "conv3x3_rows16")
                       and self.rpn_conv.in_channels % 8 == 0 and _trunk_rows())
        side = cur = None
        if (rows_head or rows16_head) and not torch.cuda.is_current_stream_capturing():
            from . import streams as _streams
            own = _streams.distinct(1)
            if own:
                side, cur = own[0], torch.cuda.current_stream()
                side.wait_stream(cur)                         # (cand / score were filled on this stream)
                cand.record_stream(side)
                score.record_stream(side)
        for lvl, p in enumerate(trunk):
            if rows16_head:
                # (the same on 16-bit rows: odx_gemm_b16_taps / odx_gemm_b16, the outputs in f32)
                Bn, C, H, W = p.shape
                wp = self._rpn_rows16_weights(be, dt16)
                x16 = be.rows16(p.permute(0, 2, 3, 1).reshape(Bn * H * W, C), dt16, zero_row=True)
                a = be.conv3x3_rows16(x16, Bn, H, W, wp[0], bias=wp[1], relu=True)
                o = be.gemm_b16(a, wp[2], bias=wp[3], out_f32=True).view(Bn, H, W, -1)
                A = self.rpn_logits.out_channels
                logits, deltas = o[..., :A].permute(0, 3, 1, 2).contiguous(), o[..., A:].permute(0, 3, 1, 2).contiguous()
            elif rows_head:
                # the RPN head of a level for the whole group as two products over the level's NHWC rows (the 3 x 3 convolution with
                # its taps gathered in the operand loads, the two 1 x 1 outputs as one product of 15 columns): the convolution
                # library's five 3 x 3 convolutions were a fifth of the group forward's device time
                Bn, C, H, W = p.shape
                wp = self._rpn_rows_weights(be)
                a = be.conv3x3_rows(p.permute(0, 2, 3, 1).reshape(Bn * H * W, C), Bn, H, W, wp[0], bias=wp[1], relu=True)
                o = be.gemm_h2(be.packed(a), wp[2], bias=wp[3]).view(Bn, H, W, -1)
                A = self.rpn_logits.out_channels
                logits, deltas = o[..., :A].permute(0, 3, 1, 2).contiguous(), o[..., A:].permute(0, 3, 1, 2).contiguous()
            else:
                p = p.contiguous()
                with (contextlib.nullcontext() if p.dtype in (torch.bfloat16, torch.float16) else self._amp()):
                    w = self._rpn_weights(p.dtype)
                    a = F.relu(F.conv2d(p, w[0], w[1], 1, 1))
                    logits, deltas = F.conv2d(a, w[2], w[3]).float(), F.conv2d(a, w[4], w[5]).float()
            _, A, H, W = logits.shape
            anchors = self._anchors(lvl, H, W, dev)
            if side is not None:
                # a level's selection kernel is ONE workgroup per image (a sort of its candidates: 50 .. 280 us on eight of the
                # chip's 256 CUs): on a stream of its own it runs under the NEXT level's head products instead of between them
                # (0.6 ms of the five selections of a group of 8 were serial)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    b, sc, _ = be.rpn_topk_decode(logits, deltas, anchors, ks[lvl], img_size, DELTA_CLAMP)
                    cand[:, lvl, :ks[lvl]] = b
                    score[:, lvl, :ks[lvl]] = sc
                for t_ in (logits, deltas):
                    t_.record_stream(side)
                continue
            b, sc, _ = be.rpn_topk_decode(logits, deltas, anchors, ks[lvl], img_size, DELTA_CLAMP)
            cand[:, lvl, :ks[lvl]] = b
            score[:, lvl, :ks[lvl]] = sc
        if side is not None:
            cur.wait_stream(side)
        # (the candidate counts of the B x 5 sets depend on the image size only: uploaded once — a host-to-device copy of pageable
        # memory waits for the stream, here in the middle of the forward)
        ckey = (B, tuple(ks), str(dev))
        hit = self._packed.get("nms_counts")
        if hit is None or hit[0] != ckey:
            hit = self._packed["nms_counts"] = (ckey, torch.tensor(ks * B, dtype=torch.int32, device=dev))
        counts = hit[1]
        keep = be.nms_batched(cand.view(B * L, Rmax, 4), counts, self.rpn_nms).view(B, L, Rmax)
        keep &= keep.cumsum(2) <= self.post_nms_top_n
        flat = torch.where(keep, score, torch.full_like(score, -1.0)).view(B, L * Rmax)
        k = min(self.fpn_post_nms_top_n, L * Rmax)
        top, order = flat.topk(k, dim=1, sorted=True)              # level after level inside equal scores: the stable order of proposals()
        boxes = cand.view(B, L * Rmax, 4).gather(1, order.unsqueeze(2).expand(B, k, 4))
        n = (top > 0).sum(dim=1).tolist()                          # survivors per image (the group's one synchronisation)
        return [(boxes[b, :n[b]], top[b, :n[b]]) for b in range(B)]

    def _rpn_rows_weights(self, be):
        """The RPN head's weights as packed GEMM operands: (3 x 3 convolution (out, ky kx in), its bias, the two 1 x 1 outputs stacked
        (A + 4 A, C), their biases); remade when a parameter is replaced or written in place."""
        ps = (self.rpn_conv.weight, self.rpn_conv.bias, self.rpn_logits.weight, self.rpn_logits.bias, self.rpn_deltas.weight, self.rpn_deltas.bias)
        key = ("rows",) + tuple((q.data_ptr(), q._version) for q in ps)
        hit = self._packed.get("rpn_rows")
        if hit is None or hit[0] != key:
            cv, lg, dl = self.rpn_conv, self.rpn_logits, self.rpn_deltas
            hit = self._packed["rpn_rows"] = (key, (
                be.packed(cv.weight.detach().float().permute(0, 2, 3, 1).reshape(cv.out_channels, -1).contiguous()), cv.bias.detach().float().contiguous(),
                be.packed(torch.cat((lg.weight.detach().float().reshape(lg.out_channels, -1), dl.weight.detach().float().reshape(dl.out_channels, -1)), dim=0).contiguous()),
                torch.cat((lg.bias.detach().float(), dl.bias.detach().float())).contiguous()))
        return hit[1]

    def _rpn_rows16_weights(self, be, dt):
        """_rpn_rows_weights for a 16-bit forward: Rows16 operands of type dt."""
        ps = (self.rpn_conv.weight, self.rpn_conv.bias, self.rpn_logits.weight, self.rpn_logits.bias, self.rpn_deltas.weight, self.rpn_deltas.bias)
        key = ("rows16", dt) + tuple((q.data_ptr(), q._version) for q in ps)
        hit = self._packed.get("rpn_rows16")
        if hit is None or hit[0] != key:
            cv, lg, dl = self.rpn_conv, self.rpn_logits, self.rpn_deltas
            hit = self._packed["rpn_rows16"] = (key, (
                be.rows16(cv.weight.detach().permute(0, 2, 3, 1).reshape(cv.out_channels, -1).to(dt).contiguous(), dt), cv.bias.detach().float().contiguous(),
                be.rows16(torch.cat((lg.weight.detach().reshape(lg.out_channels, -1), dl.weight.detach().reshape(dl.out_channels, -1)), dim=0).to(dt).contiguous(), dt),
                torch.cat((lg.bias.detach().float(), dl.bias.detach().float())).contiguous()))
        return hit[1]

    def _rpn_weights(self, dtype):
        """The RPN head's three convolutions' weights and biases in `dtype` (cached; dropped with the packed fc weights when
        the parameters change)."""
        ps = (self.rpn_conv.weight, self.rpn_conv.bias, self.rpn_logits.weight, self.rpn_logits.bias, self.rpn_deltas.weight, self.rpn_deltas.bias)
        key = (dtype,) + tuple((q.data_ptr(), q._version) for q in ps)
        hit = self._packed.get("rpn")
        if hit is None or hit[0] != key:
            hit = self._packed["rpn"] = (key, tuple(q.detach().to(dtype) for q in ps))
        return hit[1]

    def _anchors(self, lvl, H, W, device):
        """The anchors of a level's H x W grid (cached per shape: a function of the image size only)."""
        key = (lvl, H, W, str(device))
        a = self._anchor_cache.get(key)
        if a is None:
            if len(self._anchor_cache) > 64:
                self._anchor_cache.clear()
            a = self._anchor_cache[key] = grid_anchors(H, W, self.strides[lvl], self.cells[lvl].to(device))
        return a

    # ------------------------------------------------------------------ RoI features
    def _fc(self, be, name, x, layer, nhwc=None):
        """relu(x W' + b) on the split-f16 tile cores (f32 accuracy), the weight packed once.  nhwc = (C, r): x's columns are
        ordered (ph, pw, c) instead of the layer's (c, ph, pw) — the weight's columns are permuted to match when it is packed."""
        # keyed on the parameter's storage and in-place version counter as well: an optimiser step, a copy_ into the weight
        # or a re-assigned Parameter must not be multiplied with the packing of the old values
        key = (layer.weight.data_ptr(), layer.weight._version)
        hit = self._packed.get(name)
        if hit is None or hit[0] != key:
            w = layer.weight.detach().float()
            if nhwc is not None:
                C, r = nhwc
                w = w.view(w.shape[0], C, r, r).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
            hit = self._packed[name] = (key, be.packed(w.contiguous()))
        return be.gemm_h2(be.packed(x), hit[1], bias=layer.bias.detach().float(), relu=True)

    @torch.no_grad()
    def roi_features(self, trunk, boxes, batch_idx=None):
        """(R, mlp_dim): Pooler over P2..P5 (one HIP launch for all levels) -> flatten -> fc6 -> ReLU -> fc7 -> ReLU.
        batch_idx (R,): the image of a batched pyramid each box is pooled from (extract.forward_batch: the RoIs of a group of
        images go through fc6 / fc7 as ONE row matrix)."""
        be = _backend.get_backend()
        first = (torch.zeros((boxes.shape[0], 1), device=boxes.device) if batch_idx is None
                 else batch_idx.to(device=boxes.device, dtype=boxes.dtype).view(-1, 1))
        rois = torch.cat((first, boxes), dim=1)
        t0 = trunk[0]
        if (t0.is_cuda and t0.dtype in (torch.float32, torch.bfloat16, torch.float16) and not t0.is_contiguous()
                and t0.is_contiguous(memory_format=torch.channels_last)
                and hasattr(be, "roi_align_fpn_rows") and hasattr(be, "gemm_h2") and boxes.shape[0] > 0 and t0.shape[1] % 4 == 0):
            # the pyramid is the row GEMMs' NHWC rows: pooled from them as they are, the crops flattened in (ph, pw, c) order and
            # fc6's weight columns permuted to match (once) — no NCHW copy of the levels, 16-byte reads of contiguous channels
            x = be.roi_align_fpn_rows(list(trunk[:4]), rois, self.pool_scales, (self.resolution, self.resolution), self.sampling_ratio)
            return self._fc(be, "fc7", self._fc(be, "fc6_nhwc", x, self.fc6, nhwc=(t0.shape[1], self.resolution)), self.fc7)
        crops = be.roi_align_fpn([p.contiguous() for p in trunk[:4]], rois, self.pool_scales, (self.resolution, self.resolution), self.sampling_ratio)
        x = crops.reshape(crops.shape[0], -1)
        if x.shape[0] == 0:
            return x.new_zeros((0, self.mlp_dim))
        if x.is_cuda and hasattr(be, "gemm_h2"):
            # (also under bf16 autocast of the trunk: the pooled crops are f32 either way, and the split cores run these two
            # products faster than the library's bf16 GEMMs with the casts around them — 5.7 -> 4.9 ms per image)
            return self._fc(be, "fc7", self._fc(be, "fc6", x, self.fc6), self.fc7)
        with self._amp():
            return F.relu(self.fc7(F.relu(self.fc6(x)))).float()

    @torch.no_grad()
    def roi_head_maps(self, trunk, boxes, batch_idx=None):
        """The features as (R, D, 1, 1) maps: what extract's harvest loop and detect() average over the last two axes."""
        return self.roi_features(trunk, boxes, batch_idx)[:, :, None, None]

    @torch.no_grad()
    def forward(self, image, gt_boxes=None):
        """image (1, 3, H, W) already normalised / resized; returns (boxes (R, 4), feats (R, 1024), pyramid) with the
        ground-truth boxes prepended to the proposals (generalized_rcnn_getProposals.py:90-96)."""
        trunk = self.c4(image)
        img_size = (image.shape[3], image.shape[2])
        boxes, _ = self.proposals(trunk, img_size)
        if gt_boxes is not None and len(gt_boxes):
            boxes = torch.cat((gt_boxes.to(boxes.device).float(), boxes), dim=0)
        return boxes, self.roi_features(trunk, boxes), trunk
