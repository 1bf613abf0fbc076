"""The R-50-FPN variant of the forward (odx/fpn.py): level mapping + multi-level RoIAlign against the oracle
(oracle/roi_ref.py, parity unpinned: maskrcnn_benchmark's Pooler is not vendored), the whole forward on the MI355X against
a plain-torch CPU restatement of the same network, and the detector harvest / on-line heads on its D = 1024 features."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

import odx
from odx.extract import DetectorFeatureExtractor, OnlineFeatureExtractor, decode_deltas, grid_anchors
from odx.fpn import OnlineDetectionModelFPN
from tests.test_extract import _plain_roi_align, _samples


def test_fpn_levels_follow_the_level_mapper():
    """floor(4 + log2(sqrt(area) / 224 + 1e-6)) clamped to the pyramid, +1 areas: the canonical 224-pixel box sits on the
    stride-16 level (index 2 of P2..P5), halving / doubling the side moves one level, tiny and huge boxes clamp."""
    from oracle import roi_ref
    scales = (0.25, 0.125, 0.0625, 0.03125)

    def box(side):
        return [10.0, 20.0, 10.0 + side - 1, 20.0 + side - 1]         # +1 convention: area = side^2
    lv = roi_ref.fpn_levels(np.array([box(224), box(223), box(112), box(111), box(448), box(447), box(4), box(2000), box(56), box(55)]),
                            scales)
    assert lv.tolist() == [2, 1, 1, 0, 3, 2, 0, 3, 0, 0]


@pytest.mark.gpu
def test_roi_align_fpn_matches_oracle():
    """odx_roi_align_fpn_f32 (one launch, level chosen in the kernel) against the oracle's per-level loop: levels
    identical, pooled values to f32 rounding; boxes of every level, on level boundaries, degenerate and partly outside."""
    from oracle import roi_ref
    odx.set_backend(None)
    be = odx.get_backend()
    rng = np.random.default_rng(7)
    C, H0, W0 = 20, 48, 64
    scales = (0.25, 0.125, 0.0625, 0.03125)
    feats = [rng.standard_normal((1, C, H0 >> k, W0 >> k)).astype(np.float32) for k in range(4)]
    side = np.concatenate([rng.uniform(4, 400, 40), [224, 223, 112, 111, 448, 447, 56, 1]])
    x1 = rng.uniform(-20, 200, side.size)
    y1 = rng.uniform(-20, 150, side.size)
    asp = rng.uniform(0.5, 2.0, side.size)
    boxes = np.stack([x1, y1, x1 + side * asp - 1, y1 + side / asp - 1], 1).astype(np.float32)
    boxes[-8:, 2] = boxes[-8:, 0] + side[-8:] - 1                     # exact squares on the level boundaries
    boxes[-8:, 3] = boxes[-8:, 1] + side[-8:] - 1
    rois = np.concatenate([np.zeros((len(boxes), 1), np.float32), boxes], 1)
    out, lv = be.roi_align_fpn([torch.from_numpy(f) for f in feats], torch.from_numpy(rois), scales, (7, 7), 2, return_levels=True)
    want_lv = roi_ref.fpn_levels(boxes, scales)
    assert lv.cpu().numpy().tolist() == want_lv.tolist()
    assert set(want_lv.tolist()) == {0, 1, 2, 3}
    want = roi_ref.roi_align_fpn(feats, rois, scales, (7, 7), 2)
    assert out.shape == (len(boxes), C, 7, 7)
    assert np.abs(out.cpu().numpy() - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    # adaptive sampling (sampling_ratio 0) and a 14 x 14 grid through the same kernel
    out0 = be.roi_align_fpn([torch.from_numpy(f) for f in feats], torch.from_numpy(rois[:12]), scales, (14, 14), 0)
    want0 = roi_ref.roi_align_fpn(feats, rois[:12], scales, (14, 14), 0)
    assert np.abs(out0.cpu().numpy() - want0).max() < 2e-5 * max(1.0, np.abs(want0).max())
    assert be.roi_align_fpn([torch.from_numpy(f) for f in feats], torch.zeros((0, 5)), scales, (7, 7), 2).shape == (0, C, 7, 7)


def test_fpn_harvest_loop_on_cpu_with_oracle_backend():
    from tests.oracle_backend import OracleBackend
    odx.set_backend(OracleBackend(np.float64))
    try:
        model = OnlineDetectionModelFPN(width=4, fpn_channels=8, mlp_dim=24, pre_nms_top_n=30, post_nms_top_n=10, fpn_post_nms_top_n=12,
                                        resolution=3).eval()
        assert model.feat_dim == 24
        img = _samples(1, 64, 96, 2)[0][0]
        pyr = model.c4(img)
        assert [tuple(p.shape[1:]) for p in pyr] == [(8, 16, 24), (8, 8, 12), (8, 4, 6), (8, 2, 3), (8, 1, 2)]
        boxes, scores = model.proposals(pyr, (96, 64))
        assert len(boxes) <= 12 and bool((scores[:-1] >= scores[1:]).all())
        ex = DetectorFeatureExtractor(model, num_classes=2, iterations=2, batch_size=8)
        torch.manual_seed(0)
        neg, pos, COXY = ex.train(_samples(3, 64, 96, 2))
        assert len(neg) == 2 and all(b.shape[1] == 24 for n in neg for b in n) and pos[0].shape[1] == 24
        assert sum(len(p) for p in pos) == 1 + 2 + 1 and COXY["X"].shape[1] == 24
        out = OnlineFeatureExtractor(model, 2, parts=("detector",), det=dict(iterations=2, batch_size=8)).train(_samples(3, 64, 96, 2))
        assert set(out) == {"detector"} and out["detector"][2]["X"].shape[1] == 24
    finally:
        odx.set_backend(None)


def test_fpn_load_state_dict_drops_the_packed_weights():
    """load_state_dict copies parameters in place without passing through _apply: the packed fc6 / fc7 operands (derived
    data of the OLD weights) and the anchor cache must not survive it (advisor finding, round 3)."""
    model = OnlineDetectionModelFPN(width=4, fpn_channels=8, mlp_dim=24, resolution=3).eval()
    other = OnlineDetectionModelFPN(width=4, fpn_channels=8, mlp_dim=24, resolution=3, seed=5).eval()
    model._packed["fc6"] = ("stale", object())
    model._anchor_cache[(0, 1, 1, "cpu")] = object()
    model.load_state_dict(other.state_dict())
    assert not model._packed and not model._anchor_cache
    assert torch.equal(model.fc6.weight, other.fc6.weight)


@pytest.mark.gpu
def test_fpn_features_follow_a_loaded_checkpoint():
    """Forward, load another checkpoint, forward again: the RoI features are those of a fresh model holding the new weights
    (the split-core fc6 / fc7 multiply by the weights of the checkpoint, not by a packing made before it was loaded); the
    same after an in-place change of one weight (the cache is keyed on the parameter's version counter)."""
    odx.set_backend(None)
    kw = dict(width=8, fpn_channels=16, mlp_dim=64, pre_nms_top_n=100, post_nms_top_n=20, fpn_post_nms_top_n=40)
    model = OnlineDetectionModelFPN(seed=1, **kw).eval().cuda()
    donor = OnlineDetectionModelFPN(seed=2, **kw).eval()
    fresh = OnlineDetectionModelFPN(seed=2, **kw).eval().cuda()
    img = torch.randn(1, 3, 192, 256, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        # (the pooled features of FIXED boxes on a FIXED pyramid: only the RoIAlign launch and the two split-core products, all
        # deterministic, so the comparison is bit for bit; the trunk's convolutions are allowed run-to-run noise)
        trunk = fresh.c4(img)
        boxes = torch.tensor([[10.0, 20.0, 120.0, 150.0], [0.0, 0.0, 255.0, 191.0], [60.0, 30.0, 90.0, 70.0], [5.0, 100.0, 200.0, 180.0]]).cuda()
        f0 = model.roi_features(trunk, boxes)                   # fills the packed-weight cache with seed 1's weights
        model.load_state_dict(donor.state_dict())
        f1 = model.roi_features(trunk, boxes)
        f2 = fresh.roi_features(trunk, boxes)
        assert torch.equal(f1, f2) and not torch.equal(f0, f1)  # the old packing would have given the old features
        model.fc7.weight.mul_(0.5)
        fresh.fc7.weight.mul_(0.5)
        fresh._packed.clear()
        f3 = model.roi_features(trunk, boxes)
        assert torch.equal(f3, fresh.roi_features(trunk, boxes)) and not torch.equal(f3, f1)


@pytest.mark.gpu
def test_fpn_forward_gpu_equals_plain_torch_cpu():
    """The FPN forward on the MI355X — folded batch norm, pyramid, per-level top-k / decode / early-stopping NMS, the
    selection over all levels, ONE multi-level RoIAlign launch, fc6 / fc7 on the split-f16 tile cores — against a plain
    f32 torch restatement of the same network on the CPU (layer-by-layer convolution -> frozen batch norm -> ReLU, greedy
    NMS over all candidates of a level cut afterwards, the level mapper and a per-RoI RoIAlign, two nn.Linear): same
    proposals in the same order, boxes to 1e-3 px, RoI features within 1e-4 relative."""
    import copy
    from oracle import roi_ref
    odx.set_backend(None)
    torch.manual_seed(3)
    model = OnlineDetectionModelFPN(width=16, fpn_channels=32, mlp_dim=96, pre_nms_top_n=200, post_nms_top_n=40, fpn_post_nms_top_n=170,
                                    seed=9).eval()          # 170 > 4 x 40 + P6's few: proposals of every level survive the selection
    for m in model.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
            m.weight.data.normal_(1, 0.1)
            m.bias.data.normal_(0, 0.1)
    model.rpn_logits.weight.data.normal_(0, 0.3)          # well separated scores (see the C4 test)
    model.rpn_deltas.weight.data.normal_(0, 0.05)
    ref = copy.deepcopy(model)
    model = model.cuda()
    img = torch.randn(1, 3, 320, 448)
    gt = torch.tensor([[30.0, 40.0, 250.0, 300.0], [100.0, 60.0, 140.0, 110.0]])
    with torch.no_grad():
        boxes, feats, pyr = model(img.cuda(), gt)

        def block(b, x):
            idn = x if b.down is None else b.down[1](b.down[0](x))
            y = Fn.relu(b.bn1(b.conv1(x)))
            y = Fn.relu(b.bn2(b.conv2(y)))
            return Fn.relu(b.bn3(b.conv3(y)) + idn)
        bb = ref.backbone
        x = Fn.max_pool2d(Fn.relu(bb.bn1(bb.conv1(img))), 3, 2, 1)
        cs = []
        for stage in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
            for b in stage:
                x = block(b, x)
            cs.append(x)
        last = ref.fpn.inner[3](cs[3])
        ps = [ref.fpn.layer[3](last)]
        for k in (2, 1, 0):
            lat = ref.fpn.inner[k](cs[k])
            last = lat + Fn.interpolate(last, size=lat.shape[-2:], mode="nearest")
            ps.insert(0, ref.fpn.layer[k](last))
        ps.append(ps[-1][:, :, ::2, ::2])
        cand_b, cand_s = [], []
        for lvl, p in enumerate(ps):
            t = Fn.relu(ref.rpn_conv(p))
            logits, deltas = ref.rpn_logits(t), ref.rpn_deltas(t)
            _, A, H, W = logits.shape
            obj = logits.permute(0, 2, 3, 1).reshape(-1).sigmoid()
            reg = deltas.view(1, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(-1, 4)
            score, idx = obj.topk(min(ref.pre_nms_top_n, obj.numel()), sorted=True)
            anchors = grid_anchors(H, W, ref.strides[lvl], ref.cells[lvl])
            cand = decode_deltas(reg[idx], anchors[idx])
            cand[:, 0::2] = cand[:, 0::2].clamp(0, img.shape[3] - 1)
            cand[:, 1::2] = cand[:, 1::2].clamp(0, img.shape[2] - 1)
            keep = torch.from_numpy(roi_ref.nms(cand.numpy(), score.numpy(), ref.rpn_nms))[: ref.post_nms_top_n]
            cand_b.append(cand[keep])
            cand_s.append(score[keep])
        allb, alls = torch.cat(cand_b), torch.cat(cand_s)
        top, order = alls.topk(min(ref.fpn_post_nms_top_n, alls.numel()), sorted=True)
        boxes_ref = torch.cat((gt, allb[order]), dim=0)
        lv = roi_ref.fpn_levels(boxes_ref.numpy(), ref.pool_scales)
        crops = torch.zeros((len(boxes_ref), ps[0].shape[1], ref.resolution, ref.resolution))
        for l in range(4):
            I = np.flatnonzero(lv == l)
            if len(I):
                # sampling_ratio 2: a fixed 2 x 2 grid per bin (the plain helper's adaptive grid pinned to 2)
                crops[I] = _plain_roi_align_fixed(ps[l][0], boxes_ref[I], ref.pool_scales[l], ref.resolution, 2)
        feats_ref = Fn.relu(ref.fc7(Fn.relu(ref.fc6(crops.reshape(len(crops), -1)))))
    for a, b in zip(pyr, ps):
        assert float((a.cpu() - b).abs().max()) < 1e-4 * float(b.abs().max())
    assert len(set(lv.tolist())) >= 3                      # the proposals really are spread over the pyramid
    assert boxes.shape == boxes_ref.shape, (boxes.shape, boxes_ref.shape)
    assert float((boxes.cpu() - boxes_ref).abs().max()) < 1e-3
    rel = float((feats.cpu() - feats_ref).norm(dim=1).max() / feats_ref.norm(dim=1).max())
    assert feats.shape == feats_ref.shape == (len(boxes_ref), 96) and rel < 1e-4, rel


def _bf16_trunk_restatement(ref, img):
    """The 16-bit trunk + pyramid in plain torch on the CPU, stated the way the bf16 forward is defined: every convolution's
    weight is the frozen batch norm folded in f32 and rounded ONCE to bf16, activations are bf16 between layers, a
    convolution's bias, the residual and the ReLU are separate bf16 operations (each addition rounded), the pyramid's
    convolutions carry their bias inside the call."""
    bf = torch.bfloat16

    def cba(conv, bn, x, residual=None, relu=True):
        scale = bn.weight * bn.running_var.rsqrt()
        w = (conv.weight * scale.view(-1, 1, 1, 1)).to(bf)
        b = (bn.bias - bn.running_mean * scale).to(bf)
        y = Fn.conv2d(x, w, None, conv.stride, conv.padding) + b.view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        return Fn.relu(y) if relu else y

    def block(b, x):
        idn = x if b.down is None else cba(b.down[0], b.down[1], x, relu=False)
        y = cba(b.conv1, b.bn1, x)
        y = cba(b.conv2, b.bn2, y)
        return cba(b.conv3, b.bn3, y, residual=idn)
    bb = ref.backbone
    x = Fn.max_pool2d(cba(bb.conv1, bb.bn1, img.to(bf)), 3, 2, 1)
    cs = []
    for stage in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        for b in stage:
            x = block(b, x)
        cs.append(x)

    def conv(c, x):
        return Fn.conv2d(x, c.weight.to(bf), c.bias.to(bf), c.stride, c.padding)
    last = conv(ref.fpn.inner[3], cs[3])
    ps = [conv(ref.fpn.layer[3], last)]
    for k in (2, 1, 0):
        lat = conv(ref.fpn.inner[k], cs[k])
        last = lat + Fn.interpolate(last, size=lat.shape[-2:], mode="nearest")
        ps.insert(0, conv(ref.fpn.layer[k], last))
    ps.append(ps[-1][:, :, ::2, ::2])
    return ps


@pytest.mark.gpu
def test_fpn_forward_in_bf16_config2_as_stated():
    """`OnlineDetectionModelFPN(compute_dtype=torch.bfloat16)` — BASELINE config 2's forward as stated, the form the bench
    line times (round-4 review: exercised by no test).  (i) The pyramid it hands out is bf16 and within bf16 rounding of
    the f32 model's, level by level, AND of a plain-torch CPU restatement of the 16-bit network (folded weights rounded once,
    bf16 activations); (ii) given the GPU's own bf16 pyramid, the rest of the forward is f32 work and must equal plain torch
    on those maps: the multi-level RoIAlign + fc6 / fc7 features of fixed boxes within 1e-4 relative; (iii) the same boxes
    on the two models' pyramids give features within the bf16 bound; (iv) the whole forward runs — f32 features of the
    right shape, finite, proposals sorted, most of them the f32 model's proposals."""
    import copy
    from oracle import roi_ref
    odx.set_backend(None)
    kw = dict(width=16, fpn_channels=32, mlp_dim=96, pre_nms_top_n=200, post_nms_top_n=40, fpn_post_nms_top_n=170, seed=9)
    torch.manual_seed(3)
    m32 = OnlineDetectionModelFPN(**kw).eval()
    for m in m32.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
            m.weight.data.normal_(1, 0.1)
            m.bias.data.normal_(0, 0.1)
    m32.rpn_logits.weight.data.normal_(0, 0.3)
    m32.rpn_deltas.weight.data.normal_(0, 0.05)
    ref = copy.deepcopy(m32)                                      # stays on the CPU
    m16 = OnlineDetectionModelFPN(compute_dtype=torch.bfloat16, **kw).eval()
    m16.load_state_dict(m32.state_dict())
    m32, m16 = m32.cuda(), m16.cuda()
    img = torch.randn(1, 3, 320, 448)
    gt = torch.tensor([[30.0, 40.0, 250.0, 300.0], [100.0, 60.0, 140.0, 110.0]])
    fixed = torch.tensor([[10.0, 20.0, 120.0, 150.0], [0.0, 0.0, 447.0, 319.0], [60.0, 30.0, 90.0, 70.0], [5.0, 100.0, 200.0, 180.0],
                          [200.0, 50.0, 420.0, 300.0], [300.0, 200.0, 330.0, 240.0]])

    def rel(a, b):
        return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
    with torch.no_grad():
        for _ in range(3):                                        # (the third call replays the captured graph: same numbers)
            p16 = m16.c4(img.cuda())
        p32 = m32.c4(img.cuda())
        p16_eager = m16._c4_eager(img.cuda())
        assert all(a.dtype == torch.bfloat16 for a in p16) and all(a.dtype == torch.float32 for a in p32)
        assert all(rel(a, b) < 1e-2 for a, b in zip(p16, p16_eager))               # graph replay = launch by launch (library noise only)
        pcpu = _bf16_trunk_restatement(ref, img)
        errs32 = [rel(a, b) for a, b in zip(p16, p32)]
        errs16 = [rel(a, b) for a, b in zip(p16, pcpu)]
        print("bf16 pyramid vs f32 pyramid:", errs32, "vs the plain-torch bf16 restatement:", errs16)
        assert max(errs32) < 3e-2 and max(errs16) < 3e-2                            # (i)
        f16 = m16.roi_features(p16, fixed.cuda())
        assert f16.dtype == torch.float32
        maps = [p.float().cpu().numpy() for p in p16[:4]]
        rois = np.concatenate([np.zeros((len(fixed), 1), np.float32), fixed.numpy()], 1)
        crops = torch.from_numpy(roi_ref.roi_align_fpn(maps, rois, ref.pool_scales, (ref.resolution, ref.resolution), ref.sampling_ratio))
        want = Fn.relu(ref.fc7(Fn.relu(ref.fc6(crops.reshape(len(fixed), -1).float()))))
        assert float((f16.cpu() - want).norm(dim=1).max() / want.norm(dim=1).max()) < 1e-4      # (ii)
        assert rel(f16, m32.roi_features(p32, fixed.cuda())) < 3e-2                 # (iii)
        b16, feats, _ = m16(img.cuda(), gt)                                         # (iv)
        b32, _, _ = m32(img.cuda(), gt)
        assert feats.dtype == torch.float32 and feats.shape == (b16.shape[0], 96) and bool(torch.isfinite(feats).all())
        assert torch.equal(b16[:2].cpu(), gt)
        _, sc = m16.proposals(p16, (448, 320))
        assert bool((sc[:-1] >= sc[1:]).all())
        best = np.array([roi_ref.compute_overlap(b, b32.cpu().numpy()).max() for b in b16.cpu().numpy()])
        frac = float((best > 0.9).mean())
        print("bf16 proposals with an f32 proposal at IoU > 0.9: %.2f" % frac)
        assert frac > 0.5


def _plain_roi_align_fixed(feat, boxes, scale, P, g):
    """_plain_roi_align with a fixed g x g sampling grid per bin (sampling_ratio = g)."""
    C, H, W = feat.shape
    out = torch.zeros((boxes.shape[0], C, P, P))
    for r, b in enumerate(boxes):
        x1, y1, x2, y2 = (b * scale).tolist()
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        ys = y1 + (torch.arange(P)[:, None] + (torch.arange(g)[None, :] + 0.5) / g) * (rh / P)
        xs = x1 + (torch.arange(P)[:, None] + (torch.arange(g)[None, :] + 0.5) / g) * (rw / P)

        def taps(v, size):
            ok = (v >= -1.0) & (v <= size)
            v = v.clamp(min=0.0)
            lo = v.floor().long()
            top = lo >= size - 1
            lo = torch.where(top, torch.full_like(lo, size - 1), lo)
            hi = torch.where(top, lo, lo + 1)
            frac = torch.where(top, torch.zeros_like(v), v - lo.float())
            return lo, hi, frac, ok
        yl, yh, fy, oky = taps(ys.reshape(-1), H)
        xl, xh, fx, okx = taps(xs.reshape(-1), W)
        wy0, wy1 = ((1 - fy) * oky)[None, :, None], (fy * oky)[None, :, None]
        wx0, wx1 = ((1 - fx) * okx)[None, None, :], (fx * okx)[None, None, :]
        v = (wy0 * wx0 * feat[:, yl][:, :, xl] + wy0 * wx1 * feat[:, yl][:, :, xh]
             + wy1 * wx0 * feat[:, yh][:, :, xl] + wy1 * wx1 * feat[:, yh][:, :, xh])
        out[r] = v.view(C, P, g, P, g).mean(dim=(2, 4))
    return out


@pytest.mark.gpu
def test_fpn_features_through_the_online_pipeline(tmp_path):
    """FPN features (D = mlp_dim) through the rest of the path: detector harvest -> statistics -> FALKON minibootstrap ->
    RLS box regressors -> on-line box head -> detect() on a new image, all through libodx."""
    import yaml
    from odx.extract import detect
    from tests import dropin
    odx.set_backend(None)
    C = 3
    model = OnlineDetectionModelFPN(width=8, fpn_channels=32, mlp_dim=128, pre_nms_top_n=300, post_nms_top_n=60, fpn_post_nms_top_n=80).cuda().eval()
    samples = _samples(8, 192, 256, C, seed=3)
    ex = OnlineFeatureExtractor(model, C, parts=("detector",), det=dict(iterations=3, batch_size=40))
    torch.manual_seed(0)
    neg, pos, COXY = ex.train(samples)["detector"]
    assert COXY["X"].shape[1] == 128 and all(b.shape[1] == 128 for n in neg for b in n)
    cfg = {"NUM_CLASSES": C + 1,
           "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                        "CLASSIFIER": {"lambda": 0.001, "sigma": 10, "M": 60, "kernel_type": "gauss"}},
           "REGION_REFINER": {"opts": {"lambda": 10.0}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(["_background_", "a", "b", "c"])}}
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    u = dropin.load("py_od_utils")
    with redirect_stdout(io.StringIO()):
        stats = u.computeFeatStatistics_torch(pos, neg, features_dim=128, pos_fraction=0.8)
        clf = dropin.load("FALKONWrapper_with_centers_selection_incore").FALKONWrapper(cfg_path=path)
        orc = dropin.load("OnlineRegionClassifier_incore").OnlineRegionClassifier(clf, pos, neg, stats, cfg_path=path)
        models = orc.trainRegionClassifier()
        regs = dropin.load("region_refiner").RegionRefiner(path).trainRegionRefiner(COXY)
    assert len(models) == C and any(m is not None for m in models)
    model.update_model(models_detection={"classifiers": models, "regressors": regs, "stats": stats})
    res, boxes = detect(model, samples[0][0].cuda(), (256, 192), -2.0, 0.3, 50)
    assert boxes.shape[1] == 4 and (res is None or (len(res["boxes"]) <= 50 and bool(torch.isfinite(res["scores"]).all())))


@pytest.mark.gpu
def test_fpn_proposals_of_a_group_equal_the_per_image_stage():
    """OnlineDetectionModelFPN.proposals_batch (per level one RPN-head pass and one odx_rpn_topk_decode_f32 launch for the B
    images, one suppression launch pair over the B x 5 candidate sets, the selection over all levels as one top-k over a padded
    score matrix) = proposals() image after image: same boxes, same scores, same order; and forward_batch on the pyramid model
    hands every image what forward() gives it alone."""
    import odx
    from odx.extract import forward_batch
    odx.set_backend(None)
    odx.get_backend()
    torch.manual_seed(3)
    model = OnlineDetectionModelFPN(width=16, fpn_channels=32, mlp_dim=64, pre_nms_top_n=300, post_nms_top_n=100, fpn_post_nms_top_n=150).cuda().eval()
    model.rpn_logits.weight.data.normal_(0, 0.3)          # well separated objectness, candidates that move and overlap
    model.rpn_deltas.weight.data.normal_(0, 0.05)
    x = torch.randn(3, 3, 192, 256).cuda()
    with torch.no_grad():
        trunk = model.c4(x)
        one = [model.proposals(model.trunk_slice(trunk, b), (256, 192)) for b in range(3)]
        grp = model.proposals_batch(trunk, (256, 192))
        for (b1, s1), (b2, s2) in zip(one, grp):
            # (the convolution library may run the head's convolutions of three maps by another algorithm than those of one)
            assert b1.shape == b2.shape, (b1.shape, b2.shape)
            assert float((b1 - b2).abs().max()) < 1e-3 and float((s1 - s2).abs().max()) < 1e-6, (float((b1 - b2).abs().max()), float((s1 - s2).abs().max()))
        per, _, _, _ = forward_batch(model, x, [None, torch.tensor([[10.0, 20.0, 120.0, 150.0]]), None])
        boxes, feats, _ = model(x[1:2], torch.tensor([[10.0, 20.0, 120.0, 150.0]]))
        assert per[1][0].shape == boxes.shape and float((per[1][0] - boxes).abs().max()) < 1e-3
        assert float((per[1][1] - feats).abs().max()) <= 1e-4 * float(feats.abs().max())
