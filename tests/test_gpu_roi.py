"""RoIAlign forward and NMS on the MI355X against the plain-loop oracle (oracle/roi_ref.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def be():
    import odx
    odx.set_backend(None)
    return odx.get_backend()


@pytest.mark.parametrize("C,H,W,PH,PW,sr", [(20, 19, 25, 14, 14, 0), (5, 38, 50, 7, 7, 2), (33, 10, 12, 14, 14, 0)])
def test_roi_align_matches_oracle(be, C, H, W, PH, PW, sr):
    from oracle import roi_ref
    rng = np.random.default_rng(C + H)
    feat = rng.standard_normal((2, C, H, W)).astype(np.float32)
    R = 9
    xy = rng.random((R, 2)) * np.array([W * 16 * 0.7, H * 16 * 0.7])
    wh = rng.random((R, 2)) * np.array([W * 16 * 0.5, H * 16 * 0.5]) + 1
    rois = np.concatenate([rng.integers(0, 2, (R, 1)), xy, xy + wh], axis=1).astype(np.float32)
    rois[0, 1:] = [0, 0, 5, 5]                                # tiny box: size clamps to 1 cell
    rois[1, 1:] = [W * 16 - 30, H * 16 - 30, W * 16 + 40, H * 16 + 40]   # sticks out of the map
    rois[2, 1:] = [-50, -40, 60, 70]                          # negative corner
    got = be.roi_align(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16, (PH, PW), sr).cpu().numpy()
    ref = roi_ref.roi_align(feat, rois, 1.0 / 16, (PH, PW), sr)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("R,thr", [(1, 0.5), (63, 0.3), (64, 0.7), (300, 0.3), (1000, 0.7), (2500, 0.5)])
def test_nms_matches_oracle(be, R, thr):
    from oracle import roi_ref
    rng = np.random.default_rng(R)
    ctr = rng.random((R, 2)) * 400
    wh = rng.random((R, 2)) * 120 + 4
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    boxes[R // 2] = boxes[0]                                  # an exact duplicate
    scores = rng.random(R).astype(np.float32)
    scores[R // 3] = scores[0]                                # a score tie (stable order)
    got = be.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).cpu().numpy()
    ref = roi_ref.nms(boxes, scores, thr)
    assert np.array_equal(got, ref)
    assert tuple(be.nms(torch.zeros(0, 4), torch.zeros(0), thr).shape) == (0,)


def test_nms_first_k_equals_the_full_suppression_truncated():
    """be.nms(max_keep=k) (odx_nms_first_f32: the suppression stops at the k-th survivor) = be.nms()[:k], also when the
    caller hands the boxes over already sorted (sorted_desc=True, the RPN's top-k output) — sizes around the 64-box
    words of the mask, k below / at / above the number of survivors."""
    import odx
    be = odx.get_backend()
    g = torch.Generator().manual_seed(8)
    for R in (1, 63, 64, 65, 700, 6000):
        xy = torch.rand((R, 2), generator=g) * torch.tensor([700.0, 500.0])
        wh = 20 + torch.rand((R, 2), generator=g) * 160
        boxes = torch.cat((xy, xy + wh), dim=1).cuda()
        scores = torch.rand(R, generator=g).cuda()
        full = be.nms(boxes, scores, 0.7)
        order = torch.argsort(scores, descending=True, stable=True)
        for k in (1, 5, 300, int(full.numel()), int(full.numel()) + 7):
            assert torch.equal(be.nms(boxes, scores, 0.7, max_keep=k), full[:k]), (R, k)
            got = be.nms(boxes[order], scores[order], 0.7, max_keep=k, sorted_desc=True)
            assert torch.equal(order[got], full[:k]), (R, k)
        assert torch.equal(order[be.nms(boxes[order], scores[order], 0.7, sorted_desc=True)], full)


def test_roi_align_rows_is_the_strided_subset_of_the_grid():
    """odx_roi_align_rows_f32 (the bins a stride-2 1 x 1 convolution reads, as NHWC rows) = the full RoIAlign grid
    subsampled and permuted, bit for bit; odd grid sizes and steps 1 / 2 / 3."""
    import odx
    be = odx.get_backend()
    g = torch.Generator().manual_seed(2)
    feat = torch.randn((2, 37, 20, 27), generator=g).cuda()
    R = 19
    xy = torch.rand((R, 2), generator=g) * torch.tensor([300.0, 200.0])
    wh = 10 + torch.rand((R, 2), generator=g) * 150
    rois = torch.cat((torch.randint(0, 2, (R, 1), generator=g).float(), xy, xy + wh), dim=1).cuda()
    for (PH, PW), step in (((14, 14), 2), ((7, 9), 2), ((14, 14), 1), ((13, 14), 3)):
        full = be.roi_align(feat, rois, 1.0 / 16, (PH, PW), 0)
        rows, (r, OH, OW) = be.roi_align_rows(feat, rois, 1.0 / 16, (PH, PW), 0, step=step)
        ref = full[:, :, ::step, ::step]
        assert (r, OH, OW) == (R, ref.shape[2], ref.shape[3])
        assert torch.equal(rows.view(R, OH, OW, -1).permute(0, 3, 1, 2), ref)


def test_roi_align_rows_from_an_nhwc_map_equals_the_nchw_kernel():
    """odx_roi_align_rows_nhwc_f32 (the map handed over as the NHWC row matrix the trunk's GEMMs write — here a channels-last
    view, whole and sliced by image) = odx_roi_align_rows_f32 on the NCHW copy of the same map, bit for bit; channel counts
    below / above one pass of the workgroup (1024), steps 1 / 2 / 3, an odd grid."""
    import odx
    be = odx.get_backend()
    g = torch.Generator().manual_seed(4)
    for C in (16, 1024, 1500):
        nchw = torch.randn((3, C, 20, 27), generator=g).cuda()
        rows = nchw.permute(0, 2, 3, 1).contiguous()                          # (N, H, W, C): the row matrix
        nhwc = rows.permute(0, 3, 1, 2)                                       # the map as callers see it: a channels-last view
        assert not nhwc.is_contiguous() and nhwc.is_contiguous(memory_format=torch.channels_last)
        R = 23
        xy = torch.rand((R, 2), generator=g) * torch.tensor([300.0, 200.0])
        wh = 10 + torch.rand((R, 2), generator=g) * 150
        rois = torch.cat((torch.randint(0, 3, (R, 1), generator=g).float(), xy, xy + wh), dim=1).cuda()
        rois[0, 1:] = torch.tensor([-40.0, -30.0, 700.0, 500.0])               # samples outside the map on every side
        for (PH, PW), step in (((14, 14), 2), ((7, 9), 2), ((14, 14), 1), ((13, 14), 3)):
            want, shape = be.roi_align_rows(nchw, rois, 1.0 / 16, (PH, PW), 0, step=step)
            got, shape2 = be.roi_align_rows(nhwc, rois, 1.0 / 16, (PH, PW), 0, step=step)
            assert shape == shape2 and torch.equal(got, want), (C, PH, PW, step)
        one = rois[rois[:, 0] == 1].clone()
        one[:, 0] = 0
        want, _ = be.roi_align_rows(nchw[1:2], one, 1.0 / 16, (14, 14), 0, step=2)
        got, _ = be.roi_align_rows(nhwc[1:2], one, 1.0 / 16, (14, 14), 0, step=2)         # one image of the batch: still a view
        assert torch.equal(got, want)


def test_roi_align_fpn_from_nhwc_levels_equals_the_nchw_kernel():
    """odx_roi_align_fpn_nhwc_f32 (the pyramid's levels as channels-last views of the row GEMMs' outputs; crops flattened in
    (ph, pw, c) order) = odx_roi_align_fpn_f32 on NCHW copies of the same levels, permuted, bit for bit: RoIs over all four
    levels, both images of a batch, boxes hanging over the border, channel counts on every thread-count branch."""
    import odx
    be = odx.get_backend()
    g = torch.Generator().manual_seed(9)
    for C in (8, 64, 256, 300):
        sizes = [(56, 72), (28, 36), (14, 18), (7, 9)]
        nchw = [torch.randn((2, C, h, w), generator=g).cuda() for h, w in sizes]
        views = [f.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for f in nchw]
        R = 61
        xy = torch.rand((R, 2), generator=g) * torch.tensor([220.0, 160.0])
        wh = 4 + torch.rand((R, 2), generator=g) ** 2 * 900                    # sides from 4 to 904: every level of LevelMapper
        rois = torch.cat((torch.randint(0, 2, (R, 1), generator=g).float(), xy, xy + wh), dim=1).cuda()
        rois[0, 1:] = torch.tensor([-30.0, -20.0, 400.0, 300.0])
        scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
        for (PH, PW), sr in (((7, 7), 2), ((14, 14), 2), ((7, 5), 0)):
            want, lv = be.roi_align_fpn(nchw, rois, scales, (PH, PW), sr, return_levels=True)
            got = be.roi_align_fpn_rows(views, rois, scales, (PH, PW), sr)
            assert got.shape == (R, PH * PW * C)
            assert torch.equal(got.view(R, PH, PW, C).permute(0, 3, 1, 2), want), (C, PH, PW, sr)
            got2 = be.roi_align_fpn_rows(nchw, rois, scales, (PH, PW), sr)     # NCHW levels: through a channels-last copy
            assert torch.equal(got2, got)
        assert len(set(lv.tolist())) == 4
    assert be.roi_align_fpn_rows(views, rois[:0], scales, (7, 7), 2).shape == (0, 49 * C)
    # a pyramid held in 16 bits (odx_roi_align_fpn_nhwc_16): the samples decode exactly, the rest is the f32 kernel's arithmetic
    for dt in (torch.bfloat16, torch.float16):
        v16 = [f.to(dt) for f in views]
        assert all(not f.is_contiguous() and f.is_contiguous(memory_format=torch.channels_last) for f in v16)
        got16 = be.roi_align_fpn_rows(v16, rois, scales, (7, 7), 2)
        want16 = be.roi_align_fpn_rows([f.float() for f in v16], rois, scales, (7, 7), 2)
        assert got16.dtype == torch.float32 and torch.equal(got16, want16), dt


@pytest.mark.parametrize("N,C,H,W", [(1, 64, 150, 200), (2, 7, 19, 25), (1, 3, 1, 1), (1, 256, 38, 50),
                                     (300, 512, 7, 7), (70001, 1, 2, 3), (64, 1024, 4, 4)])     # > 65535 planes (advisor, round 4)
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_trunk_epilogue_equals_the_three_torch_ops_bit_for_bit(be, N, C, H, W, with_res, relu, dtype):
    """odx_bias_act_nchw_f32 / _16 = (y + bias) (+ residual), ReLU: one addition per term in the reference's order, rounded
    to the map's type each time, so the result is the separate torch ops' bit for bit — planes whose size is and is not a
    multiple of the 16-byte vector (vector / scalar path), NaN passed through as torch.relu does."""
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + C + H)
    dt = getattr(torch, dtype)
    y = torch.randn((N, C, H, W), generator=g, device="cuda").to(dt)
    bias = torch.randn(C, generator=g, device="cuda").to(dt)
    res = torch.randn((N, C, H, W), generator=g, device="cuda").to(dt) if with_res else None
    y.view(-1)[0] = float("nan")
    ref = y + bias.view(1, -1, 1, 1)
    if with_res:
        ref = ref + res
    if relu:
        ref = torch.relu(ref)
    got = be.bias_act_(y.clone(), bias, res, relu=relu)
    assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(ref, nan=123.0))
    assert bool(torch.isnan(got.view(-1)[0])) == bool(torch.isnan(ref.view(-1)[0]))


@pytest.mark.parametrize("B,A,H,W,k", [(3, 15, 20, 26, 700), (1, 15, 38, 50, 6000), (2, 3, 5, 7, 105), (4, 15, 12, 16, 1), (2, 15, 24, 32, 8192)])
def test_rpn_topk_decode_equals_the_tensor_op_sequence(be, B, A, H, W, k):
    """odx_rpn_topk_decode_f32 — objectness top-k, sorting, delta gather, decoding, clipping in one launch — against the
    sequence of tensor operations it replaces (extract.rpn_proposals: permute, top-k, gather, decode_deltas, clamp): the same
    candidates in the same order (ties — duplicated logits are planted — by the lower flat index, what a stable sort gives),
    boxes to 1e-4 px, scores = sigmoid(logit)."""
    from odx.extract import DELTA_CLAMP, decode_deltas, grid_anchors, cell_anchors
    g = torch.Generator().manual_seed(B * 1000 + k)
    logits = torch.randn((B, A, H, W), generator=g) * 3
    lv = logits.view(-1)
    dup = lv[3::7].clone()
    lv[0:7 * len(dup):7] = dup                                                          # ties
    logits.view(-1)[5] = 0.0
    logits.view(-1)[6] = -0.0
    deltas = torch.randn((B, 4 * A, H, W), generator=g) * 0.3
    deltas[:, 2::4] *= 10                                                              # some dw past the clamp
    cells = cell_anchors(16)[:A] if A <= 15 else None
    anchors = grid_anchors(H, W, 16, cells)
    img = (W * 16.0, H * 16.0)
    boxes, scores, index = be.rpn_topk_decode(logits.cuda(), deltas.cuda(), anchors.cuda(), k, img, DELTA_CLAMP)
    obj = logits.permute(0, 2, 3, 1).reshape(B, -1)
    reg = deltas.view(B, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(B, -1, 4)
    for b in range(B):
        order = torch.from_numpy(np.argsort(-obj[b].numpy().astype(np.float64), kind="stable"))[:k]
        assert torch.equal(index[b].cpu().long(), order), b
        want = decode_deltas(reg[b][order], anchors[order])
        want[:, 0::2] = want[:, 0::2].clamp(0, img[0] - 1)
        want[:, 1::2] = want[:, 1::2].clamp(0, img[1] - 1)
        assert float((boxes[b].cpu() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()) / 1e3)
        assert float((scores[b].cpu() - obj[b][order].sigmoid()).abs().max()) < 1e-6
        assert bool((scores[b][:-1] >= scores[b][1:]).all())


@pytest.mark.parametrize("B,R,P", [(3, 500, 40), (1, 6000, 300), (4, 64, 64), (2, 130, 1000)])
def test_nms_compact_packs_the_first_survivors(be, B, R, P):
    g = torch.Generator().manual_seed(R + P)
    boxes = torch.rand((B, R, 4), generator=g).cuda() * 100
    keep = (torch.rand((B, R), generator=g) < 0.3).cuda()
    keep[0] = False if B > 1 else keep[0]
    counts = torch.tensor([R - 7 * b for b in range(B)], dtype=torch.int32).cuda()
    out, n = be.nms_compact(boxes, keep, P, counts=counts)
    assert tuple(out.shape) == (B, P, 4) and n.dtype == torch.int32
    for b in range(B):
        kb = keep[b].clone()
        kb[int(counts[b]):] = False
        want = boxes[b][kb][:P]
        assert int(n[b]) == len(want)
        assert torch.equal(out[b, :len(want)], want)
        assert bool((out[b, len(want):] == torch.tensor([0.0, 0.0, 15.0, 15.0], device="cuda")).all())
    out2, n2 = be.nms_compact(boxes, keep.to(torch.uint8), P)                       # no counts: every set holds R boxes
    assert [int(v) for v in n2] == [min(P, int(keep[b].sum())) for b in range(B)]


def test_stem_pool_rows_equals_epilogue_pooling_and_transposition():
    """odx_stem_pool_rows_f32 / _16 (bias + ReLU + 3 x 3 / 2 / 1 max pooling + NCHW -> NHWC rows + the maximum, one pass) = the
    in-place epilogue (bias_act_), the library's max_pool2d and the permuting copy it replaces, bit for bit: even and odd map
    sizes (a last window hanging over the border), channel counts below / at / above one 64-channel pass, f32 and both 16-bit
    types (whose sums are rounded once, as odx_bias_act_nchw_16 does)."""
    import torch.nn.functional as F
    import odx
    be = odx.get_backend()
    g = torch.Generator().manual_seed(12)
    for (B, C, H, W) in ((2, 64, 30, 40), (1, 16, 7, 9), (3, 72, 33, 130), (1, 64, 300, 400)):
        x = torch.randn((B, C, H, W), generator=g).cuda()
        bias = torch.randn(C, generator=g).cuda()
        want = F.max_pool2d(be.bias_act_(x.clone(), bias, None, relu=True), 3, 2, 1)
        rows, (b2, Ho, Wo), meta = be.stem_pool_rows(x, bias)
        assert (b2, Ho, Wo) == (B, want.shape[2], want.shape[3]) and rows.shape == (B * Ho * Wo, C)
        assert torch.equal(rows.view(B, Ho, Wo, C).permute(0, 3, 1, 2), want)
        assert meta.view(torch.int32)[1].item() == want.abs().max().view(torch.int32).item()
        for dt in (torch.bfloat16, torch.float16):
            x16, b16 = x.to(dt), bias.to(dt)
            want16 = F.max_pool2d(be.bias_act_(x16.clone(), b16, None, relu=True), 3, 2, 1)
            r16, dims, none = be.stem_pool_rows(x16, b16)
            assert none is None and dims == (B, Ho, Wo) and r16.zero_row and r16.K == C
            assert torch.equal(r16.dense.reshape(B, Ho, Wo, C).permute(0, 3, 1, 2), want16), dt
            assert float(r16.buf[:, C:].abs().max()) == 0.0 if r16.buf.shape[1] > C else True
