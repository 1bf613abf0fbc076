"""Driver-visible numbers for the halves of BASELINE configs 2 and 3 that the headline metric does not time
(bench.py adds them to its JSON line as extra keys, measured OUTSIDE the timed headline region, rank 0, N = 1 only):

  rls            config 3's second half: the 30 per-class RLS box regressors on COXY n = 3e5, D = 1024, lambda = 1000
                 (train_region_refiner.py:25-119) through the drop-in trainer, with the numpy-f64 oracle timed beside it
                 on one class (a bounded sample) on the host cores
  forward        config 2's first half: OnlineDetectionModel.forward on a synthetic 600 x 800 image, 300 RoIs,
                 f32 and bf16 autocast, random weights (feature_proposal_extractor.py:228-281)
  minibootstrap  the reference regime (OnlineRegionClassifier_incore.py:96-155): 30 classes x 10 negative batches of
                 2000 rows, M = 2000, D = 2048, in the reference's sequential order and in the opt-in class-parallel modes (streams; batched preconditioners)
"""
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64_MFMA_PEAK_TFLOPS = 78.6    # /opt/skills/guides/MI355X_MICROARCH.md: FP64 matrix


def _sync_time(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def _best_time(fn, batches=3):
    """Shortest of `batches` timed calls: the steady state (one slow batch — an allocator refill after the cache was emptied
    between extras, a clock ramp — is not what these keys report)."""
    best, out = None, None
    for _ in range(batches):
        dt, out = _sync_time(fn)
        best = dt if best is None else min(best, dt)
    return best, out


def rls_extra(n=300_000, D=1024, C=30, lam=1000.0, cpu=True):
    from odx.rls import RegionRefinerTrainer
    g = torch.Generator(device="cuda").manual_seed(1234 + 3)
    X = torch.randn((n, D), generator=g, device="cuda") * 0.6 + 0.15
    cls = (torch.arange(n, device="cuda") % C) + 1
    Y = torch.randn((n, 4), generator=g, device="cuda") * 0.2
    cfg = {"CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}, "REGION_REFINER": {"opts": {"lambda": lam}}}
    COXY = {"C": cls.float().view(-1, 1), "O": None, "X": X, "Y": Y}
    best = None
    for _ in range(8):                                  # the first repetitions warm kernels, allocations and clocks (three were not
        # enough: the best of a first call's three is 1 ms above the best of the next call's)
        with redirect_stdout(io.StringIO()):
            dt, models = _sync_time(lambda: RegionRefinerTrainer(cfg, lam, False)(COXY))
        best = dt if best is None else min(best, dt)
    nc = n // C
    flop = C * (nc * (D + 1.0) ** 2 + (D + 1.0) ** 3 / 3)           # lower-triangle Gram + Cholesky, per class
    out = {"workload": "%d RLS box regressors, COXY n=%d D=%d lambda=%g (%d rows per class), f64" % (C, n, D, lam, nc),
           "ms": round(best * 1e3, 2), "regressors_per_s": round(C / best, 1),
           "achieved_TFLOPs_f64": round(flop / best / 1e12, 2), "peak_TFLOPs_f64_mfma": F64_MFMA_PEAK_TFLOPS,
           # (a register-only loop of the instruction the Grams run on, tools/micro/mfma_f64_rate.hip, round 5: the part sustains this)
           "sustained_TFLOPs_v_mfma_f64_16x16x4_measured": 67.0}
    if cpu:
        from oracle import rls_ref
        I = torch.where(cls == 1)[0]
        Xh, Yh = X[I].cpu().numpy(), Y[I].cpu().numpy()
        dtc = None
        for _ in range(3):                              # a cold first call (BLAS threads, page faults) read 6 x slower once
            t0 = time.perf_counter()
            ref = rls_ref.train_class(Xh, Yh, lam)
            dtc = min(time.perf_counter() - t0, dtc or 1e9)
        W = torch.stack([models[0]["Beta"][str(k)]["weights"] for k in range(4)]).cpu().numpy()
        out["cpu_baseline"] = {"value": round(1.0 / dtc, 2), "unit": "regressors/s", "kind": "port", "cores": int(torch.get_num_threads()),
                               "sample": "oracle/rls_ref.py (numpy/scipy f64) on 1 class of %d rows in %.2f s (best of 3)" % (len(I), dtc)}
        out["max_abs_weight_diff_vs_oracle_class1"] = float(np.abs(W - ref["W"]).max())
    return out


def forward_extra(height=600, width=800, rois=300, reps=8, dtypes=("f32", "bf16"), groups=(4, 8)):
    from odx.extract import OnlineDetectionModel
    dev = torch.device("cuda")
    model = OnlineDetectionModel(post_nms_top_n=rois).to(dev).eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randn((1, 3, height, width), device=dev, generator=g)
    out = {"workload": "R-50-C4 trunk + RPN proposals (HIP NMS) + RoIAlign (HIP) + conv5 head on one synthetic %dx%d image, "
                       "%d RoIs, random weights" % (height, width, rois)}
    for name, ctx in (("f32", torch.autocast("cuda", enabled=False)), ("bf16", torch.autocast("cuda", dtype=torch.bfloat16))):
        if name not in dtypes:
            continue
        with torch.no_grad(), ctx:
            for _ in range(3):
                model(img)
            dt, _ = _best_time(lambda: [model(img) for _ in range(reps)])
        out["images_per_s_" + name] = round(reps / dt, 1)
        out["ms_per_image_" + name] = round(dt / reps * 1e3, 2)
        out["roofline_" + name] = _forward_roofline(forward_flop_c4(model, height, width, rois), dt / reps * 1e3, name)
    # groups of same-size images through ONE forward each (extract.forward_batch: one trunk call, one proposal stage, one pass
    # of the RoI head per group; in f32 from three images on the trunk stages, the RPN head and the conv5 head are one chain of
    # row GEMMs on the split-f16 tile cores) — what the harvest loop runs; the one-image keys above are detect()'s route
    for name, cd in (("f32", None), ("bf16", torch.bfloat16)):
        if name not in dtypes or not groups:
            continue
        gm = OnlineDetectionModel(post_nms_top_n=rois, compute_dtype=cd).to(dev).eval()
        from odx.extract import forward_batch
        for B in groups:
            imgs = torch.randn((B, 3, height, width), device=dev, generator=g)
            with torch.no_grad():
                for _ in range(3):
                    forward_batch(gm, imgs)
                dt, _ = _best_time(lambda: [forward_batch(gm, imgs) for _ in range(4)])
            ms = dt / (4 * B) * 1e3
            out["ms_per_image_%s_group%d" % (name, B)] = round(ms, 2)
            if B == 8:
                out["roofline_%s_group8" % name] = _forward_roofline(forward_flop_c4(gm, height, width, rois), ms, name)
        del gm
    return out


def _conv_flop(conv, H, W):
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    return 2.0 * conv.in_channels * conv.out_channels * kh * kw * Ho * Wo / conv.groups, Ho, Wo


def _stage_flop(stage, H, W):
    """Multiply-add flop (x 2) of a ResNet stage of extract.Bottleneck blocks on an H x W input map; returns (flop, Ho, Wo)."""
    tot = 0.0
    for blk in stage:
        f1, H1, W1 = _conv_flop(blk.conv1, H, W)
        f2, H2, W2 = _conv_flop(blk.conv2, H1, W1)
        f3, H3, W3 = _conv_flop(blk.conv3, H2, W2)
        tot += f1 + f2 + f3
        if blk.down is not None:
            tot += _conv_flop(blk.down[0], H, W)[0]
        H, W = H3, W3
    return tot, H, W


def forward_flop_c4(model, H, W, rois):
    """Algorithmic flop per image of OnlineDetectionModel.forward (SURVEY A11's count): trunk (stem + res2..res4), RPN head
    (3 x 3 convolution + the two 1 x 1), conv5 head on `rois` crops (the 7 x 7 positions its stride-2 entry keeps)."""
    bb = model.backbone
    f, h, w = _conv_flop(bb.conv1, H, W)
    h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1               # max_pool2d(3, 2, 1)
    trunk = f
    for st in (bb.layer1, bb.layer2, bb.layer3):
        f, h, w = _stage_flop(st, h, w)
        trunk += f
    rpn = _conv_flop(model.rpn_conv, h, w)[0] + _conv_flop(model.rpn_logits, h, w)[0] + _conv_flop(model.rpn_deltas, h, w)[0]
    head = rois * _stage_flop(model.head.layer4, model.resolution, model.resolution)[0]
    return {"trunk": trunk, "rpn_head": rpn, "roi_head": head, "total": trunk + rpn + head}


def forward_flop_fpn(model, H, W, rois):
    bb = model.backbone
    f, h, w = _conv_flop(bb.conv1, H, W)
    h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    trunk, sizes = f, []
    for st in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        f, h, w = _stage_flop(st, h, w)
        trunk += f
        sizes.append((h, w))
    for k, (hh, ww) in enumerate(sizes):
        trunk += _conv_flop(model.fpn.inner[k], hh, ww)[0] + _conv_flop(model.fpn.layer[k], hh, ww)[0]
    sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    rpn = sum(_conv_flop(model.rpn_conv, hh, ww)[0] + _conv_flop(model.rpn_logits, hh, ww)[0] + _conv_flop(model.rpn_deltas, hh, ww)[0]
              for hh, ww in sizes)
    head = 2.0 * rois * (model.fc6.in_features * model.fc6.out_features + model.fc7.in_features * model.fc7.out_features)
    return {"trunk": trunk, "rpn_head": rpn, "roi_head": head, "total": trunk + rpn + head}


def _forward_roofline(flop, ms, name):
    """The forward against the matrix-core ceiling of its dtype: f32 = the 3-MFMA split form's ceiling (a third of the dense f16
    peak: what the hand-written layers can reach at f32 accuracy; the library's f32 convolutions of the trunk have the 157 TF
    f32-MFMA peak, which would flatter the figure), bf16 = the dense bf16 peak."""
    peak = F16_MFMA_PEAK_TFLOPS / 3.0 if name == "f32" else F16_MFMA_PEAK_TFLOPS
    tf = flop["total"] / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "gflop_per_image": {k: round(v / 1e9, 1) for k, v in flop.items()},
            "ceiling": "dense f16 MFMA peak / 3 (three MFMAs per f32-accurate product)" if name == "f32" else "dense bf16 MFMA peak"}


def forward_fpn_extra(height=600, width=800, reps=8, dtypes=("f32", "bf16"), groups=(4, 8)):
    """BASELINE config 2's forward as named — ResNet50-FPN + RoIAlign -> D = 1024 RoI features (odx/fpn.py): trunk + pyramid,
    RPN over five levels (1000 candidates per level, NMS, 1000 kept over all levels), ONE multi-level RoIAlign launch, fc6 /
    fc7 on the split-f16 tile cores; f32 and bf16 autocast, random weights."""
    from odx.fpn import OnlineDetectionModelFPN
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randn((1, 3, height, width), device=dev, generator=g)
    out = {"workload": "R-50-FPN trunk + pyramid + RPN over 5 levels (HIP NMS) + multi-level RoIAlign (one HIP launch) + fc6 / fc7 on "
                       "one synthetic %dx%d image, random weights" % (height, width)}
    for name, dt in (("f32", None), ("bf16", torch.bfloat16)):
        if name not in dtypes:
            continue
        model = OnlineDetectionModelFPN(compute_dtype=dt).to(dev).eval()
        with torch.no_grad():
            for _ in range(3):
                boxes, feats, _ = model(img)
            dt_s, _ = _best_time(lambda: [model(img) for _ in range(reps)])
        out["images_per_s_" + name] = round(reps / dt_s, 1)
        out["ms_per_image_" + name] = round(dt_s / reps * 1e3, 2)
        out["rois"], out["feature_dim"] = int(boxes.shape[0]), int(feats.shape[1])
        out["roofline_" + name] = _forward_roofline(forward_flop_fpn(model, height, width, int(boxes.shape[0])), dt_s / reps * 1e3, name)
        # groups of images of one size through extract.forward_batch: one trunk + pyramid call (f32, three images or more: the
        # stages and the pyramid as row GEMMs on the split-f16 tile cores), the proposal stage per image, ONE fc6 / fc7 pass
        from odx.extract import forward_batch
        for B in groups:
            imgs = torch.randn((B, 3, height, width), device=dev, generator=g)
            with torch.no_grad():
                for _ in range(3):
                    forward_batch(model, imgs)
                dt_g, _ = _best_time(lambda: [forward_batch(model, imgs) for _ in range(3)])
            ms = dt_g / (3 * B) * 1e3
            out["ms_per_image_%s_group%d" % (name, B)] = round(ms, 2)
            if B == 8:
                out["roofline_%s_group8" % name] = _forward_roofline(forward_flop_fpn(model, height, width, int(boxes.shape[0])), ms, name)
        del model
    out["note"] = ("trunk + pyramid replayed from a HIP graph per image size (extract.GraphedCall): launch by launch this forward was "
                   "host-bound at batch 1 and a 16-bit trunk, whose kernels are shorter, gained nothing (5.1-6.4 ms against 5.0-5.3 in "
                   "f32); replayed, the 16-bit trunk's device time is what counts")
    return out


def harvest_extra(images=64, C=30, height=600, width=800, passes=6, one_image_per_call=True):
    """The on-line training's feature pass per image (FeatureExtractorRPNDetector.train, extract_features_rpn_detector.py:105-369 —
    by the builder's own figures 99 % of the reference's reported on-line training time): forward + detector rows + on-line RPN
    rows + mask pixel rows through OnlineFeatureExtractor on synthetic images of one size with 1-3 ground-truth boxes each,
    random weights, the images already on the device.  ms per image, best of `passes` passes over the list.  64 images = 8
    groups of 8: the loop is a two-stage pipeline (group k + 1's forward on the GPU under group k's harvest on the host) whose
    first forward overlaps with nothing — at 24 images (round 5's harness) that fill was a fifth of a pass, at the reference's
    8000 it is nothing."""
    from odx.extract import OnlineDetectionModel, OnlineFeatureExtractor
    dev = torch.device("cuda")
    model = OnlineDetectionModel().to(dev).eval()
    g = torch.Generator().manual_seed(3)
    samples = []
    for i in range(images):
        img = torch.randn((1, 3, height, width), generator=g)
        G = 1 + i % 3
        xy = torch.rand((G, 2), generator=g) * torch.tensor([width - 300.0, height - 300.0])
        wh = 80 + torch.rand((G, 2), generator=g) * 200
        boxes = torch.cat((xy, xy + wh), dim=1)
        labels = [1 + (i + j) % C for j in range(G)]
        masks = torch.zeros((G, height, width), dtype=torch.uint8)
        for j in range(G):
            x1, y1, x2, y2 = [int(v) for v in boxes[j]]
            masks[j, y1 + 10:y2 - 10, x1 + 10:x2 - 10] = 1
        samples.append((img.to(dev), boxes.to(dev), labels, masks.to(dev)))
    out = {"workload": "forward + detector / on-line RPN / mask-pixel harvest of %d synthetic %dx%d images (1-3 boxes each, %d classes), "
                       "R-50-C4, 300 proposals, f32, random weights" % (images, height, width, C)}
    # (ms_per_image: the images ARE the stream — ceil(20 000 / images) negatives per class and image;
    # ms_per_image_stream_of_8000: the same images as a part of a stream of 8000, the reference's scale (iCWT: 7976 training
    # images) — 3 negatives per class and image, the quota the reference's own runs harvest with)
    for key, kw in (("ms_per_image", {}), ("ms_per_image_stream_of_8000", {"num_images": 8000}),
                    ("ms_per_image_one_image_per_call", {"trunk_batch": 1})):
        if kw and not one_image_per_call:
            continue
        ex = OnlineFeatureExtractor(model, C, parts=("rpn", "detector", "mask"), **kw)
        torch.manual_seed(0)
        ex.train(samples[:6])
        best = None
        for _ in range(passes):               # (a host-bound loop: another tenant's burst on the box's cores reads as + 10 % in one pass)
            torch.manual_seed(0)
            dt, _ = _sync_time(lambda: ex.train(samples))
            best = dt if best is None else min(best, dt)
        out[key] = round(best / images * 1e3, 2)
        if not kw:
            out["images_per_call"] = int(ex.trunk_batch)
    return out


def detect_extra(height=600, width=800, rois=300, C=30, M=1000, reps=10, groups=(4, 8)):
    """One test-time image end to end (odx.extract.detect: trunk, RPN proposals, RoIAlign, conv5 head, the on-line box
    head of C FALKON classifiers + C box regressors, decode / threshold / per-class NMS / top-k): ms per image, with the
    post-processing also timed on its own.  Random weights and random models (shape only)."""
    import odx
    from odx.extract import OnlineDetectionModel, detect
    from odx.heads import OnlineBoxPredictor
    from odx.postprocess import postprocess_detections
    dev = torch.device("cuda")
    model = OnlineDetectionModel(post_nms_top_n=rois).to(dev).eval()
    D = model.feat_dim
    g = torch.Generator().manual_seed(4)
    clfs = []
    for c in range(C):
        m = odx.InCoreFalkon(kernel=odx.GaussianKernel(20.0), penalty=1e-4, M=M)
        m.ny_points_ = (torch.randn(M, D, generator=g) * (20.0 / D ** 0.5)).to(dev)
        m.alpha_ = (torch.randn(M, 1, generator=g, dtype=torch.float64) * 0.05).to(dev)
        clfs.append(m)
    regs = [{"mu": torch.zeros(4), "T": torch.eye(4), "T_inv": torch.eye(4),
             "Beta": {str(k): {"weights": torch.randn(D + 1, generator=g) * 0.01} for k in range(4)}} for _ in range(C)]
    stats = {"mean": torch.zeros(D, device=dev), "std": torch.ones(D, device=dev), "mean_norm": torch.tensor(20.0, device=dev)}
    model.online_box = OnlineBoxPredictor(clfs, regs, stats)
    img = torch.randn((1, 3, height, width), generator=g).to(dev)
    with torch.no_grad():
        for _ in range(3):
            res, boxes = detect(model, img, (width, height), -2.0, 0.3, 100)
        dt, _ = _best_time(lambda: [detect(model, img, (width, height), -2.0, 0.3, 100) for _ in range(reps)])
        maps = model.roi_head_maps(model.c4(img), boxes)
        scores, deltas = model.online_box(maps.mean(dim=(2, 3)))
        dtp, _ = _best_time(lambda: [postprocess_detections(scores, deltas, boxes, (width, height), -2.0, 0.3, 100) for _ in range(reps)])
        # groups of images through one forward and one pass of the heads (extract.detect_batch: what the evaluator drop-in runs)
        from odx.extract import detect_batch
        grp = {}
        for B in groups:
            imgs = torch.randn((B, 3, height, width), generator=g).to(dev)
            for _ in range(3):
                detect_batch(model, imgs, [(width, height)] * B, -2.0, 0.3, 100)
            dtg, _ = _best_time(lambda: [detect_batch(model, imgs, [(width, height)] * B, -2.0, 0.3, 100) for _ in range(3)])
            grp["ms_per_image_group%d" % B] = round(dtg / (3 * B) * 1e3, 2)
    out = {"workload": "detect(): %dx%d image, %d proposals, %d FALKON classifiers (M=%d, D=%d) + %d box regressors, decode + "
                       "per-class NMS + top-100, f32" % (height, width, boxes.shape[0], C, M, D, C),
           "ms_per_image": round(dt / reps * 1e3, 2), "images_per_s": round(reps / dt, 1),
           "postprocessing_ms": round(dtp / reps * 1e3, 2), "detections": 0 if res is None else int(len(res["scores"]))}
    out.update(grp)
    return out


def minibootstrap_extra(C=30, D=2048, IT=10, positives=800, sigma=15.0, modes=(("default", None), ("class_by_class_loop", {"reference_order": "sequential"}), ("class_streams4", {"class_streams": 4}),
                               ("class_batch4", {"class_batch": 4}))):
    import yaml
    from tests import dropin
    names = ["_background_"] + ["c%d" % i for i in range(C)]
    cfg = {"NUM_CLASSES": C + 1,
           "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                        "CLASSIFIER": {"lambda": 0.0001, "sigma": sigma, "M": 2000, "kernel_type": "gauss"}},
           "CHOSEN_CLASSES": {i: c for i, c in enumerate(names)}}
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    g = torch.Generator(device="cuda").manual_seed(1)
    mu = torch.randn((C, D), device="cuda", generator=g)

    def data():
        pos = [mu[c] + 0.7 * torch.randn((positives, D), device="cuda", generator=g) for c in range(C)]
        neg = [[mu[(c + 1 + j) % C] * 0.5 + 0.8 * torch.randn((2000, D), device="cuda", generator=g) for j in range(IT)] for c in range(C)]
        return pos, neg

    u = dropin.load("py_od_utils")
    clf_mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    out = {"workload": "%d classes x %d negative batches of 2000 rows, %d positives, D=%d, M=2000 (1 fit + 2 predicts per class and "
                       "batch) through OnlineRegionClassifier_incore.trainRegionClassifier" % (C, IT, positives, D)}
    for name, opts in modes:
        best = None
        for rep in range(3 if name in ("default", "class_batch4") else 2):     # the first repetition warms every kernel and allocation
            pos, neg = data()
            with redirect_stdout(io.StringIO()):
                torch.manual_seed(7)
                stats = u.computeFeatStatistics_torch(pos, neg, features_dim=D, pos_fraction=0.8)
                orc = orc_mod.OnlineRegionClassifier(clf_mod.FALKONWrapper(cfg_path=path), pos, neg, stats, cfg_path=path)
                dt, models = _sync_time(lambda: orc.trainRegionClassifier(opts=dict(opts) if opts else None))
            best = dt if best is None else min(best, dt)
        out["s_" + name] = round(best, 3)
        out["trained_" + name] = sum(1 for m in models if m is not None)
        if name == "default":
            out["default_ran_as"] = getattr(orc, "last_order", "class by class")
    return out


F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide: dense f16 / bf16 MFMA
F8_MFMA_PEAK_TFLOPS = 5000.0    # same guide: dense fp8 MFMA (MX-scaled K = 128 forms)
HBM_PEAK_GBS = 8000.0


def _blob_rows(n, D, C, seed, device):
    """Rows of C class blobs (row i of class i % C), normalised with the reference rule (OnlineRegionClassifier.py:224-227)."""
    g = torch.Generator(device=device).manual_seed(seed)
    mu = torch.randn((C, D), generator=g, device=device)
    X = torch.empty((n, D), dtype=torch.float32, device=device)
    for b0 in range(0, n, 131072):
        b1 = min(n, b0 + 131072)
        X[b0:b1] = mu[torch.arange(b0, b1, device=device) % C] + 0.7 * torch.randn((b1 - b0, D), generator=g, device=device)
    X -= mu.mean(0)
    X *= 20.0 / torch.sqrt(((mu - mu.mean(0)) ** 2).sum(1) + 0.49 * D).mean()
    return X


def _centre_indices(y, M, rng):
    """The reference's Nystroem index rule (FALKONWrapper_with_centers_selection_incore.py:87-99) on host labels: all
    positives when there are at most M // 2, else M // 2 of them drawn with replacement; negatives fill up to M the same
    way; positives first."""
    pos, neg = np.flatnonzero(y == 1), np.flatnonzero(y == -1)
    half = M // 2
    p = pos[rng.integers(0, len(pos), half)] if len(pos) > half else pos
    room = M - len(p)
    q = neg[rng.integers(0, len(neg), room)] if len(neg) > room else neg
    return np.concatenate([p, q]).astype(np.int64)


def _events_ms(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def falkon_config_extra(name, C, n, D, M, sigma, lam, classes_run, labels="one_vs_rest", seed=1234, maxiter=20, gauss=None,
                        storage=None):
    """`gauss` / `storage`: run with the backend's contraction (h2 / f8) and K_nM storage (f32 / u24 / bf16) set to these
    for the duration of the call — the throughput-only variants BASELINE configs 2 and 5 name."""
    import odx
    be = odx.get_backend()
    old = (be.gauss, be.knm_storage)
    try:
        if gauss is not None:
            be.gauss = gauss
        if storage is not None:
            be.knm_storage = storage
        return _falkon_config_extra(name, C, n, D, M, sigma, lam, classes_run, labels, seed, maxiter)
    finally:
        be.gauss, be.knm_storage = old


def _falkon_config_extra(name, C, n, D, M, sigma, lam, classes_run, labels="one_vs_rest", seed=1234, maxiter=20):
    """One BASELINE config's FALKON leg on this GPU: `classes_run` of its C classes fitted (Nystroem centres by the
    reference rule, class-batched preconditioner chain, K_nM build with the fused right-hand side, the CG loops of the
    classes in lock step) and every row scored with each model.  Reports the config's metric (rows / seconds for ALL C
    classes, extrapolated from the classes run when fewer) and, per kernel family, the rate of ONE launch at this shape
    measured on its own against its roofline (F_K = 2 n M D flop, B_CG = n M s_K bytes per pass)."""
    import odx
    from odx.solver import SolverOptions
    be = odx.get_backend()
    dev = be.device
    rows_c = 2 if labels == "pixels" else C
    X = _blob_rows(n, D, rows_c, seed, dev)
    rng = np.random.default_rng(seed)
    ids = torch.arange(n, device=dev) % rows_c
    opt = SolverOptions(check_pivots=False)
    run = list(range(classes_run))

    def labels_of(c):
        if labels == "pixels":          # foreground / background pixels of class c's own boxes, 5 % label noise
            g = torch.Generator(device=dev).manual_seed(seed + 17 * c)
            y = torch.where(ids == 0, 1.0, -1.0)
            return torch.where(torch.rand(n, generator=g, device=dev) < 0.05, -y, y).to(torch.float64)
        return torch.where(ids == c, 1.0, -1.0).to(torch.float64)

    def job():
        F = be.features(X)
        scores = torch.empty((n, len(run)), dtype=torch.float32, device=dev)
        group = max(1, min(len(run), be.MAX_CLASS_BATCH, int(8e9 // (4 * M * M * 8))))     # factors of a chain: <= 8 GB
        infos = []
        for g0 in range(0, len(run), group):
            cls = run[g0:g0 + group]
            ys = [labels_of(c) for c in cls]
            idx = [_centre_indices(np.asarray(y.cpu()), M, rng) for y in ys]
            Zfs = [be.rows(F, torch.as_tensor(np.asarray(i))) for i in idx]
            Ps = be.precond_batched(Zfs, sigma, lam, opt.pc_epsilon, ws_key="extra_precond")
            infos.extend(p.info for p in Ps)
            Mp = (M + 1) // 2 * 2
            b0s = torch.zeros((len(cls), Mp), dtype=torch.float64, device=dev)
            alphas = None
            if be.cg_batched_supported([n] * len(cls), [M] * len(cls), be.knm_format(n, M)) and len(cls) * be.knm_bytes(n, M) <= 60e9:
                # all K_nM builds, then the CG loops of the chain's classes in lock step (f32 or compact blocks alike)
                Ks = [be.knm_rhs(F, Zfs[k], sigma, ys[k] * (1.0 / n), rhs_out=b0s[k, :Zfs[k].n])[0] for k in range(len(cls))]
                alphas = be.cg_solve_batched(Ks, Ps, b0s, [n] * len(cls), lam, maxiter, opt)
                del Ks
            if alphas is None:          # HBM-bound blocks (compact storage) or M outside the batched configurations: one fit per class
                alphas = torch.stack([torch.nn.functional.pad(odx.falkon_fit(be, F, ys[k], Zfs[k], sigma, lam, maxiter, opt, precond=Ps[k]),
                                                              (0, Mp - Zfs[k].n)) for k in range(len(cls))])
            for k, c in enumerate(cls):
                be.mmv(F, Zfs[k], sigma, alphas[k, :Zfs[k].n], None, out=scores[:, g0 + k:g0 + k + 1])
        return scores, infos, (F, Zfs[-1], alphas[-1, :Zfs[-1].n])

    job()                                                  # warms kernels, workspaces and the allocator
    dt, (scores, infos, (F, Zf, alpha)) = _sync_time(job)
    failed = int(sum(int(i.item() != 0) for i in infos))
    finite = bool(torch.isfinite(scores).all().item())
    s_all = dt * C / len(run)
    out = {"workload": "%s: %d classes x (FALKON fit + score-all) on n=%d rows, D=%d, M=%d, sigma=%g, lambda=%g, %d CG iterations; "
                       "%d class(es) run on this GPU%s" % (name, C, n, D, M, sigma, lam, maxiter, len(run),
                                                          "" if len(run) == C else ", the config's time extrapolated linearly in the class count"),
           "dtype": {"h2": "f32-accurate K_nM entries via two-term f16 split", "f8": "K_nM entries from e4m3 operands (throughput only)",
                     "f32": "f32 K_nM entries (f32 MFMA)"}[be.gauss] + ", stored as %s, f64 solver" % be.knm_format(n, M),
           "s_classes_run": round(dt, 4), "classes_run": len(run), "samples_per_s": round(n / s_all, 1),
           "samples_x_classes_per_s": round(n * C / s_all, 1), "failed_choleskys": failed, "scores_finite": finite}
    if labels != "pixels" and len(run) == C:
        out["argmax_accuracy_on_blobs"] = round(float((scores.argmax(1) == ids).float().mean().item()), 4)
    # ---- one launch of each kernel family at this shape, alone
    yv = labels_of(0) * (1.0 / n)
    Kst = {}

    def build():
        Kst["K"] = be.knm_rhs(F, Zf, sigma, yv)[0]
    ms_b = _events_ms(build, 3)
    K = Kst["K"]
    v = torch.ones(Zf.n, dtype=torch.float64, device=dev)
    o = torch.empty(Zf.n, dtype=torch.float64, device=dev)
    ms_p = _events_ms(lambda: be.ktk(K, v=v, out=o), 5)
    sc = torch.empty((n, 1), dtype=torch.float32, device=dev)
    ms_s = _events_ms(lambda: be.mmv(F, Zf, sigma, alpha, None, out=sc), 3)
    flop = 2.0 * n * Zf.n * D
    kbytes = float(be.knm_bytes(n, Zf.n))
    peak = {"h2": F16_MFMA_PEAK_TFLOPS, "f8": F8_MFMA_PEAK_TFLOPS}.get(be.gauss, 157.3)
    tile = 256 if be.gauss == "f8" else (be.lib.odx_gauss_h2_tile(n, Zf.n) if be.gauss == "h2" else 0)
    out["kernels"] = {
        "build": {"kernel": ("gauss_knm_h2w256_kernel<RHS>" if tile == 256 else "gauss_knm_h2s16_kernel + pass") if be.gauss in ("h2", "f8") else "gauss_knm_f32_kernel",
                  "ms": round(ms_b, 3), "TFLOPs": round(flop / ms_b / 1e9, 1), "frac_mfma": round(flop / ms_b / 1e9 / peak, 4),
                  "K_write_GBps": round(kbytes / ms_b / 1e6, 1), "frac_hbm_write": round(kbytes / ms_b / 1e6 / HBM_PEAK_GBS, 4),
                  "bound": "mfma" if flop / (peak * 1e12) > kbytes / (HBM_PEAK_GBS * 1e9) else "hbm (K write)"},
        "pass": {"kernel": "knm_pass_kernel" if K.fmt == "f32" else "knm_passq_kernel", "storage": K.fmt, "ms": round(ms_p, 3), "GBps": round(kbytes / ms_p / 1e6, 1),
                 "frac_hbm": round(kbytes / ms_p / 1e6 / HBM_PEAK_GBS, 4), "bound": "hbm"},
        "score": {"kernel": "gauss_mmv_h2w256_kernel" if tile == 256 else "gauss_mmv_h2s16_kernel", "ms": round(ms_s, 3),
                  "TFLOPs": round(flop / ms_s / 1e9, 1), "frac_mfma": round(flop / ms_s / 1e9 / peak, 4), "bound": "mfma"}}
    del K, Kst
    return out


def config_extras():
    """BASELINE configs 2, 4 and 5 (FALKON legs) at the sizes BASELINE.json states, on one GPU: config 5 as the shard one
    of its 8 GPUs holds (625 000 x 20 000), three of its 100 classes."""
    out = {}
    c2 = dict(C=30, n=100_000, D=1024, M=2000, sigma=15.0, lam=1e-5, classes_run=30, seed=1234 + 2)
    c5 = dict(C=100, n=625_000, D=1024, M=20_000, sigma=15.0, lam=1e-5, classes_run=3, seed=1234 + 5)
    for key, kw in (("config2", dict(name="BASELINE config 2 (FALKON leg)", **c2)),
                    ("config2_bf16", dict(name="BASELINE config 2 (FALKON leg) with K_nM stored as bf16 — throughput only, alpha off by 1e-2..6e-1 "
                                               "(tools/precision_storage_study.py)", storage="bf16", **c2)),
                    ("config4", dict(name="BASELINE config 4 (O-OS mask-pixel rows, one fit per class)", C=21, n=500_000, D=256, M=2000,
                                     sigma=10.0, lam=1e-5, classes_run=21, labels="pixels", seed=1234 + 4)),
                    ("config5_shard", dict(name="BASELINE config 5, the 625 000-row shard of one of 8 GPUs, at f32 accuracy", **c5)),
                    ("config5_shard_f8", dict(name="BASELINE config 5 as stated — fp8 (e4m3) inputs to the X Z' MFMA, f32 accumulate, "
                                                   "throughput only —, the 625 000-row shard of one of 8 GPUs", gauss="f8", **c5))):
        try:
            out[key] = falkon_config_extra(**kw)
        except Exception as e:          # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        import odx
        odx.get_backend().release_workspaces()
        torch.cuda.empty_cache()
    return out


def after_headline():
    """The latency-bound extras measured IN THE HEADLINE PROCESS, straight after its job released its buffers (round-5 review,
    item 6: a user's process will have run a fit before it runs a forward): one-image forwards (R-50-C4, R-50-FPN, f32),
    detect() on one image, the harvest loop per image at its default group size, the default Minibootstrap.  The same keys
    are measured again in a fresh child process by collect(); both figures sit side by side in the bench line."""
    out = {}

    def run(key, fn, pick):
        try:
            r = fn()
            out.update({k2: r.get(k1) for k1, k2 in pick})
        except Exception as e:          # noqa: BLE001
            out[key + "_error"] = "%s: %s" % (type(e).__name__, e)
        torch.cuda.empty_cache()
    run("minibootstrap", lambda: minibootstrap_extra(modes=(("default", None),)), (("s_default", "minibootstrap_s_default"),))
    run("forward", lambda: forward_extra(dtypes=("f32",), groups=(8,)),
        (("ms_per_image_f32", "forward_ms_per_image_f32"), ("ms_per_image_f32_group8", "forward_ms_per_image_f32_group8")))
    run("forward_fpn", lambda: forward_fpn_extra(dtypes=("f32",), groups=(8,)),
        (("ms_per_image_f32", "forward_fpn_ms_per_image_f32"), ("ms_per_image_f32_group8", "forward_fpn_ms_per_image_f32_group8")))
    run("detect", lambda: detect_extra(groups=(8,)), (("ms_per_image", "detect_ms_per_image"), ("ms_per_image_group8", "detect_ms_per_image_group8")))
    run("harvest", lambda: harvest_extra(passes=4, one_image_per_call=False), (("ms_per_image", "harvest_ms_per_image"),))
    out["note"] = ("measured in the headline process right after its job released its buffers; the same keys under forward / "
                   "forward_fpn / detect / harvest / minibootstrap come from a fresh child process")
    return out


def collect(args):
    """Everything above; a failing extra is reported as its error string, never as a missing headline.  The Minibootstrap is
    measured TWICE (round-4 review, item 8): first of all, on the chip the ~150 s headline job has just heated, and last, after
    the other extras and 20 s of idle — `s_default_straight_after_headline` and `s_default_after_idle` sit side by side."""
    out = {}

    def run(key, fn):
        try:
            out[key] = fn()
        except Exception as e:          # noqa: BLE001 — an extra must not take the headline line down with it
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    run("minibootstrap_hot", lambda: minibootstrap_extra(modes=(("default", None),)))
    for key, fn in (("rls", lambda: rls_extra(cpu=not args.no_cpu_baseline)), ("forward", forward_extra), ("forward_fpn", forward_fpn_extra),
                    ("harvest", harvest_extra), ("detect", detect_extra)):
        run(key, fn)
    out.update(config_extras())
    time.sleep(20.0)
    run("minibootstrap", minibootstrap_extra)
    hot = out.pop("minibootstrap_hot")
    if isinstance(out["minibootstrap"], dict) and "s_default" in out["minibootstrap"]:
        out["minibootstrap"]["s_default_after_idle"] = out["minibootstrap"]["s_default"]
        out["minibootstrap"]["s_default_straight_after_headline"] = hot.get("s_default", hot.get("error"))
        out["minibootstrap"]["note"] = ("s_default_straight_after_headline: the first extra, right behind the timed headline job; "
                                        "s_default (= s_default_after_idle): the last one, after the other extras and 20 s without GPU work")
    return out


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
    import json

    class _A:
        no_cpu_baseline = "--no-cpu-baseline" in sys.argv
    import odx
    odx.get_backend()
    print(json.dumps(collect(_A())))
