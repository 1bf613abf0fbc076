"""Shared pieces of the "reference driver, unchanged" tests (tests/test_reference_driver.py) and of the script that made
their fixture (tests/golden/make_driver_golden.py): a tiny seeded network and image stream (handed to the drop-in
FeatureExtractor / AccuracyEvaluator through ODX_MODEL / ODX_SAMPLES, as a driver that passes no cfg_options needs them), the
experiment YAML, the feature cache packed into / unpacked from one array file, a summary of what a run leaves in its output
directory, and `replay` — the call sequence of experiments/run_experiment_online_rpn_ood_oos.py under
`--load_RPN_detector_segmentation_features --save_RPN_detector_segmentation_models`, statement for statement in the
driver's order (:98-120 on-line RPN, :164-247 detector from the caches, :252-266 segmentation, :291-307 evaluation), for the
boxes where the reference itself is not available (the GPU box)."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import torch

C = 2
SEED = 1234
REF_DRIVER = "/root/reference/experiments/run_experiment_online_rpn_ood_oos.py"


def config():
    base = {"NUM_CLASSES": C + 1, "CHOSEN_CLASSES": {i: ("__background__" if i == 0 else "obj%d" % i) for i in range(C + 1)},
            "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                         "CLASSIFIER": {"lambda": 0.001, "sigma": 10, "M": 24, "kernel_type": "gauss"}},
            "ONLINE_SEGMENTATION": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                    "CLASSIFIER": {"lambda": 0.0001, "sigma": 5, "M": 30, "kernel_type": "gauss"}},
            "REGION_REFINER": {"opts": {"lambda": 10.0}},
            "EVALUATION": {"SCORE_THRESH": -2, "NMS": 0.3, "DETECTIONS_PER_IMAGE": 20, "IOU_THRESHOLDS": [0.5]},
            "MINIBOOTSTRAP": {"DETECTOR": {"NUM_CLASSES": C, "ITERATIONS": 2, "BATCH_SIZE": 16, "NEG_IOU_THRESH": 0.3},
                              "RPN": {"NUM_CLASSES": 15, "ITERATIONS": 2, "BATCH_SIZE": 10, "NEG_IOU_THRESH": 0.3, "POS_IOU_THRESH": 0.7}},
            "SEGMENTATION": {"BATCH_SIZE": 400, "SAMPLING_FACTOR": 0.5}, "REGRESSORS": {"MIN_OVERLAP": 0.6}}
    cfg = dict(base)
    rpn = dict(base)
    rpn["CHOSEN_CLASSES"] = {i: "anchor%d" % i for i in range(15)}          # 15 anchor types; is_rpn adds 1 (OnlineRegionClassifier.py:52-53)
    rpn["ONLINE_REGION_CLASSIFIER"] = {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                       "CLASSIFIER": {"lambda": 0.001, "sigma": 25, "M": 12, "kernel_type": "gauss"}}
    rpn["REGION_REFINER"] = {"opts": {"lambda": 0.01}}
    cfg["RPN"] = rpn
    return cfg


def _device():
    import odx
    be = odx.get_backend()
    return torch.device("cuda") if getattr(be, "name", "") == "hip-gfx950" else torch.device("cpu")


def model(cfg_path=None):
    """ODX_MODEL provider: the tiny R-50-C4-shaped network the caches were harvested with (seeded random weights)."""
    from odx.extract import OnlineDetectionModel
    return OnlineDetectionModel(width=4, post_nms_top_n=20, pre_nms_top_n=120, mask_dim=8, seed=3).eval().to(_device())


def samples(split, cfg_path=None):
    """ODX_SAMPLES provider: seeded synthetic images with 1-2 boxes and box-shaped masks."""
    n, seed = (6, 5) if split == "train" else (4, 5)        # (the test images are the first training images: the models have seen them)
    g = torch.Generator().manual_seed(seed)
    out = []
    H, W = 96, 128
    for i in range(n):
        img = torch.randn(1, 3, H, W, generator=g)
        G = 1 + i % 2
        xy = torch.rand(G, 2, generator=g) * torch.tensor([W * 0.5, H * 0.5])
        gt = torch.cat([xy, xy + 20 + torch.rand(G, 2, generator=g) * torch.tensor([W * 0.4, H * 0.4])], 1)
        labels = [1 + (i + k) % C for k in range(G)]
        masks = torch.zeros(G, H, W, dtype=torch.uint8)
        for k, b in enumerate(gt):
            x1, y1, x2, y2 = [int(v) for v in b]
            masks[k, y1 + 2:max(y2 - 2, y1 + 3), x1 + 2:max(x2 - 2, x1 + 3)] = 1
        out.append((img, gt, labels, masks))
    return out


CACHE_DIRS = ("features_RPN", "features_detector", "features_segmentation")


def pack_cache(out_dir):
    """{relative file name: array} of every cache file under out_dir."""
    arrays = {}
    for d in CACHE_DIRS:
        for f in sorted(os.listdir(os.path.join(out_dir, d))):
            arrays["%s/%s" % (d, f)] = torch.load(os.path.join(out_dir, d, f), map_location="cpu").numpy()
    return arrays


def unpack_cache(arrays, out_dir, device="cpu"):
    """The cache files as a harvest on `device` would have left them (torch.save'd tensors of that device)."""
    for name, a in arrays.items():
        d, f = name.split("/")
        os.makedirs(os.path.join(out_dir, d), exist_ok=True)
        torch.save(torch.from_numpy(np.asarray(a)).to(device), os.path.join(out_dir, d, f))


def summarize(out_dir):
    """What a run left behind: the model files (structure, shapes, norms, a few leading values) and the result.txt lines that
    do not carry a wall-clock time."""
    from odx import storage
    out = {"files": sorted(f for f in os.listdir(out_dir) if os.path.isfile(os.path.join(out_dir, f)))}
    for tag in ("rpn", "detector", "segmentation"):
        clf, reg, stats = storage.load_models(out_dir, tag, map_location="cpu")
        ent = {"classifiers": []}
        for m in clf:
            if m is None:
                ent["classifiers"].append(None)
                continue
            a = m.alpha_.detach().cpu().double().reshape(-1).numpy()
            ent["classifiers"].append({"M": int(m.ny_points_.shape[0]), "D": int(m.ny_points_.shape[1]), "alpha_shape": list(m.alpha_.shape),
                                       "alpha_norm": float(np.linalg.norm(a)), "alpha_head": [float(v) for v in a[:4]],
                                       "centres_sum": float(m.ny_points_.detach().cpu().double().sum())})
        if reg is not None:
            ent["regressors"] = []
            for r in reg:
                if r["Beta"] is None or r["Beta"]["0"] is None:
                    ent["regressors"].append(None)
                    continue
                W = torch.stack([r["Beta"][str(k)]["weights"].detach().cpu().double() for k in range(4)])
                ent["regressors"].append({"weights_shape": list(W.shape), "weights_norm": float(W.norm()),
                                          "mu": [float(v) for v in r["mu"].cpu()], "losses_rows": int(r["Beta"]["0"]["losses"].shape[0])})
        ent["stats"] = {"mean_norm": float(stats["mean_norm"]), "mean_abs_sum": float(stats["mean"].cpu().double().abs().sum()),
                        "dim": int(stats["mean"].shape[0])}
        out[tag] = ent
    lines = open(os.path.join(out_dir, "result.txt")).read().splitlines()
    out["result_lines"] = [l for l in lines if l.strip() and "time" not in l.lower()]
    return out


def _stats_on_the_host(u, positives, negatives, dim, pos_fraction):
    """computeFeatStatistics_torch on host copies of the rows (the same index draws from the global CPU generator — the index
    tensors are host tensors either way — and the same f32 reductions as a `--CPU` run), handed back on the rows' device:
    lets the GPU arm of the driver test start every fit from bit-identical statistics, so that what is compared is the fit."""
    dev = positives[0].device if len(positives) else torch.device("cpu")
    st = u.computeFeatStatistics_torch([p.cpu() for p in positives], [[b.cpu() for b in nb] for nb in negatives],
                                       features_dim=dim, cpu_tensor=True, pos_fraction=pos_fraction)
    return {k: v.to(dev) for k, v in st.items()}


def replay(out_dir, cfg_path, cpu, host_stats=False):
    """The driver's statements for `--load_RPN_detector_segmentation_features --save_RPN_detector_segmentation_models`
    (+ `--CPU` when cpu), in its order, on the drop-in modules imported the way it imports them.  host_stats (GPU arm only):
    the three feature statistics are reduced on the host, as in the reference run the fixture was made from — every fit then
    sees bit-identical normalised rows (the normalisation itself is elementwise f32: the same on both devices)."""
    from tests import dropin
    FeatureExtractor = dropin.load("feature_extractor").FeatureExtractor
    AccuracyEvaluator = dropin.load("accuracy_evaluator").AccuracyEvaluator
    RegionRefiner = dropin.load("region_refiner").RegionRefiner
    u = dropin.load("py_od_utils")
    ocr = dropin.load("OnlineRegionClassifier" if cpu else "OnlineRegionClassifier_incore")
    falkon = dropin.load("FALKONWrapper_with_centers_selection" if cpu else "FALKONWrapper_with_centers_selection_incore")
    pos_fraction_feat_stats = 0.8
    training_device = "cpu" if cpu else "cuda"
    with redirect_stdout(io.StringIO()):
        FeatureExtractor(cfg_path, train_in_cpu=cpu)                                                              # :72
        positives_RPN, negatives_RPN = u.load_features_classifier(features_dir=os.path.join(out_dir, "features_RPN"), cfg_feature_extraction=cfg_path)
        if host_stats and not cpu:
            stats_rpn = _stats_on_the_host(u, positives_RPN, negatives_RPN, positives_RPN[0].size()[1], pos_fraction_feat_stats)
        else:
            stats_rpn = u.computeFeatStatistics_torch(positives_RPN, negatives_RPN, features_dim=positives_RPN[0].size()[1], cpu_tensor=cpu,
                                                      pos_fraction=pos_fraction_feat_stats)
        classifier = falkon.FALKONWrapper(cfg_path=cfg_path, is_rpn=True)
        rc = ocr.OnlineRegionClassifier(classifier, positives_RPN, negatives_RPN, stats_rpn, cfg_path=cfg_path, is_rpn=True)
        models_falkon_rpn = u.falkon_models_to_cuda(rc.trainRegionClassifier(opts={"is_rpn": True}, output_dir=out_dir))
        region_refiner = RegionRefiner(cfg_path, is_rpn=True)
        COXY_RPN = u.load_features_regressor(features_dir=os.path.join(out_dir, "features_RPN"))
        models_reg_rpn = region_refiner.trainRegionRefiner(u.normalize_COXY(COXY_RPN, stats_rpn, cpu), output_dir=out_dir)
        torch.save(models_falkon_rpn, os.path.join(out_dir, "classifier_rpn"))
        torch.save(models_reg_rpn, os.path.join(out_dir, "regressor_rpn"))
        torch.save(stats_rpn, os.path.join(out_dir, "stats_rpn"))
        # detector, from the caches (:164-247 with use_only_gt_positives_detection and normalize_features_regressor_detector off)
        region_refiner = RegionRefiner(cfg_path)
        COXY = u.load_features_regressor(features_dir=os.path.join(out_dir, "features_detector"))
        for k in ("C", "X", "Y"):
            COXY[k] = COXY[k].to(training_device)
        models = region_refiner.trainRegionRefiner(COXY, output_dir=out_dir)
        positives, negatives = u.load_features_classifier(features_dir=os.path.join(out_dir, "features_detector"), cfg_feature_extraction=cfg_path)
        positives = u.load_positives_from_COXY(COXY, samples_fraction=1.0)
        for i in range(len(positives)):
            positives[i] = positives[i].to(training_device)
            for j in range(len(negatives[i])):
                negatives[i][j] = negatives[i][j].to(training_device)
        if host_stats and not cpu:
            stats = _stats_on_the_host(u, positives, negatives, negatives[0][0].size()[1], pos_fraction_feat_stats)
        else:
            stats = u.computeFeatStatistics_torch(positives, negatives, features_dim=negatives[0][0].size()[1], cpu_tensor=cpu,
                                                  pos_fraction=pos_fraction_feat_stats)
        classifier = falkon.FALKONWrapper(cfg_path=cfg_path)
        rc = ocr.OnlineRegionClassifier(classifier, positives, negatives, stats, cfg_path=cfg_path)
        model_det = u.falkon_models_to_cuda(rc.trainRegionClassifier(output_dir=out_dir))
        torch.save(model_det, os.path.join(out_dir, "classifier_detector"))
        torch.save(models, os.path.join(out_dir, "regressor_detector"))
        torch.save(stats, os.path.join(out_dir, "stats_detector"))
        # segmentation (:252-266)
        ps, ns = u.load_features_classifier(features_dir=os.path.join(out_dir, "features_segmentation"), is_segm=True, sample_ratio=0.3)
        for i in range(len(ps)):
            ps[i] = ps[i].to(training_device)
            ns[i] = [ns[i].to(training_device)]
        if host_stats and not cpu:
            stats_segm = _stats_on_the_host(u, ps, ns, ps[0].size()[1], pos_fraction_feat_stats)
        else:
            stats_segm = u.computeFeatStatistics_torch(ps, ns, features_dim=ps[0].size()[1], cpu_tensor=cpu, pos_fraction=pos_fraction_feat_stats)
        classifier = falkon.FALKONWrapper(cfg_path=cfg_path, is_segmentation=True)
        rc = ocr.OnlineRegionClassifier(classifier, ps, ns, stats_segm, cfg_path=cfg_path, is_segmentation=True)
        model_segm = u.falkon_models_to_cuda(rc.trainRegionClassifier(output_dir=out_dir))
        torch.save(model_segm, os.path.join(out_dir, "classifier_segmentation"))
        torch.save(stats_segm, os.path.join(out_dir, "stats_segmentation"))
        # evaluation (:291-307)
        ae = AccuracyEvaluator(cfg_path, train_in_cpu=cpu)
        ae.falkon_rpn_models, ae.regressors_rpn_models, ae.stats_rpn = models_falkon_rpn, models_reg_rpn, stats_rpn
        ae.falkon_detector_models, ae.regressors_detector_models, ae.stats_detector = model_det, models, stats
        ae.falkon_segmentation_models, ae.stats_segmentation = model_segm, stats_segm
        return ae.evaluateAccuracyDetection(is_train=False, output_dir=out_dir, eval_segm_with_gt_bboxes=False,
                                            normalize_features_regressors=False, evaluate_segmentation_icwt=True)
