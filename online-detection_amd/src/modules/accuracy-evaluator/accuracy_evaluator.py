"""Drop-in for the reference's accuracy-evaluator/accuracy_evaluator.py: class AccuracyEvaluator with the same constructor,
settable model attributes and evaluateAccuracyDetection entry point
(src/modules/accuracy-evaluator/accuracy_evaluator.py:11-41 -> accuracy_evaluator_detector/evaluate_accuracy_detector.py:34-196
-> mrcnn_modified/engine/inference.py:268-357 -> icw_eval.py:150-224).

What it does: runs every test image through the detection network with the trained on-line heads injected (FALKON RPN,
FALKON + RLS detector, FALKON mask pixels), post-processes in the original image frame (OnlineDetectionPostProcessor:
decode, clip, score threshold, per-class NMS on the MI355X, top-k), and scores the detections with the VOC-style
precision / recall / AP the reference uses, writing the reference's `result.txt` lines ("Detection mAP50: ..." per
class, "Segmentation mAP50: ..." when masks are evaluated).

As for the FeatureExtractor drop-in, dataset classes, images and pretrained weights are outside this repository
(SURVEY §2 rows 17-19): the image stream and (optionally) the network come in through `cfg_options`:
    cfg_options['samples']  iterable of (image, gt_boxes (G, 4), gt_labels list[int][, gt_masks (G, H, W)][, difficult])
                            image: (1, 3, H, W) pre-processed float tensor, or (H, W, 3) uint8 BGR as the reference
                            reads it (then odx.extract.preprocess_image applies the reference transform)
    cfg_options['model']    an odx.extract.OnlineDetectionModel (default: R-50-C4 with seeded random weights; load a
                            reference checkpoint with odx.extract.load_reference_checkpoint)
    (neither given, as in the reference's unchanged drivers: ODX_SAMPLES / ODX_MODEL = "module:callable", odx/providers.py)
    cfg_options['class_names']  names for the per-class lines (default: the YAML's CHOSEN_CLASSES, else "class_i")
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir, os.path.pardir)))
import _odx_path  # noqa: F401,E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from odx.contracts import AccuracyEvaluatorAbstract  # noqa: E402
from odx.extract import OnlineDetectionModel, detect_batch, preprocess_image  # noqa: E402
from odx.postprocess import eval_detection  # noqa: E402


class AccuracyEvaluator(AccuracyEvaluatorAbstract):
    def __init__(self, cfg_path_target_task=None, cfg_path_RPN=None, train_in_cpu=False):
        self.cfg_path_target_task = cfg_path_target_task
        self.cfg_path_RPN = cfg_path_RPN
        self.falkon_rpn_models = None
        self.regressors_rpn_models = None
        self.stats_rpn = None
        self.falkon_detector_models = None
        self.regressors_detector_models = None
        self.stats_detector = None
        self.falkon_segmentation_models = None
        self.stats_segmentation = None
        self.regions_post_nms = None
        self.train_in_cpu = train_in_cpu

    def _cfg(self):
        p = self.cfg_path_target_task
        if p is None or not os.path.exists(p):
            return {}
        with open(p) as fid:
            return yaml.load(fid, Loader=yaml.FullLoader) or {}

    def _model(self, cfg_options, normalize_features_regressors):
        model = cfg_options.get('model')
        if model is None:
            model = OnlineDetectionModel()
            if torch.cuda.is_available():
                model = model.cuda()
        model.eval()
        if self.regions_post_nms is not None:
            model.post_nms_top_n = self.regions_post_nms
        # evaluate_accuracy_detector.py:131-150: heads receive classifiers / regressors / stats
        model.update_model(
            models_rpn={'classifiers': self.falkon_rpn_models, 'regressors': self.regressors_rpn_models, 'stats': self.stats_rpn}
            if self.falkon_rpn_models is not None else None,
            models_detection={'classifiers': self.falkon_detector_models, 'regressors': self.regressors_detector_models,
                              'stats': self.stats_detector} if self.falkon_detector_models is not None else None,
            models_segmentation={'classifiers': self.falkon_segmentation_models, 'stats': self.stats_segmentation}
            if self.falkon_segmentation_models is not None else None)
        if model.online_box is not None:
            model.online_box.normalize_features_regressors = normalize_features_regressors
        return model

    def evaluateAccuracyDetection(self, is_train, output_dir=None, save_features=False, evaluate_segmentation=True,
                                  eval_segm_with_gt_bboxes=False, normalize_features_regressors=False,
                                  evaluate_segmentation_icwt=False, cfg_options={}):
        from odx import providers
        cfg_options = providers.fill(cfg_options, 'train' if is_train else 'test', self.cfg_path_target_task)      # ODX_SAMPLES / ODX_MODEL
        if 'samples' not in cfg_options:
            raise NotImplementedError("dataset loading (iCWT / YCB-V / HO-3D through maskrcnn_benchmark) is outside this "
                                      "repository: pass the test images as cfg_options['samples']")
        cfg = self._cfg()
        ev = cfg.get('EVALUATION') or {}
        thresholds = tuple(ev.get('IOU_THRESHOLDS', (0.5,)))
        use_07 = bool(ev.get('USE_VOC07_METRIC', True))
        score_thresh, nms, per_img = ev.get('SCORE_THRESH', -2.0), ev.get('NMS', 0.3), ev.get('DETECTIONS_PER_IMAGE', 100)
        model = self._model(cfg_options, normalize_features_regressors)
        dev = next(model.parameters()).device
        do_masks = bool(evaluate_segmentation or evaluate_segmentation_icwt) and model.online_mask is not None
        preds, gts = [], []
        # the reference's test loop walks one image per iteration (engine/inference.py:268-357); consecutive images of one
        # pre-processed size go through the network together here (extract.detect_batch, cfg_options['trunk_batch'] at a time,
        # default 4): one forward and one pass of the on-line heads per group, the post-processing per image
        k = max(1, int(cfg_options.get('trunk_batch', 4))) if dev.type == "cuda" else 1
        pending = []                                               # (image (1, 3, H, W), original size, wants masks)

        def flush():
            if not pending:
                return
            images = torch.cat([p[0] for p in pending], dim=0)
            want = any(p[2] for p in pending)
            for (res, _), (_, orig, want_masks) in zip(detect_batch(model, images, [p[1] for p in pending], score_thresh, nms, per_img,
                                                                    with_masks=want), pending):
                if res is None:
                    res = {"boxes": torch.zeros((0, 4)), "scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.int64)}
                if not want_masks:
                    res.pop("masks", None)
                elif "masks" not in res:                           # an image without detections still counts its ground truth
                    res["masks"] = torch.zeros((0, orig[1], orig[0]), dtype=torch.uint8)
                preds.append({k2: v.cpu().numpy() for k2, v in res.items()})
            del pending[:]

        for sample in cfg_options['samples']:
            image, gt_boxes, gt_labels = sample[0], torch.as_tensor(sample[1]).float().reshape(-1, 4), list(sample[2])
            gt_masks = sample[3] if len(sample) > 3 else None
            difficult = sample[4] if len(sample) > 4 else None
            image = image.to(dev)
            if image.dim() == 3:                                   # (H, W, 3) uint8 BGR as read by the reference
                orig = (int(image.shape[1]), int(image.shape[0]))
                image, _ = preprocess_image(image, min_size=(cfg.get('INPUT') or {}).get('MIN_SIZE_TEST', 600))
            else:
                orig = (int(image.shape[3]), int(image.shape[2]))
            if pending and (len(pending) >= k or tuple(pending[0][0].shape) != tuple(image.shape)):
                flush()
            pending.append((image, orig, do_masks and gt_masks is not None))
            g = {"boxes": gt_boxes.numpy(), "labels": np.asarray(gt_labels, dtype=np.int64)}
            if difficult is not None:
                g["difficult"] = np.asarray(difficult, dtype=bool)
            if gt_masks is not None:
                g["masks"] = torch.as_tensor(gt_masks).cpu().numpy()
            gts.append(g)
        flush()
        names = cfg_options.get('class_names') or cfg.get('CHOSEN_CLASSES') or {}
        result = None
        for thr in thresholds:
            result = eval_detection(preds, gts, thr, use_07, key="boxes")
            self._report("Detection", thr, result, names, output_dir)
            if do_masks and all("masks" in p for p in preds) and all("masks" in g for g in gts):
                result = eval_detection(preds, gts, thr, use_07, key="masks")
                self._report("Segmentation", thr, result, names, output_dir)
        return result

    @staticmethod
    def _report(kind, thr, result, names, output_dir):
        # icw_eval.py:184-222: "<kind> mAP50: 0.1234", a blank line, one "<class name, 26 wide>: 0.1234" per foreground class
        text = "{} mAP{}: {:.4f}\n\n".format(kind, int(thr * 100), result["map"])
        for i, ap in enumerate(result["ap"]):
            if i == 0:
                continue
            name = names[i] if (isinstance(names, (list, tuple)) and i < len(names)) else (names.get(i, "class_%d" % i) if isinstance(names, dict) else "class_%d" % i)
            text += "{:<26}: {:.4f}\n".format(name, ap)
        text += "\n"
        print(text)
        if output_dir:
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                fid.write(text)
