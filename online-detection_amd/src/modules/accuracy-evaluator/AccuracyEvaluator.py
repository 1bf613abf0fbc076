"""Drop-in for the reference's stand-alone evaluator, accuracy-evaluator/AccuracyEvaluator.py (reference :11-43, with
OnlineDetectionPostProcessor_standalone.py:10-103): `AccuracyEvaluator(cfg_path, output_folder).evaluate(dataset,
predictions, ...)` for the stand-alone O-OD pipeline (run_experiment_ood_standalone-style drivers), where `predictions`
are the per-image BoxLists left by OnlineRegionClassifier.testRegionClassifier (field `scores` (R, C+1)) and
RegionRefiner.predict (`bbox` (R, C+1, 4) refined boxes, or (R, 4) class-agnostic ones).

Post-processing (clip, score threshold, per-class NMS on the MI355X, global top-k) is odx.postprocess.filter_results; the
scoring is the reference's VOC-style AP (odx.postprocess.eval_detection), reported in its result.txt format.  The
ground truth comes from the dataset object by duck typing — `dataset.get_groundtruth(i)` returning an object with `.bbox`
(G, 4) and fields `labels` (and optionally `difficult`), as maskrcnn_benchmark's dataset classes do; datasets themselves
are outside this repository."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir, os.path.pardir)))
import _odx_path  # noqa: F401,E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from odx.postprocess import eval_detection, filter_results  # noqa: E402


class AccuracyEvaluator():
    def __init__(self, cfg_path, output_folder):
        cfg = yaml.load(open(cfg_path), Loader=yaml.FullLoader)
        self.score_thresh = cfg['EVALUATION']['SCORE_THRESH']
        self.nms = cfg['EVALUATION']['NMS']
        self.detections_per_img = cfg['EVALUATION']['DETECTIONS_PER_IMAGE']
        self.num_classes = cfg['NUM_CLASSES']
        self.iou_thresholds = tuple(cfg['EVALUATION'].get('IOU_THRESHOLDS', (0.5,)))
        self.use_07_metric = bool(cfg['EVALUATION'].get('USE_VOC07_METRIC', True))
        self.class_names = cfg.get('CHOSEN_CLASSES') or {}
        self.output_folder = output_folder

    def evaluate(self, dataset, predictions, cls_agnostic_bbox_reg=True, box_only=False, iou_types=("bbox",),
                 expected_results=(), draw_preds=False, expected_results_sigma_tol=4, is_target_task=False, icwt_21_objs=False):
        print('Evaluating predictions')
        dev = 'cuda' if torch.cuda.is_available() else 'cpu'
        preds, gts = [], []
        for i, p in enumerate(predictions):
            scores = torch.as_tensor(p.get_field('scores')).to(dev).float()
            boxes = torch.as_tensor(p.bbox).to(dev).float().reshape(scores.shape[0], -1)
            res = filter_results(boxes, scores[:, :self.num_classes], p.size, self.score_thresh, self.nms, self.detections_per_img)
            if res is None:
                res = {"boxes": torch.zeros((0, 4)), "scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.int64)}
            preds.append({k: v.cpu().numpy() for k, v in res.items()})
            g = dataset.get_groundtruth(i)
            gt = {"boxes": np.asarray(torch.as_tensor(g.bbox).cpu(), dtype=np.float32).reshape(-1, 4),
                  "labels": np.asarray(torch.as_tensor(g.get_field("labels")).cpu(), dtype=np.int64)}
            if hasattr(g, "has_field") and g.has_field("difficult"):
                gt["difficult"] = np.asarray(torch.as_tensor(g.get_field("difficult")).cpu()).astype(bool)
            gts.append(gt)
        result = None
        for thr in self.iou_thresholds:
            result = eval_detection(preds, gts, thr, self.use_07_metric)
            text = "Detection mAP{}: {:.4f}\n\n".format(int(thr * 100), result["map"])
            for c, ap in enumerate(result["ap"]):
                if c == 0:
                    continue
                name = self.class_names.get(c, "class_%d" % c) if isinstance(self.class_names, dict) else self.class_names[c]
                text += "{:<26}: {:.4f}\n".format(name, ap)
            text += "\n"
            print(text)
            if self.output_folder:
                with open(os.path.join(self.output_folder, "result.txt"), "a") as fid:
                    fid.write(text)
        return result
