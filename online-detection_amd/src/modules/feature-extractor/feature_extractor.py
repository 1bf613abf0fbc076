"""Drop-in for the reference's feature-extractor/feature_extractor.py: class FeatureExtractor with the
same constructor, settable model attributes and extract* entry points
(src/modules/feature-extractor/feature_extractor.py:10-81).

The reference's extractors read images and annotations through maskrcnn_benchmark dataset classes
and load Detectron weights; datasets, weights and that package are outside this repository's scope
(SURVEY §2 rows 17-19), so the image stream and (optionally) the network are handed in through
`cfg_options`:
    cfg_options['samples']  iterable of (image (1, 3, H, W) float tensor, gt_boxes (G, 4), gt_labels list[int])
    cfg_options['model']    an odx.extract.OnlineDetectionModel (default: R-50-C4, seeded random weights)
The MINIBOOTSTRAP / REGRESSORS values are read from the feature-extraction YAML when present.
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir, os.path.pardir)))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir)))
import _odx_path  # noqa: F401,E402
import torch  # noqa: E402
import yaml  # noqa: E402
from FeatureExtractorAbstract import FeatureExtractorAbstract  # noqa: E402
from odx.extract import DetectorFeatureExtractor, OnlineDetectionModel  # noqa: E402
from odx.heads import OnlineBoxPredictor, OnlineRPNHead  # noqa: E402


def _mb(cfg, key, default, section='DETECTOR'):
    try:
        return cfg['MINIBOOTSTRAP'][section][key]
    except Exception:
        return default


class FeatureExtractor(FeatureExtractorAbstract):
    def __init__(self, cfg_path_target_task=None, cfg_path_RPN=None, cfg_path_feature_task=None, train_in_cpu=False):
        self.cfg_path_target_task = cfg_path_target_task
        self.cfg_path_RPN = cfg_path_RPN
        self.cfg_path_feature_task = cfg_path_feature_task
        self.falkon_rpn_models = None
        self.regressors_rpn_models = None
        self.stats_rpn = None
        self.falkon_detector_models = None
        self.regressors_detector_models = None
        self.stats_detector = None
        self.regions_post_nms = None
        self.train_in_cpu = train_in_cpu
        self.start_of_feature_extraction_time = None
        self.end_of_feature_extraction_time = None
        self.start_of_feature_extraction_time_RPN = None
        self.start_of_feature_extraction_time_detection = None

    def _cfg(self, path):
        if path is None or not os.path.exists(path):
            return {}
        with open(path) as fid:
            return yaml.load(fid, Loader=yaml.FullLoader) or {}

    def _model(self, cfg_options):
        model = cfg_options.get('model')
        if model is None:
            model = OnlineDetectionModel()
            if torch.cuda.is_available():
                model = model.cuda()
        model.eval()
        if self.regions_post_nms is not None:
            model.post_nms_top_n = self.regions_post_nms
        if self.falkon_rpn_models is not None:      # on-line RPN injected (evaluate_accuracy_detector.py:131-150)
            model.online_rpn = OnlineRPNHead(self.falkon_rpn_models, self.regressors_rpn_models, self.stats_rpn)
        if self.falkon_detector_models is not None:
            model.online_box = OnlineBoxPredictor(self.falkon_detector_models, self.regressors_detector_models,
                                                  self.stats_detector)
        return model

    def extractFeatures(self, is_train, output_dir=None, save_features=False, extract_features_segmentation=False,
                        use_only_gt_positives_detection=True, cfg_options={}):
        if 'samples' not in cfg_options:
            raise NotImplementedError(
                "dataset loading (iCWT / YCB-V / HO-3D through maskrcnn_benchmark) is outside this repository: pass the "
                "image stream as cfg_options['samples'] = [(image, gt_boxes, gt_labels), ...]")
        if extract_features_segmentation:
            raise NotImplementedError("mask-head feature harvesting (A13) is not built yet")
        cfg = self._cfg(self.cfg_path_target_task)
        model = self._model(cfg_options)
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ.get('WORLD_SIZE', '1'))
        ex = DetectorFeatureExtractor(model, num_classes=_mb(cfg, 'NUM_CLASSES', cfg_options.get('num_classes', 30)),
                                      iterations=_mb(cfg, 'ITERATIONS', 10), batch_size=_mb(cfg, 'BATCH_SIZE', 2000),
                                      neg_iou_thresh=_mb(cfg, 'NEG_IOU_THRESH', 0.3),
                                      reg_min_overlap=(cfg.get('REGRESSORS') or {}).get('MIN_OVERLAP', 0.6),
                                      shuffle_negatives=_mb(cfg, 'SHUFFLE_NEGATIVES', False), rank=rank, world=world)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.start_of_feature_extraction_time_detection = time.time()
        if not is_train:
            return ex.test(cfg_options['samples'])
        negatives, positives, COXY = ex.train(cfg_options['samples'], use_only_gt_positives_detection)
        if output_dir:
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                dt = time.time() - self.start_of_feature_extraction_time_detection
                fid.write("Detector's feature extraction time: {}min:{}s \n".format(int(dt / 60), round(dt % 60)))
        return negatives, positives, COXY

    def extractRPNFeatures(self, is_train, output_dir=None, save_features=False, cfg_options={}):
        raise NotImplementedError("on-line RPN feature harvesting (A12) is not built yet")

    def extractFeaturesRPNDetector(self, is_train, output_dir=None, save_features=False, extract_features_segmentation=False,
                                   use_only_gt_positives_detection=True, cfg_options={}):
        raise NotImplementedError("joint RPN + detector harvesting (A12 / A13) is not built yet")

    def trainFeatureExtractor(self, *args, **kwargs):
        raise NotImplementedError("SGD training of the Mask R-CNN baselines is out of scope (SURVEY §2 row 21)")

    def testFeatureExtractor(self, *args, **kwargs):
        raise NotImplementedError("the Mask R-CNN baseline tester is out of scope (SURVEY §2 row 21)")
