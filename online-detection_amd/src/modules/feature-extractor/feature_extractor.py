"""Drop-in for the reference's feature-extractor/feature_extractor.py: class FeatureExtractor with the
same constructor, settable model attributes and extract* entry points
(src/modules/feature-extractor/feature_extractor.py:10-81).

The reference's extractors read images and annotations through maskrcnn_benchmark dataset classes
and load Detectron weights; datasets, weights and that package are outside this repository's scope
(SURVEY §2 rows 17-19), so the image stream and (optionally) the network are handed in through
`cfg_options`:
    cfg_options['samples']  iterable of (image (1, 3, H, W) float tensor, gt_boxes (G, 4), gt_labels list[int])
    cfg_options['model']    an odx.extract.OnlineDetectionModel (default: R-50-C4, seeded random weights) or an
                            odx.fpn.OnlineDetectionModelFPN; cfg_options['conv_body'] = 'R-50-FPN' builds the latter
                            (MODEL.BACKBONE.CONV_BODY of the reference's YAMLs, config/defaults.py:99).  The FPN network
                            serves the detector's features / heads; the reference defines its on-line RPN and mask heads
                            on R-50-C4 only
    cfg_options['shard_images']  True: under a multi-process launcher each rank harvests only its images (rank::world);
                                 False — every rank sees every image, so the drop-in trainers downstream (which
                                 are not sharded) build identical models on all ranks.  Default: True exactly when the
                                 caller hands in the row shard it trains with (cfg_options['shard'], an odx.dist.RowShard
                                 over more than one rank), False otherwise
    cfg_options['shard']         the odx.dist.RowShard the caller's sharded trainers use (falkon_fit(shard=),
                                 RegionRefinerTrainer(shard=)): images are then split over its ranks by default
    cfg_options['trunk_batch']   images of one size that share ONE forward (trunk, proposals, RoI head) in the harvest loop (default 8; 1 = one image
                                 per call).  With > 1 an image's features depend on its neighbour in the list in the
                                 last bits (the convolution library picks its algorithm per batch size) and so differ
                                 in rounding from the one-image detect() / forward() path used at test time: pass 1 for
                                 runs whose harvested rows must be bit-reproducible per image
A driver that passes neither (the reference's own, unchanged) gets both through the environment: ODX_SAMPLES /
ODX_MODEL = "module:callable" (odx/providers.py).
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
from odx.extract import DetectorFeatureExtractor, OnlineDetectionModel, OnlineFeatureExtractor  # noqa: E402
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
            if str(cfg_options.get('conv_body', 'R-50-C4')).upper() == 'R-50-FPN':
                from odx.fpn import OnlineDetectionModelFPN
                model = OnlineDetectionModelFPN()
            else:
                model = OnlineDetectionModel()
            if torch.cuda.is_available():
                model = model.cuda()
        model.eval()
        if self.regions_post_nms is not None:
            model.post_nms_top_n = self.regions_post_nms
        if self.falkon_rpn_models is not None:      # on-line RPN injected (evaluate_accuracy_detector.py:131-150)
            if not hasattr(model, "rpn_activation"):
                raise NotImplementedError("on-line RPN models need the R-50-C4 network (the reference's on-line RPN has 15 "
                                          "anchor types on one stride-16 map)")
            model.online_rpn = OnlineRPNHead(self.falkon_rpn_models, self.regressors_rpn_models, self.stats_rpn)
        if self.falkon_detector_models is not None:
            model.online_box = OnlineBoxPredictor(self.falkon_detector_models, self.regressors_detector_models,
                                                  self.stats_detector)
        return model

    _NEED_SAMPLES = ("dataset loading (iCWT / YCB-V / HO-3D through maskrcnn_benchmark) is outside this repository: pass the "
                     "image stream as cfg_options['samples'] = [(image, gt_boxes, gt_labels[, masks]), ...]")

    def _kw(self, cfg, section):
        kw = {}
        for key, name in (('ITERATIONS', 'iterations'), ('BATCH_SIZE', 'batch_size'), ('NEG_IOU_THRESH', 'neg_iou_thresh'),
                          ('SHUFFLE_NEGATIVES', 'shuffle_negatives'), ('POS_IOU_THRESH', 'pos_iou_thresh')):
            v = _mb(cfg, key, None, section)
            if v is not None and not (section == 'DETECTOR' and key == 'POS_IOU_THRESH'):
                kw[name] = v
        return kw

    def _harvest(self, cfg_path, parts, is_train, use_only_gt_positives, cfg_options, output_dir, label, save_features=False):
        from odx import providers
        cfg_options = providers.fill(cfg_options, 'train' if is_train else 'test', cfg_path)     # ODX_SAMPLES / ODX_MODEL
        if 'samples' not in cfg_options:
            raise NotImplementedError(self._NEED_SAMPLES)
        cfg = self._cfg(cfg_path)
        model = self._model(cfg_options)
        # Images are sharded over the ranks of a launcher ONLY on request (cfg_options['shard_images'] = True): the trainers
        # these features go to (OnlineRegionClassifier, RegionRefiner, computeFeatStatistics_torch) see the rows they are
        # given, so a silent split would train every rank's models on 1 / world of the data.  With the option every rank
        # returns the rows of its own images (rank::world) and the caller is responsible for training on shards
        # (odx.solver.falkon_fit(shard=...), RegionRefinerTrainer(shard=...)) — the multi-GPU route bench.py exercises.
        # When the caller DOES shard its training — it hands in the odx.dist.RowShard it trains with as cfg_options['shard']
        # (an initialised process group of more than one rank) — the split is the default: the rows a rank harvests are then
        # its row shard of the fit, which is the design (north star: "the backbone forward shards over images the same way").
        # An explicit cfg_options['shard_images'] = False / True always wins.
        rank, world = 0, 1
        shard = cfg_options.get('shard')
        want = cfg_options.get('shard_images')
        if want is None:
            want = shard is not None and getattr(shard, 'world', 1) > 1
        if want:
            if shard is not None and getattr(shard, 'world', 1) > 1:
                rank, world = int(shard.rank), int(shard.world)
            elif torch.distributed.is_available() and torch.distributed.is_initialized():
                rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
            else:
                rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
        num_classes = _mb(cfg, 'NUM_CLASSES', cfg_options.get('num_classes', 30))
        det_kw = self._kw(cfg, 'DETECTOR')
        det_kw['reg_min_overlap'] = (cfg.get('REGRESSORS') or {}).get('MIN_OVERLAP', 0.6)
        seg = cfg.get('SEGMENTATION') or {}
        mask_kw = {k2: seg[k1] for k1, k2 in (('BATCH_SIZE', 'batch_size'), ('SAMPLING_FACTOR', 'sampling_factor')) if k1 in seg}
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.time()
        if not is_train:
            ex = DetectorFeatureExtractor(model, num_classes=num_classes, rank=rank, world=world, **{
                'iterations': det_kw.get('iterations', 10), 'batch_size': det_kw.get('batch_size', 2000)})
            return t0, ex.test(cfg_options['samples'])
        ex = OnlineFeatureExtractor(model, num_classes, parts=parts, det=det_kw, rpn=self._kw(cfg, 'RPN'), mask=mask_kw,
                                    rank=rank, world=world, trunk_batch=int(cfg_options.get('trunk_batch', 8)))
        if save_features and not output_dir:
            raise ValueError('Output directory must be specified.')       # the reference prints this and quits
        out = ex.train(cfg_options['samples'], use_only_gt_positives, save_dir=output_dir if save_features else None)
        if output_dir:
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                dt = time.time() - t0
                fid.write("{} feature extraction time: {}min:{}s \n".format(label, int(dt / 60), round(dt % 60)))
        return t0, out

    def extractFeatures(self, is_train, output_dir=None, save_features=False, extract_features_segmentation=False,
                        use_only_gt_positives_detection=True, cfg_options={}):
        """-> negatives, positives, COXY[, segmentation negatives, positives]  (extract_features_detector.py:278-292);
        test time -> test_boxes."""
        parts = ("detector", "mask") if extract_features_segmentation else ("detector",)
        t0, out = self._harvest(self.cfg_path_target_task, parts, is_train, use_only_gt_positives_detection, cfg_options,
                                output_dir, "Detector's", save_features)
        self.start_of_feature_extraction_time_detection = t0
        if not is_train:
            return out
        if save_features:
            return None
        neg, pos, COXY = out["detector"]
        if extract_features_segmentation:
            return neg, pos, COXY, out["mask"][0], out["mask"][1]
        return neg, pos, COXY

    def extractRPNFeatures(self, is_train, output_dir=None, save_features=False, cfg_options={}):
        """-> RPN negatives, positives, COXY  (feature_extractor_RPN/extract_features_RPN.py:218)."""
        t0, out = self._harvest(self.cfg_path_RPN, ("rpn",), True, True, cfg_options, output_dir, "RPN's", save_features)
        self.start_of_feature_extraction_time_RPN = t0
        return None if save_features else out["rpn"]

    def extractFeaturesRPNDetector(self, is_train, output_dir=None, save_features=False, extract_features_segmentation=False,
                                   use_only_gt_positives_detection=True, cfg_options={}):
        """One pass -> (rpn negatives, positives, COXY, detector negatives, positives, COXY[, segmentation negatives,
        positives])  (feature_extractor_RPN_detector/extract_features_rpn_detector.py:354-364); test time -> test_boxes."""
        parts = ("rpn", "detector", "mask") if extract_features_segmentation else ("rpn", "detector")
        t0, out = self._harvest(self.cfg_path_target_task, parts, is_train, use_only_gt_positives_detection, cfg_options,
                                output_dir, "RPN and detector's", save_features)
        self.start_of_feature_extraction_time = t0
        self.end_of_feature_extraction_time = time.time()
        if not is_train:
            return out
        if save_features:
            return None
        res = tuple(out["rpn"]) + tuple(out["detector"])
        if extract_features_segmentation:
            res = res + tuple(out["mask"])
        return res

    def trainFeatureExtractor(self, *args, **kwargs):
        raise NotImplementedError("SGD training of the Mask R-CNN baselines is out of scope (SURVEY §2 row 21)")

    def testFeatureExtractor(self, *args, **kwargs):
        raise NotImplementedError("the Mask R-CNN baseline tester is out of scope (SURVEY §2 row 21)")
