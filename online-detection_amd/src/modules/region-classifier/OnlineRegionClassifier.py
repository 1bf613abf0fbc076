"""Drop-in for region-classifier/OnlineRegionClassifier.py (host tensors, `--CPU`)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, os.path.pardir, os.path.pardir)))
sys.path.insert(0, _HERE)
import _odx_path  # noqa: F401,E402
import RegionClassifierAbstract as rcA  # noqa: E402
from odx.region_classifier import OnlineRegionClassifierBase  # noqa: E402


class OnlineRegionClassifier(OnlineRegionClassifierBase, rcA.RegionClassifierAbstract):
    incore = False

    def __init__(self, classifier, positives, negatives, stats, cfg_path=None, is_rpn=False, is_segmentation=False):
        # stats is mandatory in this variant (OnlineRegionClassifier.py:21,54-57)
        super().__init__(classifier, positives, negatives, stats, cfg_path, is_rpn, is_segmentation)
        self.stats = stats
        self.mean, self.std, self.mean_norm = stats['mean'], stats['std'], stats['mean_norm']
