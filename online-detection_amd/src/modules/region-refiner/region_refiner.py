"""Drop-in for region-refiner/region_refiner.py: class RegionRefiner (RLS box regressors)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, os.path.pardir, os.path.pardir)))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, os.path.pardir)))
sys.path.insert(0, _HERE)
import _odx_path  # noqa: F401,E402
import yaml  # noqa: E402
from RegionRefinerAbstract import RegionRefinerAbstract  # noqa: E402
from region_predictor import RegionPredictor  # noqa: E402
from region_refiner_trainer import RegionRefinerTrainer  # noqa: E402


class RegionRefiner(RegionRefinerAbstract):
    """region_refiner.py:8-36 of the reference: YAML `REGION_REFINER.opts.lambda`, `RPN:` subtree
    when is_rpn; trainRegionRefiner(COXY) -> array of per-class model dicts; predict(...)."""

    def __init__(self, cfg_path_region_refiner, is_rpn=False):
        with open(cfg_path_region_refiner) as fid:
            self.cfg = yaml.load(fid, Loader=yaml.FullLoader)
        if is_rpn:
            self.cfg = self.cfg['RPN']
        try:
            self.lambd = self.cfg['REGION_REFINER']['opts']['lambda']
        except Exception:
            self.lambd = None
        self.is_rpn = is_rpn

    def loadRegionRefiner(self):
        return

    def trainRegionRefiner(self, COXY, output_dir=None):
        trainer = RegionRefinerTrainer(self.cfg, lmbd=self.cfg['REGION_REFINER']['opts']['lambda'], is_rpn=self.is_rpn)
        self.models = trainer(COXY, output_dir=output_dir)
        return self.models

    def testRegionRefiner(self):
        return

    def predict(self, boxes, features, models=None, normalize_features=False, stats=None):
        predictor = RegionPredictor(self.cfg, self.models if models is None else models)
        return predictor(boxes, features, normalize_features=normalize_features, stats=stats)
