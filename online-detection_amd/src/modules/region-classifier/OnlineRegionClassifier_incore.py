"""Drop-in for region-classifier/OnlineRegionClassifier_incore.py (device tensors)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, os.path.pardir, os.path.pardir)))
sys.path.insert(0, _HERE)
import _odx_path  # noqa: F401,E402
import RegionClassifierAbstract as rcA  # noqa: E402
from odx.region_classifier import OnlineRegionClassifierBase  # noqa: E402


class OnlineRegionClassifier(OnlineRegionClassifierBase, rcA.RegionClassifierAbstract):
    incore = True
