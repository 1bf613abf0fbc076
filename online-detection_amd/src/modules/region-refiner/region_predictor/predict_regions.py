"""Drop-in for region_predictor/predict_regions.py: RegionPredictor (odx/rls.py)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *([os.path.pardir] * 3))))
import _odx_path  # noqa: F401,E402
from odx.rls import RegionPredictor  # noqa: F401,E402
