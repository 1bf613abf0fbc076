"""Drop-in for src/modules/accuracy-evaluator/OnlineDetectionPostProcessor.py (reference :11-79):
decode + clip + score threshold + per-class NMS on the MI355X + top-k."""
import os
import sys

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir, os.pardir)))
import _odx_path  # noqa: F401,E402
from odx.postprocess import OnlineDetectionPostProcessor  # noqa: F401,E402
