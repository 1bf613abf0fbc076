"""Drop-in for region_refiner_trainer/train_region_refiner.py: RegionRefinerTrainer (odx/rls.py)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *([os.path.pardir] * 3))))
import _odx_path  # noqa: F401,E402
from odx.rls import RegionRefinerTrainer  # noqa: F401,E402
