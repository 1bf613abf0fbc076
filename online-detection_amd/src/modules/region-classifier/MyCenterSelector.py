"""Drop-in for region-classifier/MyCenterSelector.py: Nystroem centres = pre-chosen rows."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir, os.path.pardir)))
import _odx_path  # noqa: F401,E402
from odx.wrappers import CenterSelector as MyCenterSelector  # noqa: F401,E402
