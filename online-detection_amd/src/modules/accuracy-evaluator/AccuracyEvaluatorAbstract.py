"""Drop-in for the reference's AccuracyEvaluatorAbstract module: the API contract as an ABC (odx/contracts.py)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir, os.path.pardir)))
import _odx_path  # noqa: F401,E402
from odx.contracts import AccuracyEvaluatorAbstract  # noqa: F401,E402
