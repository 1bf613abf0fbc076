"""Drop-in for region-classifier/FALKONWrapper_with_centers_selection_incore.py (GPU-resident
FALKON): class FALKONWrapper with train / predict / test / compute_indices_selection."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, os.path.pardir, os.path.pardir)))
sys.path.insert(0, _HERE)
import _odx_path  # noqa: F401,E402
import ClassifierAbstract as ca  # noqa: E402
from odx.wrappers import FALKONWrapperBase  # noqa: E402


class FALKONWrapper(FALKONWrapperBase, ca.ClassifierAbstract):
    incore = True
