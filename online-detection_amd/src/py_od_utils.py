"""Drop-in for the reference's src/py_od_utils.py: same function names and behaviour, implemented
in odx/utils.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _odx_path  # noqa: F401,E402
from odx.utils import (computeFeatStatistics, computeFeatStatistics_torch, decode_boxes_detector, falkon_models_to_cuda,  # noqa: F401,E402
                       load_features_classifier, load_features_regressor, load_positives_from_COXY, mask_iou,
                       minibatch_positives, normalize_COXY, shuffle_negatives, zScores)
