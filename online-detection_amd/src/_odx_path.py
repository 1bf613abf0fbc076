"""Puts the odx package (online-detection_amd/) on sys.path for the drop-in modules below
src/, which the reference's experiment scripts import by bare name after their own
sys.path.append calls (experiments/run_experiment_online_rpn_ood_oos.py:6-18)."""
import os
import sys

_PKG = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.pardir))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
