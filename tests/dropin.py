"""Imports the drop-in modules the way the reference's experiment drivers do: by bare module
name after appending src/ and src/modules/* to sys.path
(experiments/run_experiment_online_rpn_ood_oos.py:6-18)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "online-detection_amd", "src")
for sub in ("", "modules/region-classifier", "modules/region-refiner", "modules/feature-extractor", "modules/accuracy-evaluator", "modules"):
    p = os.path.join(SRC, sub) if sub else SRC
    if p not in sys.path:
        sys.path.append(p)


def load(name):
    return importlib.import_module(name)
