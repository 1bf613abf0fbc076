import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "online-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# The f32 trunk / RPN head run as row GEMMs on the library's tile cores from three 600 x 800 images per call on (the
# convolution library is faster below that: OnlineDetectionModel.rows_min_positions).  The tests' reduced models and small
# images would never get there: they take that route at every size; the convolution route is asked for by name
# (test_forward_gpu_equals_plain_torch_cpu[conv]).
os.environ.setdefault("ODX_ROWS_MIN_POSITIONS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
