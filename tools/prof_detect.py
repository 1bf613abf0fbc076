#!/usr/bin/env python3
"""detect() (tools/bench_extras.py::detect_extra) a few times, for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import odx  # noqa: E402
from bench_extras import detect_extra  # noqa: E402

odx.get_backend()
print(detect_extra(reps=20))
