#!/usr/bin/env python3
"""The class-batched Minibootstrap of tools/bench_extras.py once (after a warm-up run), for rocprofv3 --kernel-trace;
tools/round_timeline.py then shows where one round's time goes on the GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import odx  # noqa: E402
from bench_extras import minibootstrap_extra  # noqa: E402

odx.get_backend()
print(minibootstrap_extra(modes=(("class_batch4", {"class_batch": 4}),)))
