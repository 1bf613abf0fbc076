#!/usr/bin/env python3
"""The reference-regime Minibootstrap of tools/bench_extras.py in its default mode (after a warm-up repetition), for
`rocprofv3 --kernel-trace --stats -- python tools/prof_minibootstrap.py`; tools/trace_window.py then shows where the time
goes on the GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import odx  # noqa: E402
from bench_extras import minibootstrap_extra  # noqa: E402

odx.get_backend()
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
modes = {"default": (("default", None),), "class_batch4": (("class_batch4", {"class_batch": 4}),)}[mode]
print(minibootstrap_extra(modes=modes))
