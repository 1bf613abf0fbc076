#!/usr/bin/env python3
"""Host-side profile (cProfile) of the reference-regime Minibootstrap in one of its modes; development aid.
    python tools/minibootstrap_profile.py [default|sequential|class_streams|class_batch] [k]"""
import cProfile
import io
import os
import pstats
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

from tools import bench_extras  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "class_batch"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
opts = {"reference_order": "sequential"} if mode == "sequential" else (None if mode == "default" else {mode: k})
bench_extras.minibootstrap_extra(modes=((mode, opts),))          # warm
pr = cProfile.Profile()
pr.enable()
out = bench_extras.minibootstrap_extra(modes=((mode, opts),))
pr.disable()
print(out)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
