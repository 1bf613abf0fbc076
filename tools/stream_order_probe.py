#!/usr/bin/env python3
"""The reference-regime Minibootstrap (or detect()) after K streams have been created and used by the process: how much the
result depends on the stream history (before odx/streams.py and the helper-stream rule: 0.47-0.60 s over K = 0..5; after:
0.47-0.49 s).  Usage: python tools/stream_order_probe.py K [mb|detect]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "online-detection_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import odx
import bench_extras as bx
odx.get_backend()
k = int(sys.argv[1])
keep = []
for _ in range(k):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.zeros(8, device="cuda").add_(1)
    keep.append(s)
torch.cuda.synchronize()
what = sys.argv[2] if len(sys.argv) > 2 else "mb"
if what == "mb":
    print("shift", k, bx.minibootstrap_extra(modes=(("default", None),))["s_default"])
else:
    print("shift", k, bx.detect_extra()["ms_per_image"])
