#!/usr/bin/env python3
"""Why do the host-bound extras of the bench line run slower behind the headline job than in a fresh process?  One process: the
forward numbers fresh, behind (a) 200 GB allocated and released, (b) 60 s of chip-filling products, (c) both, and after 30 s of rest."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from tools import bench_extras as b  # noqa: E402

odx.get_backend()


def show(tag):
    d = b.forward_extra()
    print(tag, {k: v for k, v in d.items() if k in ("ms_per_image_f32", "ms_per_image_f32_group4", "ms_per_image_f32_group8", "ms_per_image_bf16_group4")}, flush=True)


show("fresh:")
x = torch.empty(200 * (1 << 30), dtype=torch.uint8, device="cuda")
x.zero_()
torch.cuda.synchronize()
del x
torch.cuda.empty_cache()
show("behind 200 GB allocated, touched and released:")
a = torch.randn(16384, 16384, device="cuda", dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 60:
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
del a
torch.cuda.empty_cache()
show("behind 60 s of products:")
time.sleep(30)
show("after 30 s of rest:")
