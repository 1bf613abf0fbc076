#!/usr/bin/env python3
"""Do idle streams left behind by earlier work slow later latency-bound work?  The forward numbers fresh, with 8 more idle
normal-priority streams, with one idle LOW-priority stream, with one idle HIGH-priority stream (each used once).  Development aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from tools import bench_extras as b  # noqa: E402

odx.get_backend()
keep = []


def show(tag):
    d = b.forward_extra()
    print(tag, {k: v for k, v in d.items() if k in ("ms_per_image_f32", "ms_per_image_f32_group4", "ms_per_image_f32_group8", "ms_per_image_bf16_group4")}, flush=True)


def use(s):
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
    keep.append(s)


show("fresh:")
for _ in range(8):
    use(torch.cuda.Stream())
show("+ 8 idle normal-priority streams:")
lo, hi = 0, -1
use(torch.cuda.Stream(priority=lo))
show("+ 1 idle stream of priority 0 (torch's default):")
try:
    import ctypes
    hipl = ctypes.CDLL("libamdhip64.so")
    a, c = ctypes.c_int(), ctypes.c_int()
    hipl.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(c))
    print("priority range (least, greatest):", a.value, c.value)
    raw = ctypes.c_void_p()
    hipl.hipStreamCreateWithPriority(ctypes.byref(raw), 1, a.value)          # hipStreamNonBlocking = 1, the LOWEST priority
    use(torch.cuda.ExternalStream(raw.value))
    show("+ 1 idle stream of the LOWEST priority:")
    raw2 = ctypes.c_void_p()
    hipl.hipStreamCreateWithPriority(ctypes.byref(raw2), 1, c.value)
    use(torch.cuda.ExternalStream(raw2.value))
    show("+ 1 idle stream of the HIGHEST priority:")
except Exception as e:      # noqa: BLE001
    print("priority streams:", e)
