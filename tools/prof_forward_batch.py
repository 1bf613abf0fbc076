#!/usr/bin/env python3
"""What to run under `rocprofv3 --kernel-trace --stats`: the R-50-C4 forward of ODX_FWD_B (default 4) 600 x 800 images per
call (extract.forward_batch), ODX_FWD_DTYPE = f32 | bf16, 12 calls after the warm-up."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel, forward_batch  # noqa: E402

odx.get_backend()
B = int(os.environ.get("ODX_FWD_B", "4"))
dt = torch.bfloat16 if os.environ.get("ODX_FWD_DTYPE", "f32") == "bf16" else None
model = OnlineDetectionModel(compute_dtype=dt).cuda().eval()
x = torch.randn((B, 3, 600, 800), generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    for _ in range(15):
        forward_batch(model, x) if B > 1 else model(x)
torch.cuda.synchronize()
