#!/usr/bin/env python3
"""What to run under `rocprofv3 --kernel-trace --stats -- python3 tools/prof_forward_batch.py [B] [f32|bf16] [fpn]`: the R-50-C4
forward of B (default 4) 600 x 800 images per call (extract.forward_batch), 15 calls; the first ones carry the convolution
library's solver search (its naive reference kernels show up in the statistics: ignore `naive_conv_*`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx.extract import OnlineDetectionModel, forward_batch  # noqa: E402

odx.get_backend()
B = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("ODX_FWD_B", "4"))
dt = torch.bfloat16 if (sys.argv[2] if len(sys.argv) > 2 else os.environ.get("ODX_FWD_DTYPE", "f32")) == "bf16" else None
if len(sys.argv) > 3 and sys.argv[3] == "fpn":
    from odx.fpn import OnlineDetectionModelFPN
    model = OnlineDetectionModelFPN(compute_dtype=dt).cuda().eval()
else:
    model = OnlineDetectionModel(compute_dtype=dt).cuda().eval()
x = torch.randn((B, 3, 600, 800), generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    for _ in range(15):
        forward_batch(model, x) if B > 1 else model(x)
torch.cuda.synchronize()
