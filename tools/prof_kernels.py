#!/usr/bin/env python3
"""One launch of each hot kernel at the headline shard shape, for rocprofv3 --pmc passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
n, M, D = int(os.environ.get("ODX_N", 250000)), 10000, 1024
X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
Z = X[:M].clone()
F, Zf = be.features(X), be.features(Z)
for _ in range(2):
    K = be.knm(F, Zf, 15.0)
    v = torch.randn(M, dtype=torch.float64, device="cuda")
    be.ktk(K, v=v)
    if be.can_ktk2(K):
        be.ktk2(K, v, torch.randn(M, dtype=torch.float64, device="cuda"))     # the two-vector pass of the folded full residual
    al = torch.randn(M, dtype=torch.float64, device="cuda")
    be.mmv(F, Zf, 15.0, al)
torch.cuda.synchronize()
