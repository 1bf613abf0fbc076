#!/usr/bin/env python3
"""Three launches of each hot kernel of the headline job at its shard shape, as bench.py launches them (the build with the
fused right-hand side, K_nM in the default storage of that shape), for rocprofv3 --pmc passes.  ODX_N rows (default 250000)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402

be = odx.get_backend()
n, M, D = int(os.environ.get("ODX_N", 250000)), int(os.environ.get("ODX_M", 10000)), 1024
X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
Z = X[:M].clone()
F, Zf = be.features(X), be.features(Z)
w = torch.randn(n, dtype=torch.float64, device="cuda")
for _ in range(3):
    K, _ = be.knm_rhs(F, Zf, 15.0, w)
    v = torch.randn(M, dtype=torch.float64, device="cuda")
    be.ktk(K, v=v)
    if be.can_ktk2(K):
        be.ktk2(K, v, torch.randn(M, dtype=torch.float64, device="cuda"))     # the two-vector pass of the folded full residual
    al = torch.randn(M, dtype=torch.float64, device="cuda")
    be.mmv(F, Zf, 15.0, al)
    del K
torch.cuda.synchronize()
print("storage:", be.knm_format(n, M))
