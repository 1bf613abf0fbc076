#!/usr/bin/env python3
"""A/B of the MFMA shape in the 256 x 256 LDS-DMA tile loop (round-4 review, item 3 iii): the same packed operands through
odx_gemm_h2_f32 (v_mfma_f32_16x16x32_f16, 24 MFMAs per part) and odx_debug_gemm_h2_mf32 (v_mfma_f32_32x32x16_f16, 12 per part,
each twice as long; same fragment reads, same registers).  First a check that both give the product, then the rate at the
Gaussian kernels' depth (K = 1024) and at a depth where the epilogue is nothing (K = 4096).  Development aid (GPU box)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx import hip  # noqa: E402

be = odx.get_backend()
_p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731


def mf32(pa, pb):
    out = torch.empty((pa.n, pb.n), dtype=torch.float32, device="cuda")
    hip.check(be.lib.odx_debug_gemm_h2_mf32(_p(pa.P), pa.P.stride(0), _p(pa.meta), pa.n, _p(pb.P), pb.P.stride(0), _p(pb.meta), pb.n, pa.D,
                                            _p(out), pb.n, be._stream()), "odx_debug_gemm_h2_mf32")
    return out


g = torch.Generator().manual_seed(0)
A, B = torch.randn((70000, 200), generator=g).cuda(), torch.randn((300, 200), generator=g).cuda()
pa, pb = be.packed(A), be.packed(B)
ref = A.double() @ B.double().t()
for name, got in (("16x16x32", be.gemm_h2(pa, pb)), ("32x32x16", mf32(pa, pb))):
    print("%s: max |got - f64 product| / max |product| = %.2e" % (name, float((got.double() - ref).abs().max() / ref.abs().max())), flush=True)

for m, n, K in ((131072, 8192, 1024), (65536, 8192, 4096)):
    A, B = torch.randn((m, K), generator=g).cuda(), torch.randn((n, K), generator=g).cuda()
    pa, pb = be.packed(A), be.packed(B)
    del A, B
    res = {}
    for rep in range(2):                       # alternate the two, two rounds: the chip's clock drifts with what ran before
        for name, fn in (("16x16x32", lambda: be.gemm_h2(pa, pb)), ("32x32x16", lambda: mf32(pa, pb))):
            fn()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(5):
                fn()
            ev[1].record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(ev[0].elapsed_time(ev[1]) / 5)
    flop = 2.0 * m * n * K
    print("m = %d, n = %d, K = %d:" % (m, n, K), ", ".join("%s %.2f / %.2f ms = %.0f TF algorithmic" % (k, v[0], v[1], flop / min(v) / 1e9)
                                                         for k, v in res.items()), flush=True)
