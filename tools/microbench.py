#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (development aid; numbers quoted in DESIGN.md come from
bench.py + rocprofv3, not from here)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402
from odx import hip  # noqa: E402


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def timeit(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


def main():
    be = odx.get_backend()
    lib = be.lib
    which = sys.argv[1:] or ["gemm", "precond", "pass", "gauss", "rls"]
    if "gemm" in which:
        for n in (2048, 4096, 8192):
            for dt, fn, name in ((torch.float64, lib.odx_gemm_nt_f64, "f64"), (torch.float32, lib.odx_gemm_nt_f32, "f32")):
                A = torch.randn(n, n, dtype=dt, device="cuda")
                B = torch.randn(n, n, dtype=dt, device="cuda")
                C = torch.zeros(n, n, dtype=dt, device="cuda")
                ms = timeit(lambda: hip.check(fn(_p(A), n, _p(B), n, _p(C), n, n, n, n, 1.0, 0.0, 0, be._stream())))
                print("gemm_nt_%s n=%d: %.3f ms  %.1f TFLOP/s" % (name, n, ms, 2.0 * n ** 3 / ms / 1e9))
        # potrf-shaped update: k = 128
        for m in (4096, 10000):
            A = torch.randn(m, 128, dtype=torch.float64, device="cuda")
            C = torch.zeros(m, m, dtype=torch.float64, device="cuda")
            ms = timeit(lambda: hip.check(lib.odx_gemm_nt_f64(_p(A), 128, _p(A), 128, _p(C), m, m, m, 128, -1.0, 1.0,
                                                                hip.GEMM_LOWER_ONLY, be._stream())))
            print("syrk-shaped f64 m=%d k=128: %.3f ms  %.1f TFLOP/s (lower only)" % (m, ms, m * m * 128.0 / ms / 1e9))
    if "precond" in which:
        for M, D in ((2000, 1024), (10000, 1024)):
            Z = torch.randn(M, D, device="cuda") * (20.0 / D ** 0.5)
            Zf = be.features(Z)
            ms = timeit(lambda: be.precond(Zf, 15.0, 1e-5, 1e-5), reps=2)
            print("precond M=%d D=%d: %.2f ms" % (M, D, ms))
            G = torch.randn(M, M + 8, dtype=torch.float64, device="cuda")
            A0 = (G @ G.T) / M + torch.eye(M, dtype=torch.float64, device="cuda")
            ld = M
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            ws = torch.empty(lib.odx_potrf_workspace_bytes(M), dtype=torch.uint8, device="cuda")
            A = A0.clone()

            def potrf():
                A.copy_(A0)
                hip.check(lib.odx_potrf_f64(_p(A), ld, M, _p(info), _p(ws), ws.numel(), be._stream()))
            ms_copy = timeit(lambda: A.copy_(A0))
            ms = timeit(potrf, reps=2) - ms_copy
            print("  potrf M=%d: %.2f ms (%.1f TFLOP/s)" % (M, ms, M ** 3 / 3.0 / ms / 1e9))
            Li = torch.zeros(M, ld, dtype=torch.float64, device="cuda")
            Lit = torch.zeros(M, ld, dtype=torch.float64, device="cuda")
            ws2 = torch.empty(lib.odx_trtri_workspace_bytes(M), dtype=torch.uint8, device="cuda")
            ms = timeit(lambda: hip.check(lib.odx_trtri_f64(_p(A), ld, M, _p(Li), _p(Lit), ld, _p(ws2), ws2.numel(),
                                                             be._stream())), reps=2)
            print("  trtri M=%d: %.2f ms (%.1f TFLOP/s)" % (M, ms, M ** 3 / 3.0 / ms / 1e9))
            x = torch.randn(M, dtype=torch.float64, device="cuda")
            y = torch.empty(M, dtype=torch.float64, device="cuda")
            ms = timeit(lambda: hip.check(lib.odx_trmv_f64(_p(Li), ld, M, 0, _p(x), 1.0, 0.0, None, _p(y), be._stream())))
            print("  trmv M=%d: %.3f ms (%.0f GB/s)" % (M, ms, M * M * 4.0 / ms / 1e6))
    if "pass" in which:
        from odx.backend import Knm
        for n, M in ((100000, 2000), (250000, 10000), (1000000, 10000), (100000, 20000)):
            ld = (M + 3) // 4 * 4
            K = Knm()
            K.K = torch.rand(n, ld, device="cuda")
            K.n, K.M, K.ld = n, M, ld
            v = torch.randn(M, dtype=torch.float64, device="cuda")
            ms = timeit(lambda: be.ktk(K, v=v))
            print("knm_fwd_bwd n=%d M=%d: %.3f ms  %.0f GB/s" % (n, M, ms, n * M * 4.0 / ms / 1e6))
            if be.can_ktk2(K):
                v2 = torch.randn(M, dtype=torch.float64, device="cuda")
                ms = timeit(lambda: be.ktk2(K, v, v2))
                print("knm_fwd_bwd2 (two vectors, one read) n=%d M=%d: %.3f ms  %.0f GB/s" % (n, M, ms, n * M * 4.0 / ms / 1e6))
            del K
    if "rls" in which:
        # per-class RLS box regressor (A7): f64 Gram of the class's rows + 4 right-hand sides, Cholesky solve, predictions.
        # cfg 3 of BASELINE.json: COXY n = 3e5 rows, D = 1024, 30 classes => 1e4 rows per class.
        for n, D, nc in ((300000, 1024, 10000), (60000, 2048, 2000)):
            X = torch.randn(n, D, device="cuda")
            F = be.features(X)
            I = torch.arange(0, nc, device="cuda") * (n // nc)
            D1 = D + 1
            ldg = (D1 + 1) // 2 * 2
            Yt = torch.randn((4, (nc + 15) // 16 * 16 + 16), dtype=torch.float64, device="cuda")
            G = torch.zeros((D1, ldg), dtype=torch.float64, device="cuda")
            XtY = torch.zeros((4, ldg), dtype=torch.float64, device="cuda")
            ms_g = timeit(lambda: be.rls_gram(F, I, Yt, G, XtY))
            be.rls_gram(F, I, Yt, G, XtY)
            ms_s = timeit(lambda: be.rls_solve(G, D, 1000.0, XtY))
            W, _ = be.rls_solve(G, D, 1000.0, XtY)
            ms_p = timeit(lambda: be.rls_predict_rows(F, I, W))
            print("rls D=%d, %d rows of the class: gram %.3f ms (%.1f TFLOP/s f64, lower triangle: nc (D+1)^2 flop) | solve %.3f ms | "
                  "predict %.3f ms | per class %.3f ms" % (D, nc, ms_g, nc * D1 * D1 / ms_g / 1e9, ms_s, ms_p, ms_g + ms_s + ms_p))
    if "gauss" in which:
        for n, M, D in ((100000, 2000, 1024), (250000, 10000, 1024), (100000, 2000, 256), (100000, 2000, 2048)):
            X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
            Z = X[:M].clone()
            F, Zf = be.features(X), be.features(Z)
            buf = torch.empty(n * ((M + 3) // 4 * 4), device="cuda")
            al = torch.randn(M, dtype=torch.float64, device="cuda")
            out = torch.empty(n, 1, device="cuda")
            ms = timeit(lambda: hip.check(lib.odx_split_f16(_p(F.X), F.ld, n, D, _p(buf), (D + 63) // 64 * 64, _p(out), be._stream())))
            print("split_f16 n=%d D=%d: %.3f ms  %.0f GB/s (read + write)" % (n, D, ms, n * D * 8.0 / ms / 1e6))
            for mode in ("f32", "h2"):
                be.gauss = mode
                ms = timeit(lambda: be.knm(F, Zf, 15.0, out=buf))
                print("gauss_knm[%s] n=%d M=%d D=%d: %.3f ms  %.1f TFLOP/s" % (mode, n, M, D, ms, 2.0 * n * M * D / ms / 1e9))
                ms = timeit(lambda: be.mmv(F, Zf, 15.0, al, None, out=out))
                print("gauss_mmv[%s] n=%d M=%d D=%d: %.3f ms  %.1f TFLOP/s" % (mode, n, M, D, ms, 2.0 * n * M * D / ms / 1e9))


if __name__ == "__main__":
    main()
