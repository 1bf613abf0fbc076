#!/usr/bin/env python3
"""Do an HBM-bound CG pass (knm_pass) and an MFMA-bound Gaussian-kernel build share the chip when issued on two
streams?  Prints alone / together times (development aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402

import odx  # noqa: E402


def main():
    n, M, D = int(os.environ.get("N", 500000)), 10000, 1024
    be = odx.get_backend()
    X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
    F = be.features(X)
    Zf = be.rows(be.pack(F), torch.arange(M, device="cuda"))
    K1 = be.knm(F, Zf, 15.0)
    buf2 = torch.empty(n * K1.ld, device="cuda")
    v = torch.randn(M, dtype=torch.float64, device="cuda")
    out = torch.empty(M, dtype=torch.float64, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    passes = 20

    def cg():
        with torch.cuda.stream(sa):
            for _ in range(passes):
                be.ktk(K1, v=v, out=out)

    def build():
        with torch.cuda.stream(sb):
            be.knm(F, Zf, 15.0, out=buf2)
            be.knm(F, Zf, 15.0, out=buf2)

    def run(fns):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        sa.wait_event(a), sb.wait_event(a)
        for f in fns:
            f()
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(sa), eb.record(sb)
        torch.cuda.current_stream().wait_event(ea), torch.cuda.current_stream().wait_event(eb)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    for f in (cg, build):
        f()
    t_cg = min(run([cg]) for _ in range(2))
    t_b = min(run([build]) for _ in range(2))
    t_both = min(run([cg, build]) for _ in range(2))
    t_both2 = min(run([build, cg]) for _ in range(2))
    print("n=%d: %d passes alone %.1f ms | 2 knm builds alone %.1f ms | together %.1f ms (cg first) %.1f ms (build first) | sum %.1f"
          % (n, passes, t_cg, t_b, t_both, t_both2, t_cg + t_b))


if __name__ == "__main__":
    main()
