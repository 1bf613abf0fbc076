#!/usr/bin/env python3
"""Does it pay to run the HBM-bound CG passes of one class BESIDE the MFMA-bound Gaussian kernels of its neighbours, each on
its own part of the chip (CU-masked streams), instead of one after the other on all of it?

1. where do the bits of a CU mask land (XCC ids of a masked launch),
2. the compact pass alone on k CUs per XCD (bandwidth against CU count),
3. the K_nM build alone on the other 32 - k per XCD,
4. both at once on the disjoint partitions, against the sum of their times on the whole chip.

ODX_N rows (default 500000), ODX_M (10000), ODX_D (1024)."""
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
lib = be.lib
CUS = lib.odx_device_cus()
WORDS = (CUS + 31) // 32


def masked_stream(bits):
    arr = (ctypes.c_uint32 * WORDS)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    out = ctypes.c_void_p()
    hip.check(lib.odx_stream_create_cu_mask(arr, WORDS, ctypes.byref(out)), "odx_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value), out


def placement(stream, blocks):
    buf = torch.zeros(3 * blocks + 1, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        hip.check(lib.odx_debug_placement(buf.data_ptr(), blocks, 2000, be._stream()), "odx_debug_placement")
    stream.synchronize()
    v = buf[:3 * blocks].view(blocks, 3).cpu()
    xcc = v[:, 0].tolist()
    hw = v[:, 1].tolist()
    cu = sorted({(x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15) for x, h in zip(xcc, hw)})
    per = {}
    for x in xcc:
        per[x] = per.get(x, 0) + 1
    return per, len(cu)


print("device CUs:", CUS)
# ---- 1. mask bit -> XCC
round_robin = None
for name, bits in (("bits 0..31", range(0, 32)), ("bits 0..63", range(0, 64)), ("every 8th bit", range(0, CUS, 8)),
                   ("bits = 0 mod 2", range(0, CUS, 2))):
    st, raw = masked_stream(list(bits))
    per, ncu = placement(st, 512)
    if round_robin is None:
        round_robin = len(per) == 8        # 32 consecutive bits on all eight XCCs: bit i sits on XCC i % 8
    print("mask %-16s -> workgroups per XCC %s, distinct (xcc, se, sh, cu) seen: %d" % (name, dict(sorted(per.items())), ncu))
print("consecutive mask bits go %s" % ("round the XCCs (bit i on XCC i % 8)" if round_robin else "through one XCC after the other"))

n, M, D = int(os.environ.get("ODX_N", 500000)), int(os.environ.get("ODX_M", 10000)), int(os.environ.get("ODX_D", 1024))
X = torch.randn(n, D, device="cuda") * (20.0 / D ** 0.5)
Z = X[:M].clone()
F, Zf = be.features(X), be.features(Z)
w = torch.randn(n, dtype=torch.float64, device="cuda")
buf = torch.empty(be.knm_bytes(n, M), dtype=torch.uint8, device="cuda")
buf2 = torch.empty(be.knm_bytes(n, M), dtype=torch.uint8, device="cuda")
K = be.knm_rhs(F, Zf, 15.0, w, out=buf)[0]
v = torch.randn(M, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
NB, NP = 2, 16


def run(build_stream, pass_stream, builds, passes):
    """builds K_nM builds on one stream and `passes` passes on the other, started together; ms of each and of both."""
    e = {k: torch.cuda.Event(enable_timing=True) for k in ("b0", "b1", "p0", "p1")}
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    if builds:
        build_stream.wait_event(start)
        with torch.cuda.stream(build_stream):
            e["b0"].record()
            for _ in range(builds):
                be.knm_rhs(F, Zf, 15.0, w, out=buf2)
            e["b1"].record()
    if passes:
        pass_stream.wait_event(start)
        with torch.cuda.stream(pass_stream):
            e["p0"].record()
            for _ in range(passes):
                be.ktk(K, v=v)
            e["p1"].record()
    torch.cuda.synchronize()
    tb = e["b0"].elapsed_time(e["b1"]) / builds if builds else 0.0
    tp = e["p0"].elapsed_time(e["p1"]) / passes if passes else 0.0
    ends = [start.elapsed_time(e[k]) for k, c in (("b1", builds), ("p1", passes)) if c]
    return tb, tp, max(ends)


cur = torch.cuda.current_stream()
run(cur, cur, 1, 2)
tb0, _, _ = run(cur, cur, NB, 0)
_, tp0, _ = run(cur, cur, 0, NP)
gb = be.knm_bytes(n, M) / 1e9
print("whole chip: build %.2f ms, pass %.3f ms (%.0f GB/s); %d builds then %d passes: %.1f ms" % (tb0, tp0, gb / tp0 * 1e3, NB, NP, NB * tb0 + NP * tp0))

per_xcd = CUS // 8
for k in [int(a) for a in os.environ.get("ODX_SPLITS", "8,12,16,20").split(",")]:
    # k CUs of every XCC for the passes, the rest for the builds
    in_pass = (lambda i: i // 8 < k) if round_robin else (lambda i: i % per_xcd < k)
    pbits = [i for i in range(CUS) if in_pass(i)]
    bbits = [i for i in range(CUS) if not in_pass(i)]
    ps, _r1 = masked_stream(pbits)
    bs, _r2 = masked_stream(bbits)
    hip.check(lib.odx_set_pass_cus(len(pbits)), "odx_set_pass_cus")
    _, tp, _ = run(bs, ps, 0, NP)
    tb, _, _ = run(bs, ps, NB, 0)
    # both: as many passes as fit beside NB builds at the alone rates
    npass = max(1, int(round(NB * tb / tp)))
    tb2, tp2, both = run(bs, ps, NB, npass)
    seq = NB * tb0 + npass * tp0
    print("passes on %3d CUs (%2d per XCC), builds on %3d: pass alone %.3f ms (%.0f GB/s), build alone %.2f ms | together: build %.2f ms, "
          "pass %.3f ms (%.0f GB/s), %d builds + %d passes in %.1f ms against %.1f ms one after the other on the whole chip (x %.2f)"
          % (len(pbits), k, len(bbits), tp, gb / tp * 1e3, tb, tb2, tp2, gb / tp2 * 1e3, NB, npass, both, seq, seq / both))
    hip.check(lib.odx_set_pass_cus(0), "odx_set_pass_cus")
