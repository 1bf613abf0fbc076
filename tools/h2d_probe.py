#!/usr/bin/env python3
"""Host cost of small host -> device copies while another stream keeps the GPU busy: pageable `.to(device)` against a persistent
page-locked staging buffer + non_blocking copy.  Development aid."""
import time
import torch

dev = torch.device("cuda")
a = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
stage = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
scratch = torch.zeros(1024, device=dev)


def busy(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            a @ a


def run(kind, with_busy, with_kernels):
    torch.cuda.synchronize()
    if with_busy:
        busy(40)
    t0 = time.perf_counter()
    off = 0
    for i in range(200):
        h = torch.arange(64, dtype=torch.int64) + i
        if with_kernels:
            scratch.add_(1)                 # a small kernel of the harvest in front of the copy
        if kind == "pageable":
            d = h.to(dev)
        else:
            n = h.numel() * 8
            s = stage[off:off + n].view(torch.int64)
            s.copy_(h)
            d = torch.empty(64, dtype=torch.int64, device=dev)
            d.copy_(s, non_blocking=True)
            off = (off + n) % (1 << 19)
        scratch[:64] += d
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / 200 * 1e6


for wb in (False, True):
    for wk in (False, True):
        for kind in ("pageable", "pinned"):
            print("GPU busy on another stream: %-5s small kernel before each copy: %-5s %-8s: %.1f us per copy (host)" % (wb, wk, kind, run(kind, wb, wk)), flush=True)
