#!/usr/bin/env python3
"""Which HIP streams of this process share a hardware queue: N torch streams + the default stream, every pair probed (a marker
on stream b while a busy kernel runs on stream a: it completes only afterwards when the two share a queue).  Prints the
collision matrix.  `GPU_MAX_HW_QUEUES=8 python tools/hw_queue_probe.py 10` shows the effect of the runtime's queue count.
odx/streams.py picks its side streams with the same measurement."""
import sys, time, os
import torch
torch.cuda.init()
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.default_stream()] + [torch.cuda.Stream() for _ in range(n)]
x = torch.zeros(8, device=dev)
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
def collide(a, b):
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(4_000_000)       # ~2 ms
    e = torch.cuda.Event()
    with torch.cuda.stream(b):
        x.add_(1)
        e.record()
    time.sleep(0.0006)
    hit = not e.query()
    torch.cuda.synchronize()
    return hit
print("stream ids", [hex(s.cuda_stream) for s in streams])
for i, a in enumerate(streams):
    print(i, "".join("X" if (i != j and collide(a, b)) else "." for j, b in enumerate(streams)))
