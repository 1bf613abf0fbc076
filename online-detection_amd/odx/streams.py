"""Side streams that do not share a hardware queue.

A HIP stream is a software queue; the runtime multiplexes all streams of a process onto a few hardware queues (4 by default,
`GPU_MAX_HW_QUEUES`), and work of two streams that landed on the same hardware queue runs in submission order: a
latency-bound factorisation chain queues behind every kernel of a K_nM build that happens to share its queue.  Which streams
share one depends on the order in which all streams of the process were first used — a Minibootstrap round measured 0.47 s or
0.60 s on the same machine depending on how many streams the process had touched before.

`distinct(n)` picks side streams by MEASURING: a candidate is kept if a marker recorded on it completes while a busy kernel is
still running on the current stream and on every stream kept so far.  At most (hardware queues - 1) streams can qualify; the
callers spread their work over what they get.  The result is cached per (device, current stream): ~50 ms once.
"""
import time

import torch

_CACHE = {}
CANDIDATES = 12
BUSY_CYCLES = 4_000_000      # ~2 ms of torch.cuda._sleep: the marker is looked at after 0.3 ms, a host hiccup of 1.5 ms changes nothing
WAIT_S = 3e-4


def _collide(a, b, scratch):
    """True when a marker on stream b does not complete while stream a is busy."""
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(BUSY_CYCLES)
    done = torch.cuda.Event()
    with torch.cuda.stream(b):
        scratch.add_(1)
        done.record()
    time.sleep(WAIT_S)
    hit = not done.query()
    torch.cuda.synchronize()
    return hit


def distinct(n, device=None):
    """Up to n side streams on hardware queues of their own (not the current stream's, not each other's), in a fixed order
    per (device, current stream).  Fewer than n when the device's queues run out; [] without a GPU."""
    if not torch.cuda.is_available():
        return []
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    cur = torch.cuda.current_stream(dev)
    key = (dev, cur.cuda_stream)
    have = _CACHE.get(key)
    if not hasattr(torch.cuda, "_sleep"):
        return []                  # no busy kernel to measure with: the callers fall back to plain streams
    if have is None or (len(have[0]) < n and not have[1]):
        with torch.cuda.device(dev):
            scratch = torch.zeros(8, device="cuda")
            chosen = list(have[0]) if have else []
            exhausted = False
            tried = 0
            while len(chosen) < n and tried < CANDIDATES:
                c = torch.cuda.Stream()
                tried += 1
                with torch.cuda.stream(c):
                    scratch.add_(1)              # first use: the runtime assigns the hardware queue now
                if not any(_collide(r, c, scratch) for r in [cur] + chosen):
                    chosen.append(c)
            exhausted = len(chosen) < n
        have = _CACHE[key] = (chosen, exhausted)
    return list(have[0][:n])


def spread(streams, k):
    """k streams for round-robin use out of `streams` (repeated when there are fewer); [] when there are none."""
    return [streams[i % len(streams)] for i in range(k)] if streams else []
