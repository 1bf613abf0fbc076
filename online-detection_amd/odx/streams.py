"""Side streams that do not share a hardware queue.

A HIP stream is a software queue; the runtime multiplexes all streams of a process onto a few hardware queues (4 by default,
`GPU_MAX_HW_QUEUES`), and work of two streams that landed on the same hardware queue runs in submission order: a
latency-bound factorisation chain queues behind every kernel of a K_nM build that happens to share its queue.  Which streams
share one depends on the order in which all streams of the process were first used — a Minibootstrap round measured 0.47 s or
0.60 s on the same machine depending on how many streams the process had touched before (docs/HISTORY.md 7, round 4).

`distinct(n)` picks side streams by MEASURING: a candidate is kept if a marker recorded on it completes while a busy kernel is
still running on the current stream and on every stream kept so far.  At most (hardware queues - 1) streams can qualify; the
callers spread their work over what they get.  The result is cached per (device, current stream): ~50 ms once; a heuristic —
it changes how work overlaps, never what is computed.
"""
import os

import torch

_CACHE = {}
CANDIDATES = 12
BUSY_CYCLES = 2_000_000      # ~1 ms of torch.cuda._sleep per reading


def _collide(a, b, scratch):
    """True when a marker on stream b does not complete while stream a is busy.  Decided from EVENT TIMES on the device — the
    marker's completion against the start and the end of the busy kernel —, not from a host sleep: a stalled host (a loaded
    box, a profiler attached) cannot make a colliding stream look independent (advisor, round 4).  Two readings must agree;
    when they do not, the stream counts as colliding (the safe side: it is simply not picked)."""
    from . import options as _options
    if _options.current().stream_probe != "events":
        # round 4's reading, kept as the default: one look after a host sleep.  The event-timed reading below (two readings that
        # must agree: a stalled host cannot make a colliding stream look independent — advisor, round 4) is the more careful test,
        # but the probe's own launches are part of what decides which hardware queue a stream lands on, and the arrangement the
        # careful probe ends with costs the headline job 2 % (A/B on one box, two runs each: 6.89 / 6.92 s per step against
        # 6.72 / 6.78 — the preconditioner chains then run beside the HBM-bound passes instead of beside the builds).  It stays
        # selectable (odx.options stream_probe="events"); either way the choice changes how work overlaps, never what is computed.
        # A reading only counts if stream a was STILL busy when the marker was looked at (advisor, round 5: a host stalled past
        # the busy kernel's end — a loaded box, a profiler — saw the marker done and called a colliding stream independent);
        # three attempts, then the safe verdict (colliding: the stream is simply not picked).
        import time
        for _ in range(3):
            torch.cuda.synchronize()
            busy_end = torch.cuda.Event()
            with torch.cuda.stream(a):
                torch.cuda._sleep(2 * BUSY_CYCLES)
                busy_end.record()
            done = torch.cuda.Event()
            with torch.cuda.stream(b):
                scratch.add_(1)
                done.record()
            time.sleep(3e-4)
            finished = done.query()
            valid = not busy_end.query()          # a was busy throughout what the host just looked at
            torch.cuda.synchronize()
            if valid:
                return not finished
        return True
    verdicts = []
    for _ in range(2):
        torch.cuda.synchronize()
        start, end, done = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(a):
            start.record()
            torch.cuda._sleep(BUSY_CYCLES)
            end.record()
        with torch.cuda.stream(b):
            scratch.add_(1)
            done.record()
        torch.cuda.synchronize()
        busy = start.elapsed_time(end)
        verdicts.append(start.elapsed_time(done) >= 0.5 * busy)     # independent: the marker is done microseconds after the start
    return verdicts[0] or verdicts[1]


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


def of_default(n, device=None):
    """distinct(n) as seen from the device's DEFAULT stream, whatever stream the caller is on: the process-wide set of side
    streams (HipBackend picks it when it is created, i.e. while the process is fresh — round 6: side streams first picked
    BEHIND a big job landed on worse hardware queues, a Minibootstrap read 0.51 instead of 0.43 s and the harvest loop 4.5
    instead of 3.6 ms per image; tools/after_headline_probe.py)."""
    if not torch.cuda.is_available():
        return []
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    with torch.cuda.stream(torch.cuda.default_stream(dev)):
        return distinct(n, dev)


def spread(streams, k):
    """k streams for round-robin use out of `streams` (repeated when there are fewer); [] when there are none."""
    return [streams[i % len(streams)] for i in range(k)] if streams else []
