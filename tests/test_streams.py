"""odx/streams.py: side streams measured to sit on hardware queues of their own."""
import pytest

torch = pytest.importorskip("torch")


def test_no_gpu_no_streams():
    from odx import streams
    if not torch.cuda.is_available():
        assert streams.distinct(3) == []
    assert streams.spread([], 4) == []
    assert streams.spread(["a", "b"], 5) == ["a", "b", "a", "b", "a"]


@pytest.mark.gpu
def test_distinct_streams_do_not_wait_for_each_other():
    """Whatever distinct() returns, a marker on one of them completes while the current stream and the others are busy — and
    asking again returns the same streams (the choice is made once per calling stream)."""
    import time
    from odx import streams
    own = streams.distinct(3)
    assert 1 <= len(own) <= 3 and len({s.cuda_stream for s in own}) == len(own)
    assert [s.cuda_stream for s in streams.distinct(3)] == [s.cuda_stream for s in own]
    assert [s.cuda_stream for s in streams.distinct(2)] == [s.cuda_stream for s in own[:2]]
    x = torch.zeros(8, device="cuda")
    cur = torch.cuda.current_stream()
    for k, s in enumerate(own):
        torch.cuda.synchronize()
        for busy in [cur] + [t for j, t in enumerate(own) if j != k]:
            with torch.cuda.stream(busy):
                torch.cuda._sleep(20_000_000)            # ~10 ms
        done = torch.cuda.Event()
        with torch.cuda.stream(s):
            x.add_(1)
            done.record()
        time.sleep(0.002)
        assert done.query(), "stream %d waited for another stream's kernel" % k
    torch.cuda.synchronize()
