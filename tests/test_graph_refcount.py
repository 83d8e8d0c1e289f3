"""Host logic of unipose_amd/graph.py that needs no GPU: the per-stream-handle user count behind close()."""
from unipose_amd import graph


def test_stream_scratch_is_released_by_the_last_user_only():
    h = (0, 0xABC0)
    graph._acquire_stream(h)
    graph._acquire_stream(h)
    graph._acquire_stream((0, 0xDEF0))
    assert graph._release_stream(h) is False          # a twin is still live on this handle
    assert graph._release_stream((0, 0xDEF0)) is True
    assert graph._release_stream(h) is True
    assert h not in graph._stream_users
