"""Engine's device-buffer pool bookkeeping (no GPU: the library is a recorder).  ADVICE round 3: buffers freed while
only queue 0 ran must not be handed to another queue without a synchronisation, and a synchronisation taken inside an
on_queue(k != 0) scope must not mark later frees idle."""
from trtools_amd.engine import Engine, _QueueScope


class _Lib:
    def __init__(self):
        self.calls = []

    def trk_sync(self, ctx):
        self.calls.append('sync')
        return 0

    def trk_stream_select(self, ctx, q):
        self.calls.append(('select', q))
        return 0


def _engine():
    e = Engine.__new__(Engine)
    e.lib, e.ctx = _Lib(), 1
    e._pool, e._pool_bytes, e._pool_limit = {}, 0, 1 << 30
    e._queue, e._multi_queue = 0, False
    return e


def test_single_queue_reuse_needs_no_sync():
    e = _engine()
    assert e._pool_give(4096, 111)
    assert e._pool_take(4096) == 111
    assert 'sync' not in e.lib.calls


def test_buffer_freed_on_queue_0_is_not_idle_for_another_queue():
    e = _engine()
    e._pool_give(4096, 111)                 # freed while only queue 0 ran: queue 0 may still be using it
    with _QueueScope(e, 1):
        assert e._pool_take(4096) == 111
        assert 'sync' in e.lib.calls        # the take on queue 1 waited for the device


def test_sync_inside_a_scope_does_not_clear_the_flag():
    e = _engine()
    e._pool_give(4096, 111)
    e._pool_give(8192, 333)
    with _QueueScope(e, 1):
        e._pool_take(4096)                  # syncs; still on queue 1
        e._pool_give(4096, 222)             # freed by queue-1 work
    n = e.lib.calls.count('sync')
    assert e._pool_take(8192) == 333        # idle since the sync
    assert e.lib.calls.count('sync') == n
    assert e._pool_take(4096) == 222        # back on queue 0: must wait for queue 1
    assert e.lib.calls.count('sync') == n + 1
