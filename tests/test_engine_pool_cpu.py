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

    def trk_thread_queue(self, ctx, q):
        self.calls.append(('thread_queue', q))
        return 0

    def trk_queue_sync(self, ctx, q):
        self.calls.append(('queue_sync', q))
        return 0


def _engine():
    e = Engine.__new__(Engine)
    e.lib, e.ctx = _Lib(), 1
    e._pool, e._pool_bytes, e._pool_limit = {}, 0, 1 << 30
    e._queue, e._multi_queue = 0, False
    import threading
    e._tls, e._pool_lock, e._thread_queues = threading.local(), threading.RLock(), False
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


def test_a_helper_thread_with_a_queue_of_its_own():
    """Round 4 (trk_thread_queue): the reader's thread works on queue 3 beside the caller's queue 0.  A buffer goes back
    to the pool tagged with the queue of the thread that frees it; the same thread takes it back without waiting, the
    other one waits for THAT queue only (not for the device); frees under idle_frees() are idle for everybody; buffers
    pooled before the helper appeared count as queue 0's."""
    import threading
    e = _engine()
    e._pool_give(4096, 100)                       # before any helper: "in order on queue 0"
    seen = {}

    def helper():
        e.thread_queue(3)
        seen['old'] = e._pool_take(4096)          # pooled by queue 0 before the helper existed: waits for queue 0
        seen['calls_old'] = list(e.lib.calls)
        e._pool_give(4096, 333)                   # freed by the helper: tagged 3
        n = len(e.lib.calls)
        seen['own'] = e._pool_take(4096)          # its own buffer back: no wait
        seen['own_waited'] = len(e.lib.calls) != n
        e._pool_give(4096, 333)
        with e.idle_frees():
            e._pool_give(8192, 444)               # freed after the helper waited for everything
    t = threading.Thread(target=helper)
    t.start()
    t.join()
    assert seen['old'] == 100 and ('queue_sync', 0) in seen['calls_old'] and 'sync' not in seen['calls_old']
    assert seen['own'] == 333 and not seen['own_waited']
    e.lib.calls.clear()
    assert e._pool_take(8192) == 444 and e.lib.calls == []          # idle: nobody waits
    assert e._pool_take(4096) == 333 and e.lib.calls == [('queue_sync', 3)]
    e.lib.calls.clear()
    e._pool_give(4096, 555)                       # the caller's own free (queue 0) and take
    assert e._pool_take(4096) == 555 and e.lib.calls == []


def test_held_array_goes_back_to_the_pool_only_after_unhold():
    """DeviceArray.hold(): somebody who is not the owner may still want the array's bytes -- the owner's free() is
    carried out by unhold()."""
    from trtools_amd.engine import DeviceArray
    e = _engine()
    e._live = set()
    a = DeviceArray.adopt(e, (1024,), 'u1', 777, 4096)
    a.hold()
    a.free()
    assert a.ptr is None and e._pool_take(4096) is None      # not in the pool yet
    a.unhold()
    assert e._pool_take(4096) == 777
    b = DeviceArray.adopt(e, (1024,), 'u1', 888, 4096)
    b.hold()
    b.unhold()                                               # nobody freed it in between: still the owner's
    assert b.ptr == 888 and e._pool_take(4096) is None
    b.free()
    assert e._pool_take(4096) == 888


def test_held_plane_of_the_reserved_pair_goes_back_to_the_context():
    """unhold()'s late free must not put a plane of the reserved pair (DeviceArray._reserved: lent by
    trk_dev_alloc_pair) into the buffer pool -- it is handed back through trk_dev_free, as free() does."""
    from trtools_amd.engine import DeviceArray
    e = _engine()
    e._live = set()
    freed = []
    e.lib.trk_dev_free = lambda ctx, p: freed.append(p) or 0
    for held in (True, False):
        a = DeviceArray.adopt(e, (1024,), 'u1', 4242, 4096)
        a._reserved = True
        if held:
            a.hold()
        a.free()
        if held:
            assert freed == []                                # somebody still reads it
            a.unhold()
        assert freed == [4242] and e._pool_take(4096) is None and e._pool_bytes == 0
        freed.clear()
