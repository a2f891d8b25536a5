"""Engine: thin Python object over the libtrk C ABI (one per process per GPU).

Host code stays Python (BASELINE.json north_star); all per-sample arithmetic
runs in the HIP kernels of libtrk.so.  Arrays handed to the engine are numpy
arrays with the documented dtypes; ``upload`` makes the PCIe copy explicit and
returns a ``DeviceArray``; results stay on the device until ``.get()``.
"""
from . import _knobs
import ctypes as C
import os

import numpy as np

from . import _lib as L


class DeviceArray:
    """A typed view of device memory owned by an Engine."""

    def __init__(self, eng, shape, dtype, _ptr=None, _parent=None):
        self.eng = eng
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.parent = _parent
        if _parent is not None:          # a window of another array's memory (not owned)
            self.ptr = int(_ptr)
            return
        self.cap = eng._size_class(max(self.nbytes, 16))
        self.ptr = eng._pool_take(self.cap)
        if self.ptr is None:
            ptr = C.c_void_p()
            eng._chk(eng.lib.trk_dev_alloc(eng.ctx, self.cap, C.byref(ptr)))
            self.ptr = ptr.value
        eng._live.add(self)

    @classmethod
    def adopt(cls, eng, shape, dtype, ptr, cap, reserved=False):
        """An owning array over a device allocation of ``cap`` bytes made elsewhere (trk_dev_alloc_pair); ``cap`` must
        be the engine's size class of the array, so that ``free`` can pool it like any other buffer.  ``reserved``:
        a plane of the context's reserved pair -- ``free`` hands it back to the context, never to the pool."""
        self = cls.__new__(cls)
        self._reserved = bool(reserved)
        self.eng = eng
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.parent = None
        self.cap = int(cap)
        if self.cap < self.nbytes:
            raise ValueError("allocation smaller than the array")
        self.ptr = int(ptr)
        eng._live.add(self)
        return self

    def view(self, offset_bytes, shape, dtype):
        """A typed window [offset_bytes, ...) of this array's memory; the parent keeps ownership.  Lets a caller lay
        several result arrays out back to back in one allocation (one RCCL all-reduce over all of them)."""
        v = DeviceArray(self.eng, shape, dtype, _ptr=self.ptr + int(offset_bytes), _parent=self)
        if offset_bytes < 0 or offset_bytes + v.nbytes > self.nbytes or (self.ptr + offset_bytes) % v.dtype.itemsize:
            raise ValueError("view outside its parent or misaligned")
        return v

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            self.eng._chk(self.eng.lib.trk_memcpy_d2h(self.eng.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def get_rows(self, lo, hi):
        """Copy rows [lo, hi) of the leading axis to the host (partial D2H)."""
        row = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        out = np.empty((hi - lo,) + self.shape[1:], dtype=self.dtype)
        if hi > lo and row:
            self.eng._chk(self.eng.lib.trk_memcpy_d2h(self.eng.ctx, out.ctypes.data, self.ptr + lo * row,
                                                      (hi - lo) * row))
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.shape != self.shape:
            raise ValueError("shape mismatch %s vs %s" % (arr.shape, self.shape))
        if self.nbytes:
            self.eng._chk(self.eng.lib.trk_memcpy_h2d(self.eng.ctx, self.ptr, arr.ctypes.data, self.nbytes))
        return self

    def copy_from(self, other):
        """Device-to-device copy (async on the engine's stream)."""
        if other.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        if self.nbytes:
            self.eng._chk(self.eng.lib.trk_memcpy_d2d(self.eng.ctx, self.ptr, other.ptr, self.nbytes))
        return self

    def zero(self):
        if self.nbytes:
            self.eng._chk(self.eng.lib.trk_memset(self.eng.ctx, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        """Give the memory back: to the engine's pool of device buffers (reused by the next array of the same size
        class -- a CLI run allocates the same dozen buffers for every batch, and hipMalloc / hipFree each
        synchronise the device), or to the driver when pooling is off or the pool is full."""
        if getattr(self, '_holds', 0) > 0 and self.parent is None and self.ptr is not None:
            self._late, self.ptr = self.ptr, None       # (somebody may still want its bytes: unhold() gives it back)
            self.eng._live.discard(self)
            return
        if self.parent is None and self.ptr is not None and self.eng.ctx is not None:
            if getattr(self, '_reserved', False) or not self.eng._pool_give(self.cap, self.ptr):
                self.eng.lib.trk_dev_free(self.eng.ctx, self.ptr)
        self.ptr = None
        self.eng._live.discard(self)

    def hold(self):
        """Keep the MEMORY (not the array) until ``unhold``: a free() in between -- by an owner that is done with the
        array -- is carried out then.  For a reader of the bytes that is not the owner (RawBatch: the host copies of a
        device-parsed batch are made only if somebody asks for them)."""
        self._holds = getattr(self, '_holds', 0) + 1
        return self

    def unhold(self):
        self._holds -= 1
        late = getattr(self, '_late', None)
        if self._holds == 0 and late is not None:
            self._late = None
            # (as free(): a plane of the reserved pair goes back to the context, never into the buffer pool)
            if self.eng.ctx is not None and (getattr(self, '_reserved', False) or not self.eng._pool_give(self.cap, late)):
                self.eng.lib.trk_dev_free(self.eng.ctx, late)


class DeviceBatch:
    """trk_batch + the DeviceArrays that back it."""

    def __init__(self, struct, arrays, n_groups, sum_alleles):
        self.struct = struct
        self.arrays = arrays
        self.n_loci = struct.n_loci
        self.n_samples = struct.n_samples
        self.ploidy = struct.ploidy
        self.n_groups = n_groups
        self.sum_alleles = sum_alleles
        self._class_runs = None      # host table behind struct.class_runs (sorted_by_class): must outlive the struct
        self.class_sorted = False

    def _derived(self, s, arrays, n_groups):
        """A batch made from a copy of this one's struct: whatever host memory the struct points at stays alive."""
        out = DeviceBatch(s, arrays, n_groups, self.sum_alleles)
        out._class_runs, out.class_sorted = self._class_runs, self.class_sorted
        return out

    def with_gt(self, gt_dev):
        """Same allele tables, different genotype tensor (e.g. dumpSTR's masked GT)."""
        s = L.Batch()
        C.memmove(C.byref(s), C.byref(self.struct), C.sizeof(L.Batch))
        s.gt = gt_dev.ptr
        arrays = dict(self.arrays)
        arrays['gt'] = gt_dev
        return self._derived(s, arrays, self.n_groups)

    def with_groups(self, eng, group_bits, n_groups):
        """Same tensors, sample groups attached (bit g of group_bits[s]: sample s is in group g)."""
        s = L.Batch()
        C.memmove(C.byref(s), C.byref(self.struct), C.sizeof(L.Batch))
        gb = eng.upload(np.ascontiguousarray(group_bits, dtype=np.uint8))
        s.group_bits = gb.ptr
        s.n_groups = int(n_groups)
        arrays = dict(self.arrays)
        arrays['group_bits'] = gb
        return self._derived(s, arrays, int(n_groups))


    def sorted_by_class(self, eng, group_bits, n_groups):
        """Sample groups at the ungrouped streaming kernel's rate (trk_batch.class_runs): the genotype columns are
        gathered on the device so that the samples of one class (one pattern of group bits) are neighbours -- every
        class a 16-byte aligned column range, padded with no-call columns, samples in no group dropped -- and the
        batch carries the run table; trk_locus_stats then counts each range with the ungrouped kernel and adds the
        classes into their groups.  ``group_bits``: host uint8[S] of THIS batch's columns (padding columns 0).
        Returns a new DeviceBatch that owns the gathered tensor (free its arrays['gt'] / ['group_bits'] when done);
        diploid batches only -- others get ``with_groups``.  (The command line lays its columns out by class while the
        reader parses them -- ``class_layout`` / ``with_class_layout`` -- and needs no gather.)"""
        gb = np.ascontiguousarray(group_bits, dtype=np.uint8) & np.uint8((1 << int(n_groups)) - 1)
        S = self.n_samples
        if gb.shape != (S,):
            raise ValueError("group_bits must have one entry per sample column")
        if self.ploidy != 2 or self.n_loci == 0:
            return self.with_groups(eng, gb, n_groups)
        lay = class_layout(gb, n_groups, row_align=4)
        if lay is None:
            return self.with_groups(eng, gb, n_groups)
        S2 = lay['n_out']
        col_d = eng.upload(lay['cols'])
        gt2 = eng.empty((self.n_loci, S2, 2), np.int16)
        eng._chk(eng.lib.trk_permute_columns(eng.ctx, self.arrays['gt'].ptr, gt2.ptr, col_d.ptr, self.n_loci, S, S2, 2))
        col_d.free()
        return self.with_class_layout(eng, lay, n_groups, gt2)

    def with_class_layout(self, eng, lay, n_groups, gt_sorted=None):
        """This batch's tables over a genotype tensor whose columns are in ``class_layout`` order (``gt_sorted``: a
        DeviceArray [L, n_out, 2]; None: this batch's own tensor already is)."""
        s = L.Batch()
        C.memmove(C.byref(s), C.byref(self.struct), C.sizeof(L.Batch))
        gbd = eng.upload(lay['bits'])
        table = lay['runs']
        arrays = dict(self.arrays)
        if gt_sorted is not None:
            arrays['gt'] = gt_sorted
        s.gt, s.n_samples, s.n_pad_samples = arrays['gt'].ptr, int(lay['n_out']), 0
        s.group_bits, s.n_groups = gbd.ptr, int(n_groups)
        s.n_class_runs, s.class_runs = len(table) // 4, table.ctypes.data
        arrays['group_bits'] = gbd
        out = DeviceBatch(s, arrays, int(n_groups), self.sum_alleles)
        out._class_runs = table          # host memory the struct points at
        out.class_sorted = True
        return out


def class_layout(group_bits, n_groups, row_align=32):
    """Column order of a cohort by sample CLASS (pattern of group bits, statSTR.py:520-542): every class a run of
    columns starting on a multiple of four (16 bytes of diploid genotypes), padded with no-call columns; samples in no
    group are left out; the row padded to a multiple of ``row_align`` columns.  Returns None when no sample is in a group,
    else {'cols': int32 [n_out] file column of each output column (-1: a padding column), 'col_of': int32 [S] output
    column of each file column (-1: dropped), 'bits': uint8 [n_out], 'runs': int32 table of (start, padded length,
    samples, class bits) per run, 'n_out'}."""
    gb = np.ascontiguousarray(group_bits, dtype=np.uint8) & np.uint8((1 << int(n_groups)) - 1)
    cols, bits, runs, at = [], [], [], 0
    for c in np.unique(gb):
        if c == 0:
            continue
        idx = np.flatnonzero(gb == c).astype(np.int32)
        pad = (-idx.size) % 4
        runs.append((at, idx.size + pad, idx.size, int(c)))
        at += idx.size + pad
        cols.append(np.concatenate([idx, np.full(pad, -1, dtype=np.int32)]))
        bits.append(np.concatenate([np.full(idx.size, c, dtype=np.uint8), np.zeros(pad, dtype=np.uint8)]))
    if not runs:
        return None
    tail = (-at) % max(4, int(row_align))
    if tail:
        cols.append(np.full(tail, -1, dtype=np.int32))
        bits.append(np.zeros(tail, dtype=np.uint8))
    col = np.ascontiguousarray(np.concatenate(cols))
    col_of = np.full(gb.shape[0], -1, dtype=np.int32)
    real = col >= 0
    col_of[col[real]] = np.flatnonzero(real).astype(np.int32)
    return dict(cols=col, col_of=col_of, bits=np.ascontiguousarray(np.concatenate(bits)),
                runs=np.ascontiguousarray(np.array(runs, dtype=np.int32).reshape(-1)), n_out=int(col.size))


class StatsResult:
    def __init__(self, allele_count, locus_int, locus_f64):
        self.allele_count = allele_count
        self.locus_int = locus_int
        self.locus_f64 = locus_f64
        self.struct = L.StatsOut(allele_count.ptr, locus_int.ptr, locus_f64.ptr if locus_f64 else None)


class AssocResult:
    def __init__(self, locus_int, locus_f64, allele_count):
        self.locus_int, self.locus_f64, self.allele_count = locus_int, locus_f64, allele_count
        self.struct = L.AssocOut(locus_int.ptr, locus_f64.ptr, allele_count.ptr)


class CallResult:
    def __init__(self, gt_out, filter_mask, sample_counters, sample_totaldp, sample_dp_missing, error,
                 sample_totaldp_f64=None, filter_mask8=None):
        self.sample_totaldp_f64 = sample_totaldp_f64
        self.filter_mask8 = filter_mask8
        self.gt_out = gt_out
        self.filter_mask = filter_mask
        self.sample_counters = sample_counters
        self.sample_totaldp = sample_totaldp
        self.sample_dp_missing = sample_dp_missing
        self.error = error
        self.struct = L.CallOut(gt_out.ptr if gt_out else None, filter_mask.ptr if filter_mask else None,
                                sample_counters.ptr, sample_totaldp.ptr, sample_dp_missing.ptr, error.ptr,
                                None, None, sample_totaldp_f64.ptr if sample_totaldp_f64 is not None else None,
                                filter_mask8.ptr if filter_mask8 is not None else None)

    def with_delta(self, stats):
        """trk_call_out whose delta outputs point at ``stats`` (counts of the unfiltered genotypes)."""
        s = L.CallOut()
        C.memmove(C.byref(s), C.byref(self.struct), C.sizeof(L.CallOut))
        s.delta_allele_count = stats.allele_count.ptr
        s.delta_locus_int = stats.locus_int.ptr
        return s


class _QueueScope:
    def __init__(self, eng, queue):
        self.eng, self.queue = eng, int(queue)

    def __enter__(self):
        self.eng._chk(self.eng.lib.trk_stream_select(self.eng.ctx, self.queue))
        self.eng._queue = self.queue
        if self.queue != 0:
            # buffers freed while only queue 0 ran were pooled as idle: queue 0 may still be working on them when
            # another queue takes one, so from here on none of them is
            self.eng._multi_queue = True
            for rest in self.eng._pool.values():
                rest[:] = [(p, False) for p, _ in rest]
        return self

    def __exit__(self, *exc):
        self.eng._chk(self.eng.lib.trk_stream_select(self.eng.ctx, 0))
        self.eng._queue = 0
        return False


class Engine:
    # Planes the context reserves for the call-filter pass's two outputs as the process's FIRST device allocations
    # (trk_reserve_pair; include/trk.h: a process's first two allocations lie in two different placement classes, 8 of
    # 8 fresh processes, profiles/r05_class_probe.txt).  ``reserve_pair_gb``: GiB per plane; None: TRK_RESERVE_PAIR_GB,
    # else 4 on a device with at least 64 GB (a 100 000 x 10 000 cohort's planes are 4.0 GB), else nothing; 0: nothing
    # (the command lines: their per-batch planes are far below the 256 MB from which placement shows at all).
    last_reservation = None

    def __init__(self, device=0, reserve_pair_gb=None):
        self.lib = L.load()
        self.ctx = None
        self._live = set()
        self._pool = {}              # size class -> [(device pointer, idle: no queue can still be using it)]
        import threading
        self._pool_lock = threading.RLock()
        self._tls = threading.local()
        self._thread_queues = False  # a thread has asked for a queue of its own (thread_queue): frees carry their queue
        self._pool_bytes = 0
        self._queue = 0              # the selected queue (on_queue)
        self._multi_queue = False    # another queue than 0 has been used since the last full synchronisation
        # the pool holds at most TRK_POOL_GB of freed buffers until close() / trim() (0 = no pooling)
        self._pool_limit = int(float(_knobs.env('TRK_POOL_GB', '8')) * (1 << 30))
        self._pinned = []            # pointers of hipHostMalloc'ed staging buffers
        self._pinned_cls = {}        # pointer -> size class of the buffers handed out
        self._pinned_free = {}       # size class -> released pointers
        ctx = C.c_void_p()
        rc = self.lib.trk_init(int(device), C.byref(ctx))
        if rc != 0:
            msg = self.lib.trk_last_error(None)
            raise L.TrkError("trk_init(device=%d) failed [%d]: %s -- this package needs an MI355X; "
                             "there is no CPU fallback" % (device, rc, msg.decode() if msg else '?'))
        self.ctx = ctx
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(64)
        ncu = C.c_int()
        hbm = C.c_uint64()
        self._chk(self.lib.trk_device_info(self.ctx, name, 256, C.byref(ncu), C.byref(hbm), arch, 64))
        self.device_name = name.value.decode()
        self.arch = arch.value.decode()
        self.n_cu = ncu.value
        self.hbm_bytes = hbm.value
        if reserve_pair_gb is None:
            env = _knobs.env('TRK_RESERVE_PAIR_GB')
            reserve_pair_gb = float(env) if env else (4.0 if self.hbm_bytes >= (64 << 30) else 0.0)
        self.reserved_pair_bytes = 0
        if reserve_pair_gb and reserve_pair_gb > 0:
            info = L.PairInfo()
            nbytes = int(float(reserve_pair_gb) * (1 << 30))
            if self.lib.trk_reserve_pair(self.ctx, nbytes, C.byref(info)) == 0:    # (out of memory: no reservation)
                # (planes large enough to be probed are kept only when the pair is fast: one that would never be lent
                # is given back by trk_reserve_pair itself)
                self.reserved_pair_bytes = nbytes if (info.placed or nbytes < (1 << 28)) else 0
                Engine.last_reservation = dict(probe_ms=[round(float(info.probe_ms[k]), 3) for k in range(info.n_probed)],
                                               kept_ms=round(float(info.kept_ms), 3), fast=bool(info.placed),
                                               seconds=round(float(info.seconds), 4), plane_bytes=nbytes)

    # ---- plumbing ----
    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.trk_last_error(self.ctx)
            raise L.TrkError("libtrk error %d: %s" % (rc, msg.decode() if msg else '?'))

    # ---- device buffer pool / pinned staging ----
    @staticmethod
    def _size_class(nbytes):
        """Sizes are rounded up to 1/8 steps of a power of two (<= 12.5 % waste) so that the slightly smaller last
        batch of a file reuses the buffers of the full ones."""
        if nbytes <= 4096:
            return 4096
        step = 1 << (int(nbytes - 1).bit_length() - 3)
        return (nbytes + step - 1) // step * step

    def thread_queue(self, queue):
        """trk_thread_queue: the CALLING thread's copies and kernels go to ``queue`` from now on (-1: back to the selected
        queue).  For a helper thread that uploads and parses the next batch beside the caller's thread; buffers it frees
        are tagged with its queue, and whoever takes one on another queue waits for that queue first."""
        self._chk(self.lib.trk_thread_queue(self.ctx, int(queue)))
        self._tls.q = int(queue)
        if queue >= 0:
            with self._pool_lock:
                if not self._thread_queues:
                    self._thread_queues = True
                    for rest in self._pool.values():       # pooled as "in order on queue 0": say so
                        rest[:] = [(p, 0 if t is True else t) for p, t in rest]

    def _my_queue(self):
        q = getattr(self._tls, 'q', -1)
        return q if q >= 0 else self._queue

    class _IdleFrees:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            self.eng._tls.idle = getattr(self.eng._tls, 'idle', 0) + 1

        def __exit__(self, *exc):
            self.eng._tls.idle -= 1
            return False

    def idle_frees(self):
        """``with eng.idle_frees(): a.free()`` -- the caller has waited for every queue that touched the buffers it frees
        inside (they go back to the pool as idle for everybody)."""
        return Engine._IdleFrees(self)

    def _pool_take(self, cap):
        """A pooled buffer of this size class.  free() is NOT a synchronisation point (hipFree was): work enqueued
        earlier may still read or write a freed buffer.  A pooled buffer carries a tag: True (idle), a queue number
        (freed by a thread working on that queue: in order there, anybody else waits for that queue first), or False
        (queues were switched with on_queue -- bench.py's pipelined step -- and nobody knows which one touched it
        last: the first reuse waits for the whole device, after which every pooled buffer is idle again)."""
        # (the pool is shared with the reader's read-ahead thread when it parses on the device: one lock around its books)
        wait_for = None
        with self._pool_lock:
            lst = self._pool.get(cap)
            if not lst:
                return None
            ptr, tag = lst.pop()
            self._pool_bytes -= cap
            if tag is True:
                return ptr
            if tag is False:
                self.sync()
                # still inside an on_queue(k != 0) scope: later frees are not idle either
                self._multi_queue = self._queue != 0
                for rest in self._pool.values():
                    rest[:] = [(p, True) for p, _ in rest]
                return ptr
            if tag != self._my_queue():
                wait_for = tag
        if wait_for is not None:
            self._chk(self.lib.trk_queue_sync(self.ctx, wait_for))     # (outside the lock: the other thread goes on)
        return ptr

    def _pool_give(self, cap, ptr):
        with self._pool_lock:
            if self._pool_limit <= 0 or self._pool_bytes + cap > self._pool_limit:
                return False
            if getattr(self._tls, 'idle', 0) > 0:
                tag = True
            elif self._multi_queue or (self._queue != 0 and getattr(self._tls, 'q', -1) < 0):
                tag = False
            elif self._thread_queues:
                tag = self._my_queue()
            else:
                tag = True
            self._pool.setdefault(cap, []).append((ptr, tag))
            self._pool_bytes += cap
            return True

    def trim(self):
        """Return every pooled device buffer to the driver."""
        for lst in self._pool.values():
            for ptr, _ in lst:
                self.lib.trk_dev_free(self.ctx, ptr)
        self._pool, self._pool_bytes = {}, 0

    def host_buffer(self, nbytes):
        """A pinned (page-locked, hipHostMalloc) host buffer as a numpy uint8 array: a copy from / to it runs at the
        full PCIe rate and its pages never fault.  Taken from the engine's pool of released buffers when one of the
        same size class is free (an allocation costs ~15 ms whatever its size); lives until ``host_buffer_release``
        or the end of the engine."""
        want = max(int(nbytes), 16)
        cls = self._size_class(want)
        free = self._pinned_free.get(cls)
        if free:
            ptr = free.pop()
        else:
            p = C.c_void_p()
            self._chk(self.lib.trk_host_alloc(self.ctx, cls, C.byref(p)))
            ptr = p.value
            self._pinned.append(ptr)
        self._pinned_cls[ptr] = cls
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(cls,))[:want]

    def host_buffer_release(self, arr):
        """Hand a ``host_buffer`` array back to the pool (the caller must not touch it afterwards, and no copy from or
        to it may still be in flight)."""
        base = arr
        while getattr(base, 'base', None) is not None and isinstance(base.base, np.ndarray):
            base = base.base
        ptr = base.ctypes.data
        cls = self._pinned_cls.pop(ptr, None)
        if cls is not None:
            self._pinned_free.setdefault(cls, []).append(ptr)

    def close(self):
        if self.ctx is not None:
            for a in list(self._live):
                a.free()
            self.trim()
            for ptr in self._pinned:
                self.lib.trk_host_free(self.ctx, ptr)
            self._pinned, self._pinned_cls, self._pinned_free = [], {}, {}
            self.lib.trk_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        return DeviceArray(self, shape, dtype).zero()

    def upload(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        return DeviceArray(self, arr.shape, arr.dtype).set(arr)

    def sync(self):
        self._chk(self.lib.trk_sync(self.ctx))

    def timer_start(self, slot=0):
        self._chk(self.lib.trk_timer_start(self.ctx, slot))

    def timer_stop(self, slot=0):
        self._chk(self.lib.trk_timer_stop(self.ctx, slot))

    def timer_ms(self, slot=0):
        ms = C.c_float()
        self._chk(self.lib.trk_timer_elapsed_ms(self.ctx, slot, C.byref(ms)))
        return ms.value

    def profile(self, on=True):
        self._chk(self.lib.trk_profile_enable(self.ctx, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.lib.trk_profile_reset(self.ctx))

    def profile_get(self):
        out = {}
        for k, name in enumerate(L.KERNEL_NAMES):
            n = C.c_int64()
            ms = C.c_double()
            self._chk(self.lib.trk_profile_get(self.ctx, k, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    # ---- batches ----
    def make_batch(self, gt, allele_off, len_class, str_class, len_class_value,
                   locus_ploidy=None, group_bits=None, n_groups=1, max_alleles=None, n_pad=0):
        """Upload host arrays (numpy) or accept DeviceArrays; returns a DeviceBatch.  n_pad: the last n_pad samples
        of every row are padding (genotype -1), see trk_batch.n_pad_samples."""
        def dev(x, dt):
            return x if isinstance(x, DeviceArray) else self.upload(x, dt)
        gt_d = dev(gt, np.int16)
        if len(gt_d.shape) != 3:
            raise ValueError("gt must be [L, S, P]")
        Lc, S, P = gt_d.shape
        off_host = None
        if not isinstance(allele_off, DeviceArray):
            off_host = np.ascontiguousarray(allele_off, dtype=np.int32)
            if off_host.shape != (Lc + 1,):
                raise ValueError("allele_off must have L+1 entries")
        off_d = dev(allele_off, np.int32)
        lc_d = dev(len_class, np.uint16)
        sc_d = dev(str_class, np.uint16)
        cv_d = dev(len_class_value, np.float64)
        sumA = lc_d.shape[0]
        if off_host is not None:
            if int(off_host[-1]) != sumA:
                raise ValueError("allele_off[-1] != len(len_class)")
            if max_alleles is None:
                max_alleles = int(np.max(np.diff(off_host))) if Lc else 0
        arrays = dict(gt=gt_d, allele_off=off_d, len_class=lc_d, str_class=sc_d, len_class_value=cv_d)
        s = L.Batch()
        s.n_loci, s.n_samples, s.ploidy = Lc, S, P
        s.n_groups = int(n_groups)
        s.n_pad_samples = int(n_pad)
        s.n_alleles_total = sumA
        s.max_alleles = int(max_alleles or 0)
        s.gt, s.allele_off = gt_d.ptr, off_d.ptr
        s.len_class, s.str_class, s.len_class_value = lc_d.ptr, sc_d.ptr, cv_d.ptr
        if locus_ploidy is not None:
            arrays['locus_ploidy'] = dev(locus_ploidy, np.uint8)
            s.locus_ploidy = arrays['locus_ploidy'].ptr
        if group_bits is not None:
            arrays['group_bits'] = dev(group_bits, np.uint8)
            s.group_bits = arrays['group_bits'].ptr
        else:
            n_groups = 1
        return DeviceBatch(s, arrays, int(n_groups), sumA)

    # ---- hot path ----
    def alloc_stats(self, batch, twin=False):
        """Result arrays of trk_locus_stats.  ``twin``: the count arrays hold two copies back to back
        (TRK_STATS_TWIN); ``.twin`` is a StatsResult over the second copy with its own float columns."""
        G = batch.n_groups
        if not twin:
            return StatsResult(self.empty((G, batch.sum_alleles), np.int32),
                               self.empty((G, batch.n_loci, L.TRK_LI_COLS), np.int32),
                               self.empty((G, batch.n_loci, L.TRK_LF_COLS), np.float64))
        n_ac, n_li = G * batch.sum_alleles, G * batch.n_loci * L.TRK_LI_COLS
        ac = self.empty((2, G, batch.sum_alleles), np.int32)
        li = self.empty((2, G, batch.n_loci, L.TRK_LI_COLS), np.int32)
        first = StatsResult(ac.view(0, (G, batch.sum_alleles), np.int32),
                            li.view(0, (G, batch.n_loci, L.TRK_LI_COLS), np.int32),
                            self.empty((G, batch.n_loci, L.TRK_LF_COLS), np.float64))
        first.twin = StatsResult(ac.view(n_ac * 4, (G, batch.sum_alleles), np.int32),
                                 li.view(n_li * 4, (G, batch.n_loci, L.TRK_LI_COLS), np.int32),
                                 self.empty((G, batch.n_loci, L.TRK_LF_COLS), np.float64))
        first._owners = (ac, li)
        return first

    def locus_stats(self, batch, nalleles_thresh=0.01, out=None, count_only=False):
        if out is None:
            out = self.alloc_stats(batch)
        flags = (L.STATS_COUNT_ONLY if count_only else 0) | (L.STATS_TWIN if getattr(out, 'twin', None) else 0)
        prm = L.StatsParams(float(nalleles_thresh), flags, 0)
        self._chk(self.lib.trk_locus_stats(self.ctx, C.byref(batch.struct), C.byref(prm), C.byref(out.struct)))
        return out

    import threading as _threading
    _pool_lock = _threading.RLock()      # (class-level default; every engine gets its own in __init__)
    _tls = _threading.local()
    _thread_queues = False

    # What the last placed allocation saw (bench.py reports it): trk_pair_info as a dict
    last_placement = None
    PLACE_MIN_BYTES = 1 << 28      # planes from 256 MB on are placed (at 67 MB no levels can be told apart)

    def placed_output_pair(self, batch, max_spare=None):
        """The two big output planes of a call-filter pass (masked genotypes [L, S, 2] int16, filter mask [L, S]
        uint32) through trk_dev_alloc_pair: on MI355X the pass's two write streams run on one of two levels, 10-18 %
        apart, decided by which allocations the two planes are (profiles/r03_notes.md section 22); the library times
        the write-only half of the stream over the first plane and up to 1 + ``max_spare`` candidates for the second
        (default 2 spare planes: TRK_PLACE_SPARE) and keeps the fastest pair.  Buffers of the size class that the engine's
        pool holds are the first candidates; the ones not taken go back to the pool."""
        Lc, S = batch.n_loci, batch.n_samples
        nbytes = Lc * S * 4
        cap = self._size_class(max(nbytes, 16))
        if max_spare is None:
            max_spare = int(_knobs.lab('TRK_PLACE_SPARE', '2'))
        a, b, info = C.c_void_p(), C.c_void_p(), L.PairInfo()
        # buffers of this size class the pool holds are the first candidates (idle ones only: a probe writes them)
        have = []
        while len(have) < 4:
            lst = self._pool.get(cap)
            if not lst:
                break
            p_ = self._pool_take(cap)
            if p_ is None:
                break
            have.append(p_)
        harr = (C.c_void_p * max(len(have), 1))(*have)
        info.have_a = info.have_b = -1       # (a call that fails before it writes *info has taken none of them)
        try:
            self._chk(self.lib.trk_dev_alloc_pair(self.ctx, cap, Lc, S, int(max_spare), harr, len(have), C.byref(a),
                                                  C.byref(b), C.byref(info)))
        finally:
            used = {info.have_a, info.have_b}
            for k, p_ in enumerate(have):
                if k not in used:
                    if not self._pool_give(cap, p_):
                        self.lib.trk_dev_free(self.ctx, p_)
        Engine.last_placement = dict(probe_ms=[round(float(info.probe_ms[k]), 3) for k in range(info.n_probed)],
                                     kept_ms=round(float(info.kept_ms), 3), placed=bool(info.placed), jumps=int(info.n_jumps),
                                     seconds=round(float(info.seconds), 4), peak_extra_bytes=int(info.peak_extra_bytes),
                                     plane_bytes=int(cap), reserved=bool(info.reserved))
        res = bool(info.reserved)
        g = DeviceArray.adopt(self, (Lc, S, 2), np.int16, a.value, cap, reserved=res)
        m = DeviceArray.adopt(self, (Lc, S), np.uint32, b.value, cap, reserved=res)
        return g, m

    def alloc_call_out(self, batch, n_filters, want_gt=True, want_mask=True, want_mask8=False, place=None,
                       in_place=False):
        """Outputs of trk_call_filters for ``batch``.  The two big planes (masked genotypes, mask) of a diploid batch
        are PLACED from 256 MB each on (``placed_output_pair``; ``place=False`` or TRK_PLACE_OUTPUTS=0: plain
        allocations) -- the command lines' per-batch outputs, a strong-scaling shard's and the bench's alike.
        ``in_place=True``: the masked genotypes are written into the batch's own tensor (``gt_out`` IS ``batch.gt``,
        as dumpSTR.py:721-727 updates its record): the pass stores only the 16-byte chunks that hold a filtered call
        and has ONE big write stream, so there is no pair to place."""
        S = batch.n_samples
        if place is None:
            place = _knobs.env('TRK_PLACE_OUTPUTS', '1') != '0'
        g = m = None
        if in_place and want_gt:
            g = batch.arrays['gt']
            m = self.empty((batch.n_loci, S), np.uint32) if want_mask else None
        elif (place and want_gt and want_mask and batch.ploidy == 2 and batch.n_loci * S * 4 >= self.PLACE_MIN_BYTES and
                S % 4 == 0):
            g, m = self.placed_output_pair(batch)
        else:
            g = self.empty((batch.n_loci, S, batch.ploidy), np.int16) if want_gt else None
            m = self.empty((batch.n_loci, S), np.uint32) if want_mask else None
        return CallResult(g, m, self.zeros((1 + n_filters, S), np.int64), self.zeros((S,), np.int64),
                          self.zeros((S,), np.int64), self.zeros((4,), np.int32), self.zeros((S,), np.float64),
                          self.empty((batch.n_loci, S), np.uint8) if want_mask8 else None)

    def locus_finalize(self, batch, stats, nalleles_thresh=0.01):
        """Float statistics + HWE test from counts already in ``stats`` (trk_locus_finalize)."""
        prm = L.StatsParams(float(nalleles_thresh), 0, 0)
        self._chk(self.lib.trk_locus_finalize(self.ctx, C.byref(batch.struct), C.byref(prm), C.byref(stats.struct)))
        return stats

    def call_filters(self, batch, planes, filters, dp_plane=-1, out=None, delta_stats=None):
        """planes: list of DeviceArray ([L,S] or [L,S,k], int32/float32);
        filters: list of dicts(op, plane_a, col_a=0, plane_b=-1, col_b=0, col_a2=0, thr=0.0);
        delta_stats: StatsResult holding the counts of the unfiltered genotypes, corrected in place
        to the counts of the masked genotypes (no second pass over the tensor)."""
        np_ = len(planes)
        nf = len(filters)
        if np_ > L.TRK_MAX_PLANES or nf > L.TRK_MAX_FILTERS:
            raise ValueError("too many planes/filters")
        parr = (L.Plane * max(np_, 1))()
        for i, p in enumerate(planes):
            planar = getattr(p, 'planar', False)      # [k, L, S]: one contiguous array per column
            if (p.shape[1:] if planar else p.shape[:2]) != (batch.n_loci, batch.n_samples):
                raise ValueError("plane %d has shape %s" % (i, p.shape))
            ncol = p.shape[0] if planar else (1 if len(p.shape) == 2 else p.shape[2])
            if p.dtype == np.int32:
                dt = L.DT_I32
            elif p.dtype == np.float32:
                dt = L.DT_F32
            else:
                raise ValueError("plane %d dtype %s not supported" % (i, p.dtype))
            parr[i] = L.Plane(p.ptr, dt | (L.DT_PLANAR if planar else 0), ncol)
        farr = (L.CallFilter * max(nf, 1))()
        for k, f in enumerate(filters):
            farr[k] = L.CallFilter(int(f['op']), int(f['plane_a']), int(f.get('col_a', 0)),
                                   int(f.get('plane_b', -1)), int(f.get('col_b', 0)),
                                   int(f.get('col_a2', 0)), float(f.get('thr', 0.0)))
        if out is None:
            out = self.alloc_call_out(batch, nf)
        ostruct = out.struct if delta_stats is None else out.with_delta(delta_stats)
        self._chk(self.lib.trk_call_filters(self.ctx, C.byref(batch.struct), parr, np_, farr, nf,
                                            int(dp_plane), C.byref(ostruct)))
        return out

    def on_queue(self, queue):
        """``with eng.on_queue(1): ...``: the calls inside enqueue on the context's queue ``queue`` (HIP stream);
        pair with ``queue_wait`` for the ordering against queue 0 (trk_stream_select / trk_stream_wait)."""
        return _QueueScope(self, queue)

    def queue_wait(self, waiter, signal):
        """Queue ``waiter`` waits for everything enqueued so far on queue ``signal``."""
        self._chk(self.lib.trk_stream_wait(self.ctx, int(waiter), int(signal)))

    def event_record(self, slot):
        """Mark "everything enqueued so far on the selected queue" under ``slot`` (trk_event_record)."""
        self._chk(self.lib.trk_event_record(self.ctx, int(slot)))

    def event_wait(self, slot):
        """The selected queue waits for the mark last recorded under ``slot`` (trk_event_wait)."""
        self._chk(self.lib.trk_event_wait(self.ctx, int(slot)))

    def upload_plane(self, arr, n_pad=0):
        """Upload a FORMAT plane [L, S] or [L, S, k] (int32 / float32) as it is (n_pad: padding samples of missing
        values appended on the device, pad_samples).  Planes of up to four columns stay
        interleaved, as cyvcf2 hands them: the call-filter kernel fetches a thread's k 16-byte chunks and picks the
        columns out of its registers (5.45 against 3.8 ms on the GangSTR nine-filter set at 50k x 5k), which is
        cheaper than transposing them first (the transposition moves every byte twice: 5.5 ms for the same planes).
        Wider planes are made planar on the device ([k, L, S], TRK_DT_PLANAR: every column streams as 16-byte
        vectors; a host transpose of a 160 MB plane costs ~0.3 s).  TRK_CF_PLANARIZE=1 / 0 forces either."""
        arr = np.asarray(arr)
        d = self.pad_samples(self.upload(np.ascontiguousarray(arr)), n_pad)
        if arr.ndim == 3 and arr.shape[2] > 1:
            force = _knobs.lab('TRK_CF_PLANARIZE')
            if force == '1' or (force is None and arr.shape[2] > 4):
                p = self.planarize(d)
                d.free()
                return p
        return d

    PAD_FILL = {'i2': 0xffffffff, 'i4': 0x80000000, 'f4': 0x7fc00000}    # -1 genotypes / missing int32 / nan

    def pad_samples(self, d, n_pad):
        """Device [L, S, ...] (int16 pairs, int32 or float32) -> new device array [L, S + n_pad, ...] whose padding
        columns are missing values (trk_pad_rows); ``d`` is freed.  n_pad == 0: ``d`` itself."""
        if not n_pad:
            return d
        Lc, S = d.shape[0], d.shape[1]
        inner = int(np.prod(d.shape[2:], dtype=np.int64)) * d.dtype.itemsize
        if inner % 4:
            raise ValueError("pad_samples: %d bytes per sample" % inner)
        out = self.empty((Lc, S + n_pad) + tuple(d.shape[2:]), d.dtype)
        fill = self.PAD_FILL[d.dtype.str[1:]]
        self._chk(self.lib.trk_pad_rows(self.ctx, d.ptr, out.ptr, Lc, S * inner // 4, n_pad * inner // 4, fill))
        d.free()
        return out

    def planarize(self, plane):
        """Device [L, S, k] -> new device array [k, L, S] marked planar (trk_planarize)."""
        if len(plane.shape) != 3 or plane.shape[2] == 1:
            return plane
        Lc, S, k = plane.shape
        out = self.empty((k, Lc, S), plane.dtype)
        self._chk(self.lib.trk_planarize(self.ctx, plane.ptr, out.ptr, Lc * S, k))
        out.planar = True
        return out

    def locus_filters(self, n_loci, stats, min_callrate=None, min_hwep=None, min_het=None, max_het=None,
                      use_length=False, extern_bits=None, n_extern=0, bits_out=None, counters=None):
        nan = float('nan')
        spec = L.LocusFilterSpec(
            nan if min_callrate is None else float(min_callrate),
            nan if min_hwep is None else float(min_hwep),
            nan if min_het is None else float(min_het),
            nan if max_het is None else float(max_het),
            1 if use_length else 0, int(n_extern), extern_bits.ptr if extern_bits is not None else None)
        if bits_out is None:
            bits_out = self.empty((n_loci,), np.uint32)
        if counters is None:
            counters = self.zeros((L.TRK_LC_COLS,), np.int64)
        out = L.LocusOut(bits_out.ptr, counters.ptr)
        self._chk(self.lib.trk_locus_filters(self.ctx, int(n_loci), C.byref(stats.struct), C.byref(spec),
                                             C.byref(out)))
        return bits_out, counters

    # ---- scalar helpers ----
    # ---- associaTR scan ----
    def assoc_scan(self, batch, vec, allele_len, rlen_class, sample_in=None, non_major_cutoff=20.0, out=None):
        """trk_assoc_scan: vec [M, S] float64 (row 0 the standardised outcome, rows 1.. covariates),
        allele_len [sumA] float64, rlen_class [sumA] uint16, sample_in [S] uint8 or None.
        Returns AssocResult (device arrays locus_int [L, AI_COLS], locus_f64 [L, AF_COLS], allele_count [sumA])."""
        def dev(x, dt):
            return x if isinstance(x, DeviceArray) else self.upload(x, dt)
        vec_d = dev(vec, np.float64)
        if len(vec_d.shape) != 2 or vec_d.shape[1] != batch.n_samples:
            raise ValueError("vec must be [M, S]")
        len_d, rc_d = dev(allele_len, np.float64), dev(rlen_class, np.uint16)
        in_d = dev(sample_in, np.uint8) if sample_in is not None else None
        if out is None:
            out = AssocResult(self.empty((batch.n_loci, L.AI_COLS), np.int32),
                              self.empty((batch.n_loci, L.AF_COLS), np.float64),
                              self.empty((batch.sum_alleles,), np.int32))
        prm = L.AssocParams()
        prm.n_vec, prm.flags = vec_d.shape[0], 0
        prm.vec, prm.allele_len, prm.rlen_class = vec_d.ptr, len_d.ptr, rc_d.ptr
        prm.sample_in = in_d.ptr if in_d is not None else None
        prm.non_major_cutoff = float(non_major_cutoff)
        out._keep = (vec_d, len_d, rc_d, in_d)
        self._chk(self.lib.trk_assoc_scan(self.ctx, C.byref(batch.struct), C.byref(prm), C.byref(out.struct)))
        return out

    def assoc_scan_dosage(self, batch, vec, allele_len, rlen_class, ap1, ap2, perm, dclass, dclass_value, best_class,
                          sample_in=None):
        """trk_assoc_scan_dosage.  ap1/ap2 [L, S, K] float32; perm int32 / dclass uint16 / dclass_value float64 /
        best_class uint16, all [sumA].  Returns (AssocResult, class_sums [sumA, 4], locus_sums [L, 6]) on the device."""
        def dev(x, dt):
            return x if isinstance(x, DeviceArray) else self.upload(x, dt)
        vec_d = dev(vec, np.float64)
        len_d, rc_d = dev(allele_len, np.float64), dev(rlen_class, np.uint16)
        in_d = dev(sample_in, np.uint8) if sample_in is not None else None
        a1, a2 = dev(ap1, np.float32), dev(ap2, np.float32)
        if a1.shape != a2.shape or len(a1.shape) != 3 or a1.shape[:2] != (batch.n_loci, batch.n_samples):
            raise ValueError("ap1/ap2 must be [L, S, K]")
        pm, dc, dv, bc = dev(perm, np.int32), dev(dclass, np.uint16), dev(dclass_value, np.float64), dev(best_class, np.uint16)
        out = AssocResult(self.empty((batch.n_loci, L.AI_COLS), np.int32),
                          self.empty((batch.n_loci, L.AF_COLS), np.float64),
                          self.empty((batch.sum_alleles,), np.int32))
        cs = self.empty((batch.sum_alleles, L.ADC_COLS), np.float64)
        ls = self.empty((batch.n_loci, L.ADL_COLS), np.float64)
        prm = L.AssocParams()
        prm.n_vec, prm.flags = vec_d.shape[0], 0
        prm.vec, prm.allele_len, prm.rlen_class = vec_d.ptr, len_d.ptr, rc_d.ptr
        prm.sample_in = in_d.ptr if in_d is not None else None
        prm.non_major_cutoff = 0.0
        dos = L.AssocDosage()
        dos.ap1, dos.ap2, dos.n_alt_cols = a1.ptr, a2.ptr, a1.shape[2]
        dos.perm, dos.dclass, dos.dclass_value, dos.best_class = pm.ptr, dc.ptr, dv.ptr, bc.ptr
        out._keep = (vec_d, len_d, rc_d, in_d, a1, a2, pm, dc, dv, bc)
        self._chk(self.lib.trk_assoc_scan_dosage(self.ctx, C.byref(batch.struct), C.byref(prm), C.byref(dos),
                                                 C.byref(out.struct), cs.ptr, ls.ptr))
        return out, cs, ls

    def dosages(self, batch, allele_len, dosage_type, ap1=None, ap2=None):
        """trk_dosages -> (float32 [L, S] device array, int32 [L] error bits device array)."""
        def dev(x, dt):
            return x if isinstance(x, DeviceArray) else self.upload(x, dt)
        t = L.DOS_TYPES[dosage_type] if isinstance(dosage_type, str) else int(dosage_type)
        len_d = dev(allele_len, np.float64)
        a1 = dev(ap1, np.float32) if ap1 is not None else None
        a2 = dev(ap2, np.float32) if ap2 is not None else None
        out = self.empty((batch.n_loci, batch.n_samples), np.float32)
        err = self.empty((batch.n_loci,), np.int32)
        out._keep = (len_d, a1, a2)
        self._chk(self.lib.trk_dosages(self.ctx, C.byref(batch.struct), len_d.ptr, t, a1.ptr if a1 is not None else None,
                                       a2.ptr if a2 is not None else None, a1.shape[2] if a1 is not None else 0,
                                       out.ptr, err.ptr))
        return out, err

    def deflate_bgzf(self, data, address=None, nbytes=None, out=None):
        """trk_deflate_bgzf (include/trk.h): text in host memory -> its BGZF members (0xff00 bytes of text each, no end-of-file
        member), made on the device.  ``data``: bytes / bytearray, or None with ``address`` / ``nbytes`` of a host buffer
        (a pinned output block: no copy on the way up).  Returns a memoryview of the members in ``out`` (a bytearray kept
        and regrown by the engine when None: valid until the next call)."""
        if data is not None:
            nbytes = len(data)
            hold = (C.c_char * nbytes).from_buffer(data) if isinstance(data, bytearray) else C.c_char_p(bytes(data) if not isinstance(data, bytes) else data)
            address = C.addressof(hold) if isinstance(data, bytearray) else C.cast(hold, C.c_void_p).value
        need = int(self.lib.trk_deflate_bound(int(nbytes)))
        if out is None:
            out = getattr(self, '_deflate_out', None)
            if out is None or len(out) < need:
                out = self._deflate_out = bytearray(need)
        dst = (C.c_char * len(out)).from_buffer(out)
        got = C.c_size_t(0)
        try:
            self._chk(self.lib.trk_deflate_bgzf(self.ctx, C.c_void_p(address), int(nbytes), dst, len(out), C.byref(got)))
        finally:
            del dst
        return memoryview(out)[:got.value]

    def inflate_blocks(self, comp, in_off, in_len, out_off, out_len, text=None, text_bytes=None):
        """trk_inflate_blocks (include/trk.h): BGZF members inflated on the device.  comp: the compressed bytes (host
        bytes / uint8 array, or a DeviceArray); in_off / in_len: where each member's raw DEFLATE payload lies in comp;
        out_off / out_len: where its text goes and how long it is (the member's ISIZE).  Returns (text DeviceArray uint8,
        flags uint8 array on the host: 0 or _lib.INFLATE_* bits per member)."""
        n = int(np.asarray(in_off).shape[0])
        if isinstance(comp, DeviceArray):
            comp_d, own = comp, False
            n_comp = comp.nbytes
        else:
            host = np.frombuffer(comp, dtype=np.uint8) if not isinstance(comp, np.ndarray) else comp.view(np.uint8).reshape(-1)
            n_comp = int(host.shape[0])
            comp_d = self.empty((n_comp + 64,), np.uint8)
            if n_comp:
                self._chk(self.lib.trk_memcpy_h2d(self.ctx, comp_d.ptr, host.ctypes.data, n_comp))
            own = True
        oo = np.ascontiguousarray(out_off, dtype=np.int64)
        ol = np.ascontiguousarray(out_len, dtype=np.int32)
        if text is None:
            need = int((oo + ol).max()) if n else 0
            text = self.empty((max(need if text_bytes is None else int(text_bytes), 16) + 32,), np.uint8)
        io, il = self.upload(np.ascontiguousarray(in_off, dtype=np.int64), np.int64), self.upload(np.ascontiguousarray(in_len, dtype=np.int32), np.int32)
        oo_d, ol_d = self.upload(oo, np.int64), self.upload(ol, np.int32)
        flags = self.empty((max(n, 1),), np.uint8)
        pin = L.InflateIn(comp_d.ptr, n_comp, n, 0, io.ptr, il.ptr, oo_d.ptr, ol_d.ptr)
        pout = L.InflateOut(text.ptr, flags.ptr)
        self._chk(self.lib.trk_inflate_blocks(self.ctx, C.byref(pin), C.byref(pout)))
        fl = flags.get()[:n]
        for a in [io, il, oo_d, ol_d, flags] + ([comp_d] if own else []):
            a.free()
        return text, fl

    def parse_samples(self, text, smp_off, line_end, n_samples, ploidy, gt_idx, planes=(), want_phased=False):
        """trk_parse_samples (include/trk.h): the sample columns of L records parsed on the device.
        text: the batch's text (bytes / uint8 array on the host, or a DeviceArray that is 16-byte aligned and padded by
        16 bytes); smp_off / line_end: int64 [L] offsets INTO text of every record's first sample token and of its
        newline; gt_idx: int8 [L] index of GT among the record's FORMAT keys; planes: [(idx int8 [L], 'i' | 'f'), ...]
        (at most four scalar Integer / Float fields).  Returns a dict of DeviceArrays: gt int16 [L, S, P], planes
        (int32 / float32 [L, S]), locus_ploidy uint8 [L], flags uint8 [L] (0, or _lib.PARSE_* bits: the rows of a
        flagged record are undefined -- parse it with the host reader), phased uint8 [L, S] when asked for."""
        if isinstance(text, DeviceArray):
            text_d, own_text = text, False
        else:
            host = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text.view(np.uint8).reshape(-1)
            text_d = self.empty((host.shape[0] + 32,), np.uint8)
            if host.shape[0]:
                self._chk(self.lib.trk_memcpy_h2d(self.ctx, text_d.ptr, host.ctypes.data, host.shape[0]))
            own_text = True
        n_rec = int(np.asarray(smp_off).shape[0]) if not isinstance(smp_off, DeviceArray) else smp_off.shape[0]

        def dev(x, dt):
            return x if isinstance(x, DeviceArray) else self.upload(np.ascontiguousarray(x, dtype=dt), dt)
        tmp = []
        so, le, gi = dev(smp_off, np.int64), dev(line_end, np.int64), dev(gt_idx, np.int8)
        tmp += [x for x, src in ((so, smp_off), (le, line_end), (gi, gt_idx)) if not isinstance(src, DeviceArray)]
        if len(planes) > L.PARSE_MAX_PLANES:
            raise ValueError("at most %d planes" % L.PARSE_MAX_PLANES)
        pin = L.ParseIn()
        pin.text, pin.n_bytes = text_d.ptr, text_d.nbytes
        pin.n_records, pin.n_samples, pin.ploidy, pin.n_planes = n_rec, int(n_samples), int(ploidy), len(planes)
        pin.smp_off, pin.line_end, pin.gt_idx = so.ptr, le.ptr, gi.ptr
        out = dict(gt=self.empty((n_rec, n_samples, ploidy), np.int16), locus_ploidy=self.empty((n_rec,), np.uint8),
                   flags=self.empty((n_rec,), np.uint8), planes=[])
        pout = L.ParseOut()
        for i, (idx, kind) in enumerate(planes):
            d = dev(idx, np.int8)
            if not isinstance(idx, DeviceArray):
                tmp.append(d)
            pin.plane_idx[i] = d.ptr
            pin.plane_kind[i] = L.PARSE_FLOAT if kind == 'f' else L.PARSE_INT
            arr = self.empty((n_rec, n_samples), np.float32 if kind == 'f' else np.int32)
            out['planes'].append(arr)
            pout.planes[i] = arr.ptr
        pout.gt, pout.locus_ploidy, pout.flags = out['gt'].ptr, out['locus_ploidy'].ptr, out['flags'].ptr
        if want_phased:
            out['phased'] = self.empty((n_rec, n_samples), np.uint8)
            pout.phased = out['phased'].ptr
        self._chk(self.lib.trk_parse_samples(self.ctx, C.byref(pin), C.byref(pout)))
        if tmp or own_text:
            self.sync()                       # (the temporaries go back to the pool: the kernel must be done with them)
        with self.idle_frees():
            for t in tmp:
                t.free()
            if own_text:
                text_d.free()
        return out

    def qc_reduce(self, batch, quality=None, sample_in=None, ignore_no_call=False):
        """trk_qc_reduce (qcSTR's reductions, include/trk.h): per-sample and per-locus call counts, and with a
        quality plane (float32 [L, S], host or device) the quality sums and the numbers of entries summed.
        Returns a dict of device arrays: sample_calls, locus_calls (int64) and, with a plane, sample_qual_sum,
        locus_qual_sum (float64), sample_qual_n, locus_qual_n (int64)."""
        def dev(x, dt):
            return x if isinstance(x, DeviceArray) else self.upload(np.ascontiguousarray(x, dtype=dt), dt)
        nl, ns = batch.n_loci, batch.n_samples
        q = dev(quality, np.float32) if quality is not None else None
        sin = dev(sample_in, np.uint8) if sample_in is not None else None
        res = dict(sample_calls=self.empty((ns,), np.int64), locus_calls=self.empty((nl,), np.int64))
        if q is not None:
            res.update(sample_qual_sum=self.empty((ns,), np.float64), sample_qual_n=self.empty((ns,), np.int64),
                       locus_qual_sum=self.empty((nl,), np.float64), locus_qual_n=self.empty((nl,), np.int64))
        prm = L.QcParams(sin.ptr if sin is not None else None, q.ptr if q is not None else None,
                         1 if ignore_no_call else 0, 0)
        out = L.QcOut(*[res[k].ptr if k in res else None
                        for k in ('sample_calls', 'locus_calls', 'sample_qual_sum', 'sample_qual_n', 'locus_qual_sum',
                                  'locus_qual_n')])
        res['_keep'] = (q, sin)
        self._chk(self.lib.trk_qc_reduce(self.ctx, C.byref(batch.struct), C.byref(prm), C.byref(out)))
        return res

    def student_t_two_sided(self, t, df):
        return float(self.lib.trk_student_t_two_sided(float(t), float(df)))

    def binomtest(self, k, n, p):
        return self.lib.trk_binomtest_two_sided(int(k), int(n), float(p))

    def binomtest_batch(self, k, n, p, lanes=2):
        """scipy.stats.binomtest(k, n, p).pvalue for arrays of triples, on the device by the routine the deferred HWE
        tests use (trk_binomtest_batch; lanes=1: the serial routine in one lane)."""
        k = np.ascontiguousarray(k, dtype=np.int64)
        n = np.ascontiguousarray(n, dtype=np.int64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert k.shape == n.shape == p.shape and k.ndim == 1
        out = np.empty(k.shape, dtype=np.float64)
        self._chk(self.lib.trk_binomtest_batch(self.ctx, k.ctypes.data, n.ctypes.data, p.ctypes.data, k.size,
                                               out.ctypes.data, lanes))
        return out

    # ---- measurement aids ----
    def stream_probe(self, in0, in1, in2, out0, out1, n_loci, n_samples, reps=5):
        """Average launch time (ms) of the call-filter pass's bare stream shape on these planes (trk_stream_probe);
        out0 / out1 are overwritten."""
        ms = C.c_float()
        self._chk(self.lib.trk_stream_probe(self.ctx, in0.ptr, in1.ptr, in2.ptr, out0.ptr, out1.ptr, int(n_loci),
                                            int(n_samples), int(reps), C.byref(ms)))
        return float(ms.value)

    def device_clocks(self):
        """{'sclk_khz', 'mclk_khz', 'mem_bus_bits'} as the runtime reports them (peak values, not the live clocks)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self.lib.trk_device_clocks(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return dict(sclk_khz=a.value, mclk_khz=b.value, mem_bus_bits=c.value)

    # ---- multi-GPU ----
    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        rc = self.lib.trk_comm_unique_id(buf)
        if rc != 0:
            raise L.TrkError("trk_comm_unique_id: %s" % self.lib.trk_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, rank, n_ranks, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(self.lib.trk_comm_init(self.ctx, int(rank), int(n_ranks), buf))

    def allreduce_sum_i64(self, arr):
        self._chk(self.lib.trk_allreduce_sum_i64(self.ctx, arr.ptr, int(np.prod(arr.shape))))

    def allgather(self, send, recv):
        self._chk(self.lib.trk_allgather(self.ctx, send.ptr, recv.ptr, send.nbytes))

    def exchange(self, sums, send=None, recv=None):
        """trk_exchange: one grouped RCCL launch -- in-place int64 sum of ``sums`` over the ranks and the rank-major
        gather of ``send`` into ``recv``."""
        self._chk(self.lib.trk_exchange(self.ctx, sums.ptr if sums is not None else None,
                                        int(np.prod(sums.shape)) if sums is not None else 0,
                                        send.ptr if send is not None else None,
                                        recv.ptr if recv is not None else None,
                                        send.nbytes if send is not None else 0))

    # ---- synthetic batches ----
    def synth_fill(self, seed, n_loci, n_samples, allele_off_d, cdf_d, miss_d, inb_d, locus_base=0,
                   planes=('dp', 'q')):
        gt = self.empty((n_loci, n_samples, 2), np.int16)
        out = {'gt': gt}
        ptrs = {}
        for nm, dt in (('dp', np.int32), ('q', np.float32), ('dstutter', np.int32), ('dflankindel', np.int32)):
            if nm in planes:
                out[nm] = self.empty((n_loci, n_samples), dt)
                ptrs[nm] = out[nm].ptr
            else:
                ptrs[nm] = None
        spec = L.SynthSpec(int(seed), int(n_loci), int(n_samples), allele_off_d.ptr, cdf_d.ptr, miss_d.ptr,
                           inb_d.ptr, int(locus_base), 0)
        self._chk(self.lib.trk_synth_fill(self.ctx, C.byref(spec), gt.ptr, ptrs['dp'], ptrs['q'],
                                          ptrs['dstutter'], ptrs['dflankindel']))
        return out

    def synth_fill_gangstr(self, seed, n_loci, n_samples, allele_off_d, gt, dp, allele_repcn_d, locus_base=0):
        out = {'qexp': self.empty((n_loci, n_samples, 3), np.float32),
               'repcn': self.empty((n_loci, n_samples, 2), np.int32),
               'rc': self.empty((n_loci, n_samples, 4), np.int32),
               'repci': self.empty((n_loci, n_samples, 4), np.int32)}
        spec = L.SynthSpec(int(seed), int(n_loci), int(n_samples), allele_off_d.ptr, None, None, None,
                           int(locus_base), 0)
        self._chk(self.lib.trk_synth_fill_gangstr(self.ctx, C.byref(spec), gt.ptr, dp.ptr, allele_repcn_d.ptr,
                                                  out['qexp'].ptr, out['repcn'].ptr, out['rc'].ptr,
                                                  out['repci'].ptr))
        return out
