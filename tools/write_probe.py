"""How fast does a 150 MB block reach a file on this box: one write(), pwrite() by several threads, memcpy into an mmap of
the file by several threads.  (dumpSTR's writer thread: profiles/r05_notes.md section 8.)"""
import ctypes as C
import mmap
import os
import sys
import threading
import time

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/e2e/wprobe.bin'
N, B = 150_000_000, 8
blk = np.random.randint(32, 120, N, dtype=np.uint8)
mv = memoryview(blk)


def run(name, fn):
    if os.path.exists(path):
        os.remove(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    ts = []
    for i in range(B):
        t = time.perf_counter()
        fn(fd, i * N)
        ts.append((time.perf_counter() - t) * 1e3)
    t = time.perf_counter()
    os.close(fd)
    print('%-28s per block ms: %s  (close %.1f)' % (name, ' '.join('%.1f' % x for x in ts), (time.perf_counter() - t) * 1e3), flush=True)
    os.remove(path)


def one_write(fd, off):
    os.pwrite(fd, mv, off)


def par_pwrite(nt):
    def fn(fd, off):
        step = (N + nt - 1) // nt
        th = [threading.Thread(target=lambda k=k: os.pwrite(fd, mv[k * step:min(N, (k + 1) * step)], off + k * step)) for k in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
    return fn


def par_mmap(nt):
    def fn(fd, off):
        os.ftruncate(fd, off + N)
        pg = off - off % mmap.ALLOCATIONGRANULARITY
        m = mmap.mmap(fd, off + N - pg, offset=pg)
        base = C.addressof(C.c_char.from_buffer(m)) + (off - pg)
        step = (N + nt - 1) // nt
        src = blk.ctypes.data
        th = [threading.Thread(target=lambda k=k: C.memmove(base + k * step, src + k * step, min(step, N - k * step))) for k in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
        del base
        m.close()
    return fn


run('one write', one_write)
for nt in (2, 4, 8):
    run('pwrite x %d threads' % nt, par_pwrite(nt))
for nt in (1, 4, 8, 16):
    run('mmap + memcpy x %d threads' % nt, par_mmap(nt))
run('one write', one_write)
