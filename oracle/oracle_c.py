"""ctypes wrapper of oracle/oracle_c.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle_c.so')
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'oracle_c.c')):
            subprocess.check_call(['make', '-C', _HERE, '-s'])
        _lib = C.CDLL(_SO)
        _lib.orc_binomtest.restype = C.c_double
        _lib.orc_binomtest.argtypes = [C.c_int64, C.c_int64, C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _Plane(C.Structure):
    _fields_ = [('data', C.c_void_p), ('is_f32', C.c_int32), ('ncol', C.c_int32)]


class _Filter(C.Structure):
    _fields_ = [('op', C.c_int32), ('plane_a', C.c_int32), ('col_a', C.c_int32), ('plane_b', C.c_int32),
                ('col_b', C.c_int32), ('col_a2', C.c_int32), ('thr', C.c_double)]


def n_cores():
    return max(1, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))


_tuned = None


def tuned_threads():
    """The OpenMP team size at which the per-locus work runs fastest on this host (calibrated once on a small random
    call set): the visible core count is an upper bound only -- containers cap CPU time below it, and hyper-threads
    or far memory can make a full team slower than a partial one."""
    global _tuned
    if _tuned is None:
        import time
        rng = np.random.default_rng(0)
        Lc, S, A = 2048, 2000, 8
        gt = rng.integers(-1, A, size=(Lc, S, 2)).astype(np.int16)
        off = (np.arange(Lc + 1) * A).astype(np.int32)
        lc = np.tile(np.arange(A, dtype=np.uint16), Lc)
        cv = np.tile(np.arange(A, dtype=np.float64), Lc)
        cands = sorted({n for n in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256, n_cores()) if n <= n_cores()})
        best, best_t = 1, None
        for nt in cands:
            t = []
            outb = (np.zeros(Lc * A, dtype=np.int32), np.zeros((Lc, 8), dtype=np.int32), np.zeros((Lc, 10)))
            for _ in range(3):
                t0 = time.perf_counter()
                batch_stats(gt, None, off, lc, lc, cv, n_threads=nt, out=outb)
                t.append(time.perf_counter() - t0)
            if best_t is None or min(t) < best_t:
                best, best_t = nt, min(t)
        _tuned = best
    return _tuned


def batch_stats(gt, locus_ploidy, off, lc, sc, cv, n_threads=1, out=None):
    """``out``: optional preallocated (cnt int32 [sumA], oi int32 [L, 8], of float64 [L, 10]) -- the timing legs of
    bench.py keep numpy's allocation and page faults out of the measured region with it."""
    lib = load()
    gt = np.ascontiguousarray(gt, dtype=np.int16)
    Lc, S, P = gt.shape
    off = np.ascontiguousarray(off, dtype=np.int32)
    lc = np.ascontiguousarray(lc, dtype=np.uint16)
    sc = np.ascontiguousarray(sc, dtype=np.uint16)
    cv = np.ascontiguousarray(cv, dtype=np.float64)
    lp = None if locus_ploidy is None else np.ascontiguousarray(locus_ploidy, dtype=np.uint8)
    if out is not None:
        cnt, oi, of = out
    else:
        cnt = np.zeros(int(off[-1]), dtype=np.int32)
        oi = np.zeros((Lc, 8), dtype=np.int32)
        of = np.zeros((Lc, 10), dtype=np.float64)
    if n_threads > 1:
        lib.orc_batch_stats_mt(_p(gt), Lc, S, P, None if lp is None else _p(lp), _p(off), _p(lc), _p(sc), _p(cv),
                               _p(cnt), _p(oi), _p(of), int(n_threads))
    else:
        lib.orc_batch_stats(_p(gt), Lc, S, P, None if lp is None else _p(lp), _p(off), _p(lc), _p(sc), _p(cv),
                            _p(cnt), _p(oi), _p(of))
    return cnt, oi, of


def call_filters(gt, planes, filters, dp_plane=-1, locus_ploidy=None, n_threads=1, want_gt=True, want_mask=True,
                 out=None):
    """orc_call_filters: ``planes`` interleaved [L, S] / [L, S, k] int32 or float32 arrays, ``filters`` dicts with the
    keys of Engine.call_filters (op numbered as TRK_F_*).  Returns (gt_out | None, mask | None, counters [(1+nf), S],
    totaldp [S], dpmiss [S], err [4])."""
    lib = load()
    gt = np.ascontiguousarray(gt, dtype=np.int16)
    Lc, S, P = gt.shape
    keep = []
    parr = (_Plane * max(len(planes), 1))()
    for i, pl in enumerate(planes):
        pl = np.ascontiguousarray(pl)
        if pl.dtype not in (np.int32, np.float32):
            raise ValueError("plane dtype %s" % pl.dtype)
        keep.append(pl)
        parr[i] = _Plane(pl.ctypes.data, 1 if pl.dtype == np.float32 else 0, 1 if pl.ndim == 2 else pl.shape[2])
    nf = len(filters)
    farr = (_Filter * max(nf, 1))()
    for k, f in enumerate(filters):
        farr[k] = _Filter(int(f['op']), int(f['plane_a']), int(f.get('col_a', 0)), int(f.get('plane_b', -1)),
                          int(f.get('col_b', 0)), int(f.get('col_a2', 0)), float(f.get('thr', 0.0)))
    lp = None if locus_ploidy is None else np.ascontiguousarray(locus_ploidy, dtype=np.uint8)
    if out is not None:          # preallocated (gt_out, mask)
        gout, mask = out
    else:
        gout = np.empty_like(gt) if want_gt else None
        mask = np.zeros((Lc, S), dtype=np.uint32) if want_mask else None
    counters = np.zeros((1 + nf, S), dtype=np.int64)
    totaldp = np.zeros(S, dtype=np.int64)
    dpmiss = np.zeros(S, dtype=np.int64)
    err = np.zeros(4, dtype=np.int32)
    lib.orc_call_filters(_p(gt), Lc, S, P, None if lp is None else _p(lp), parr, len(planes), farr, nf, int(dp_plane),
                         None if gout is None else _p(gout), None if mask is None else _p(mask), _p(counters),
                         _p(totaldp), _p(dpmiss), _p(err), int(n_threads))
    return gout, mask, counters, totaldp, dpmiss, err


def call_filters_dpq(gt, dp, q, min_dp, max_dp, min_q):
    lib = load()
    gt = np.ascontiguousarray(gt, dtype=np.int16)
    dp = np.ascontiguousarray(dp, dtype=np.int32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    Lc, S, _ = gt.shape
    gout = np.empty_like(gt)
    mask = np.zeros((Lc, S), dtype=np.uint32)
    counters = np.zeros((4, S), dtype=np.int64)
    totaldp = np.zeros(S, dtype=np.int64)
    dpmiss = np.zeros(S, dtype=np.int64)
    lib.orc_call_filters_dpq(_p(gt), _p(dp), _p(q), Lc, S, C.c_double(min_dp), C.c_double(max_dp),
                             C.c_double(min_q), _p(gout), _p(mask), _p(counters), _p(totaldp), _p(dpmiss))
    return gout, mask, counters, totaldp, dpmiss


def assoc_scan(gt, off, allele_len, x, y, sample_in=None, non_major_cutoff=20.0, n_threads=1):
    """orc_assoc_scan_mt: the associaTR scan of a diploid batch.  x: design [S, M] (column 0 reserved for the genotype,
    column 1 the intercept, then standardised covariates), y: outcome [S].  Returns (out_i [L, 2] = n_tested, status;
    out_f [L, 4] = p, coef_std, se_std, R^2)."""
    lib = load()
    gt = np.ascontiguousarray(gt, dtype=np.int16)
    Lc, S, P = gt.shape
    assert P == 2
    off = np.ascontiguousarray(off, dtype=np.int32)
    al = np.ascontiguousarray(allele_len, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    assert x.shape[0] == S and y.shape == (S,)
    si = None if sample_in is None else np.ascontiguousarray(sample_in, dtype=np.uint8)
    oi = np.zeros((Lc, 2), dtype=np.int32)
    of = np.full((Lc, 4), np.nan)
    lib.orc_assoc_scan_mt.restype = None
    lib.orc_assoc_scan_mt(_p(gt), C.c_int(Lc), C.c_int(S), _p(off), _p(al), None if si is None else _p(si), _p(x),
                          C.c_int(x.shape[1]), _p(y), C.c_double(non_major_cutoff), _p(oi), _p(of), C.c_int(n_threads))
    return oi, of
