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


def batch_stats(gt, locus_ploidy, off, lc, sc, cv):
    lib = load()
    gt = np.ascontiguousarray(gt, dtype=np.int16)
    Lc, S, P = gt.shape
    off = np.ascontiguousarray(off, dtype=np.int32)
    lc = np.ascontiguousarray(lc, dtype=np.uint16)
    sc = np.ascontiguousarray(sc, dtype=np.uint16)
    cv = np.ascontiguousarray(cv, dtype=np.float64)
    lp = None if locus_ploidy is None else np.ascontiguousarray(locus_ploidy, dtype=np.uint8)
    cnt = np.zeros(int(off[-1]), dtype=np.int32)
    oi = np.zeros((Lc, 8), dtype=np.int32)
    of = np.zeros((Lc, 10), dtype=np.float64)
    lib.orc_batch_stats(_p(gt), Lc, S, P, None if lp is None else _p(lp), _p(off), _p(lc), _p(sc), _p(cv),
                        _p(cnt), _p(oi), _p(of))
    return cnt, oi, of


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
