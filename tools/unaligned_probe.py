#!/usr/bin/env python3
"""Cohort sizes that are not a multiple of four samples (rows not 16-byte aligned): time of the statSTR pass, the dumpSTR
call-filter pass and the one-trait association scan at S = 10000 vs S = 9999 / 10001 / 10002."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
from trtools_amd import _lib as L
eng = Engine(0)
eng.profile(True)
Lc = int(os.environ.get('L', 20000))
for S in (10000, 9999, 10001, 10002):
    sb = SynthBatch(eng, Lc, S, seed=5, planes=('dp', 'q'))
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=1000), dict(op=L.F_LT, plane_a=1, thr=0.9)]
    planes = [sb.dev['dp'], sb.dev['q']]
    st = eng.alloc_stats(sb.batch)
    out = eng.alloc_call_out(sb.batch, 3)
    for it in range(4):
        if it == 1:
            eng.sync(); eng.profile_reset()
        eng.locus_stats(sb.batch, out=st, count_only=True)
        eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st)
        eng.locus_finalize(sb.batch, st)
    eng.sync()
    pg = eng.profile_get()
    c = pg['k_locus_count'][1] / pg['k_locus_count'][0]
    f = pg['k_call_filter'][1] / pg['k_call_filter'][0]
    print("S=%5d: count %.3f ms (%.0f GB/s), call filter %.3f ms (%.0f GB/s)" % (
        S, c, Lc * S * 4 / c / 1e6, f, Lc * S * 20 / f / 1e6))
    del sb, st, out, planes
