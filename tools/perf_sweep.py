#!/usr/bin/env python3
"""Sweep launch knobs of the two streaming kernels on the GPU box (experiments only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
from trtools_amd import _lib as L
eng = Engine(0)
Lc, S = 100000, 10000
sb = SynthBatch(eng, Lc, S, seed=20260931)
cells = Lc * S
res = eng.alloc_stats(sb.batch)
planes = [sb.dev['dp'], sb.dev['q']]
filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=1000), dict(op=L.F_LT, plane_a=1, thr=0.9)]
out = eng.alloc_call_out(sb.batch, len(filters))
eng.profile(True)
def t_count():
    eng.profile_reset()
    for _ in range(4): eng.locus_stats(sb.batch, out=res, count_only=True)
    eng.sync(); n, ms = eng.profile_get()['k_locus_count']; return ms / n
def t_cf():
    eng.profile_reset()
    for _ in range(4): eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out)
    eng.sync(); n, ms = eng.profile_get()['k_call_filter']; return ms / n
for u in (2, 4, 8):
    os.environ['TRK_CNT_U'] = str(u)
    t = t_count(); print("count U=%d: %.3f ms  %.0f GB/s" % (u, t, cells * 4 / t / 1e6))
os.environ.pop('TRK_CNT_U')
for u in (1, 2, 4):
    for lpb in (32, 64, 128, 256, 512, 1024):
        os.environ['TRK_CF_U'] = str(u); os.environ['TRK_CF_LPB'] = str(lpb)
        t = t_cf(); print("callfilter U=%d lpb=%4d: %.3f ms  %.0f GB/s" % (u, lpb, t, cells * 20 / t / 1e6))
