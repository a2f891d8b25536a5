#!/usr/bin/env python3
"""Ablations of the count kernel on the GPU box (experiments only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
Lc, S = 100000, 10000
sb = SynthBatch(eng, Lc, S, seed=20260931, planes=())
cells = Lc * S
res = eng.alloc_stats(sb.batch)
eng.profile(True)
def t_count():
    eng.profile_reset()
    for _ in range(4): eng.locus_stats(sb.batch, out=res, count_only=True)
    eng.sync(); n, ms = eng.profile_get()['k_locus_count']; return ms / n
for u in (1, 2, 4):
    for mode in (0, 1, 2):
        os.environ['TRK_CNT_U'] = str(u); os.environ['TRK_CNT_MODE'] = str(mode)
        t = t_count(); print("count U=%d mode=%d: %.3f ms  %.0f GB/s" % (u, mode, t, cells * 4 / t / 1e6))
