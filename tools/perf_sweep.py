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
planes = [sb.dev['dp'], sb.dev['q']]
F3 = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=1000), dict(op=L.F_LT, plane_a=1, thr=0.9)]
st = eng.locus_stats(sb.batch, count_only=True)
eng.profile(True)
out = eng.alloc_call_out(sb.batch, 3)
def t_cf(delta):
    eng.profile_reset()
    for _ in range(3): eng.call_filters(sb.batch, planes, F3, dp_plane=0, out=out, delta_stats=st if delta else None)
    eng.sync(); n, ms = eng.profile_get()['k_call_filter']; return ms / n
for lpb in (128, 256, 512, 1024):
    for sub in (32, 64, 90):
        os.environ['TRK_CF_LPB'] = str(lpb); os.environ['TRK_CF_SUB'] = str(sub)
        t = t_cf(True)
        print("delta lpb=%4d sub=%3d: %.3f ms  %.0f GB/s(20B)" % (lpb, sub, t, cells * 20 / t / 1e6))
