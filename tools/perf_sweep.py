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
F2 = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_LT, plane_a=1, thr=0.9)]
st = eng.locus_stats(sb.batch, count_only=True)
eng.profile(True)
def t_cf(filters, delta, out):
    eng.profile_reset()
    for _ in range(3): eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st if delta else None)
    eng.sync(); n, ms = eng.profile_get()['k_call_filter']; return ms / n
for name, F in (('3 filters', F3), ('2 filters', F2)):
    out = eng.alloc_call_out(sb.batch, len(F))
    for delta in (False, True):
        for lpb in (32, 64, 118, 256):
            if delta and lpb > 118: continue
            os.environ['TRK_CF_LPB'] = str(lpb)
            t = t_cf(F, delta, out)
            print("%s delta=%d lpb=%3d: %.3f ms  %.0f GB/s(20B)" % (name, delta, lpb, t, cells * 20 / t / 1e6))
os.environ.pop('TRK_CF_LPB')
os.environ['TRK_CF_GENERIC'] = '1'
out = eng.alloc_call_out(sb.batch, 3)
print("generic kernel 3 filters no delta: %.3f ms" % t_cf(F3, False, out))
