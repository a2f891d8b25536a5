import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
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
def t_cf(delta=True):
    eng.profile_reset()
    for _ in range(4): eng.call_filters(sb.batch, planes, F3, dp_plane=0, out=out, delta_stats=st if delta else None)
    eng.sync(); n, ms = eng.profile_get()['k_call_filter']; return ms / n
print("dedupe on, nt ld/st : %.3f ms" % t_cf())
os.environ['TRK_CF_NODEDUPE'] = '1'; print("dedupe off          : %.3f ms" % t_cf()); os.environ.pop('TRK_CF_NODEDUPE')
for mm in (1, 2, 3):
    os.environ['TRK_CF_MEM'] = str(mm); print("mem_mode=%d (1 plain ld, 2 plain st): %.3f ms" % (mm, t_cf()))
os.environ.pop('TRK_CF_MEM')
print("no delta            : %.3f ms" % t_cf(False))
