"""After another process has just released its memory, do fresh allocations all land in one class -- and does a large
spacer allocation move the next ones elsewhere?  Planes allocated one at a time; for each, its probe time paired with
every earlier plane ('s' = slow pair, 'f' = fast pair); after plane 5 a spacer of SPACER_GB is allocated and held."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
os.environ['TRK_POOL_GB'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
eng = Engine(0)
L, S = 100000, 10016
ins = [eng.empty((L, S), np.uint32) for _ in range(3)]
for a in ins: a.zero()
planes, spacers = [], []
lo = None
for k in range(10):
    if k == 6:
        gb = int(os.environ.get('SPACER_GB', '40'))
        spacers.append(eng.empty((gb << 28,), np.uint32))
        print("   -- spacer of %d GB at %x" % (gb, spacers[-1].ptr))
    p = eng.empty((L, S), np.uint32)
    ts = [eng.stream_probe(ins[0], ins[1], ins[2], q, p, L, S, reps=3) for q in planes]
    planes.append(p)
    if ts:
        lo = min(ts) if lo is None else min(lo, min(ts))
    print("plane %d at %x: %s" % (k, p.ptr, " ".join("%.2f" % t for t in ts)), flush=True)
