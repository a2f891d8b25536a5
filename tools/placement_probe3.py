"""Is the slow level a property of ONE output plane or of the PAIR?  Six planes, every ordered pair as (gt_out, mask)."""
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
P = [eng.empty((L, S), np.uint32) for _ in range(6)]
print("addresses:", ["%x" % p.ptr for p in P])
print("      " + "  ".join("mask%d" % j for j in range(6)))
for i in range(6):
    row = []
    for j in range(6):
        row.append("  -  " if i == j else "%.2f" % eng.stream_probe(ins[0], ins[1], ins[2], P[i], P[j], L, S, reps=3))
    print("gt%d   " % i + "   ".join(row), flush=True)
