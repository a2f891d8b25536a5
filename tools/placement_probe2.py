"""Which planes' placement decides the bare stream's time?  (a) inputs fixed, outputs re-allocated; (b) outputs fixed,
inputs re-allocated; with each allocation's memset time (a write-only stream over it) beside the probe."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
os.environ['TRK_POOL_GB'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
eng = Engine(0)
L, S = 100000, 10016
def fill_ms(a, reps=3):
    a.zero(); eng.sync()
    eng.timer_start(0)
    for _ in range(reps): a.zero()
    eng.timer_stop(0)
    return eng.timer_ms(0) / reps
ins = [eng.empty((L, S), np.uint32) for _ in range(3)]
for a in ins: a.zero()
hold = []
print("(a) inputs fixed, outputs new each time (old ones kept allocated)")
for it in range(8):
    outs = [eng.empty((L, S), np.uint32) for _ in range(2)]
    f = [fill_ms(o) for o in outs]
    ms = eng.stream_probe(ins[0], ins[1], ins[2], outs[0], outs[1], L, S, reps=5)
    print("  outs %d: probe %.3f ms   memset of the two outputs %.3f / %.3f ms" % (it, ms, f[0], f[1]), flush=True)
    hold += outs
best_outs = hold[:2]
print("(b) outputs fixed (first pair), inputs new each time")
for it in range(5):
    ins2 = [eng.empty((L, S), np.uint32) for _ in range(3)]
    for a in ins2: a.zero()
    ms = eng.stream_probe(ins2[0], ins2[1], ins2[2], best_outs[0], best_outs[1], L, S, reps=5)
    print("  ins %d: probe %.3f ms" % (it, ms), flush=True)
    hold += ins2
