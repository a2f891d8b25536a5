"""Does the bare call-filter stream depend on WHICH allocation the planes landed in?  One process: five 4 GB planes
allocated, probed, then (mode free) freed and allocated again or (mode keep) kept while the next set is allocated.
TRK_POOL_GB=0 so that a free really returns the memory."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
os.environ['TRK_POOL_GB'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
eng = Engine(0)
L, S = 100000, 10016
for mode in ('free', 'keep', 'free'):
    kept = []
    for it in range(6):
        arrs = [eng.empty((L, S), np.uint32) for _ in range(5)]
        for a in arrs[:3]:
            a.zero()
        eng.sync()
        ms = eng.stream_probe(arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], L, S, reps=5)
        print("%s %d: %.3f ms  (lowest address %x)" % (mode, it, ms, min(a.ptr for a in arrs)), flush=True)
        if mode == 'free':
            for a in arrs: a.free()
        else:
            kept += arrs
    for a in kept: a.free()
    eng.sync()
