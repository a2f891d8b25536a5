#!/usr/bin/env python3
"""Are the 'classes' of the output-plane pair effect (r03_notes section 22) large contiguous regions of the device's
address space?  Planes of 4 GB with GAP GB of untouched spacer between them (the driver hands memory out in order), the
bare 3-in / 2-out stream over every pair: a block structure in the matrix would say yes."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 7
eng = Engine(0)
L, S = 100000, 10016
inp = eng.empty((L, S), np.uint32)
planes, spacers = [], []
for k in range(n):
    planes.append(eng.empty((L, S), np.uint32))
    if k + 1 < n and gap > 0:
        try:
            spacers.append(eng.empty((int(gap * (1 << 30)),), np.uint8))
        except Exception as e:
            print("spacer %d failed: %s" % (k, e)); break
print("planes at", ' '.join(hex(p.ptr) for p in planes))
print("pair matrix (ms), rows = masked-genotype plane, columns = mask plane; gap %.0f GB" % gap)
for i, a in enumerate(planes):
    row = []
    for j, b in enumerate(planes):
        row.append('   -  ' if i == j else '%6.3f' % eng.stream_probe(inp, inp, inp, a, b, L, S, reps=3))
    print(' '.join(row), flush=True)
eng.close()
