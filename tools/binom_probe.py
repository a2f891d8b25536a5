#!/usr/bin/env python3
"""Kernel time of the exact binomial test on the device, one lane per test against the lane pair, by input class
(run under rocprofv3 --kernel-trace; tools/binom_probe.py prints the call order, the trace holds the durations)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(3)
C = 10000
for name, n_, width in (('n=1000 k within 1 sd', 1000, 1.0), ('n=1000 k at 3 sd', 1000, 3.0), ('n=10000 k within 1 sd', 10000, 1.0),
                        ('n=10000 k at 6 sd', 10000, 6.0)):
    n = np.full(C, n_, dtype=np.int64)
    p = rng.uniform(0.2, 0.6, C)
    sd = np.sqrt(n * p * (1 - p))
    z = rng.normal(size=C) if width == 1.0 else rng.choice([-1, 1], C) * width
    k = np.clip(np.rint(n * p + z * sd), 0, n).astype(np.int64)
    for lanes in (1, 2, 1, 2):
        out = eng.binomtest_batch(k, n, p, lanes=lanes)
    print(name, float(out.mean()), flush=True)
