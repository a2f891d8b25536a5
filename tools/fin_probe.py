import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
sb = SynthBatch(eng, 100000, 10000, seed=20260931, planes=())
res = eng.alloc_stats(sb.batch)
eng.profile(True); eng.profile_reset()
for _ in range(4): eng.locus_stats(sb.batch, out=res)
eng.sync()
for k, (n, ms) in eng.profile_get().items():
    if n: print("%-18s avg %.3f ms" % (k, ms / n))
