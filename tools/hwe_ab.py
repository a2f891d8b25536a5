#!/usr/bin/env python3
"""configs[1] (statSTR 10k x 1k) finaliser + HWE test time; run once with TRK_HWE_SERIAL=1 and once without."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
eng.profile(True)
for L_, S in ((10000, 1000), (12500, 10000), (100000, 10000)):
    sb = SynthBatch(eng, L_, S, seed=20260928 + 1, planes=())
    res = eng.alloc_stats(sb.batch)
    for it in range(41):
        if it == 1:
            eng.sync(); eng.profile_reset(); t0 = time.perf_counter()
        eng.locus_stats(sb.batch, out=res)
    eng.sync(); w = (time.perf_counter() - t0) / 40
    pg = eng.profile_get()
    print("serial=%s  %d x %d: %.4f ms/pass; count %.4f ms, finalize+hwe %.4f ms" % (
        os.environ.get('TRK_HWE_SERIAL', '0'), L_, S, w * 1e3, pg['k_locus_count'][1] / pg['k_locus_count'][0],
        pg['k_locus_finalize'][1] / pg['k_locus_finalize'][0]), flush=True)
