"""statSTR pass, fused (count + finaliser in one launch, HWE slots) against the chain, at several batch sizes.
TRK_FUSED_STATS=<max loci> moves the limit of the fused pass (0: never)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
for L, S in ((1000, 1000), (10000, 1000), (30000, 2000), (100000, 1000), (400000, 1000), (100000, 2048)):
    sb = SynthBatch(eng, L, S, seed=5, planes=())
    if os.environ.get('PAD') == '1':
        sb.pad_rows(32)
    res = eng.alloc_stats(sb.batch)
    row = []
    for mode in ('0', '100000000'):
        os.environ['TRK_FUSED_STATS'] = mode
        for it in range(23):
            if it == 3:
                eng.sync(); t0 = time.perf_counter()
            eng.locus_stats(sb.batch, out=res)
        eng.sync()
        row.append((time.perf_counter() - t0) / 20 * 1e3)
    print("%7d x %5d  chain %.4f ms  fused %.4f ms" % (L, S, row[0], row[1]), flush=True)
    for a in (res.allele_count, res.locus_int, res.locus_f64): a.free()
    for a in list(sb.dev.values()) + list(sb.batch.arrays.values()): a.free()
