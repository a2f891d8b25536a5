"""The finaliser after the count pass of a big batch: sixteen lanes per locus (k_locus_finalize_coop) against one
thread per locus (TRK_FIN_COOP=0); HIP-event brackets of the library."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
for L, S in ((800, 10000), (1677, 5000), (3355, 5000), (6000, 2400)):
    sb = SynthBatch(eng, L, S, seed=20260931, planes=())
    res = eng.alloc_stats(sb.batch)
    for mode in ('0', '1', '0', '1'):
        os.environ['TRK_FIN_COOP'] = mode
        os.environ['TRK_FUSED_STATS'] = '0'
        eng.profile(True); eng.profile_reset()
        for _ in range(6): eng.locus_stats(sb.batch, out=res)
        eng.sync()
        pg = eng.profile_get(); eng.profile(False)
        print("%7d x %5d  coop=%s  " % (L, S, mode) + "  ".join("%s %.3f ms" % (k, ms / n) for k, (n, ms) in pg.items() if n), flush=True)
    for a in (res.allele_count, res.locus_int, res.locus_f64): a.free()
    for a in list(sb.dev.values()) + list(sb.batch.arrays.values()): a.free()
