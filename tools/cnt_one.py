"""k_locus_count on one shape: `python tools/cnt_one.py [loci samples]` (TRK_CNT_U / TRK_CNT_R / TRK_LIBTRK select
variants); also the command tools/sq_counters.sh wraps for the count kernel's instruction counters."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
L, S = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
sb = SynthBatch(eng, L, S, seed=20260931, planes=())
res = eng.alloc_stats(sb.batch)
eng.profile(True)
for it in range(8):
    if it == 3:
        eng.sync(); eng.profile_reset()
    eng.locus_stats(sb.batch, out=res, count_only=True)
eng.sync()
k, ms = eng.profile_get()['k_locus_count']
print("%dx%d U=%s R=%s: %.4f ms  %.0f GB/s" % (L, S, os.environ.get('TRK_CNT_U', '-'), os.environ.get('TRK_CNT_R', '-'), ms / k, L * S * 4 / (ms / k) / 1e6), flush=True)
