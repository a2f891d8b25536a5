"""k_locus_count on three shapes (100k x 10k, 12.5k x 10k, 400k x 1k at 1 and 4 loci per wave); run once per build with
TRK_LIBTRK=<other libtrk.so> for a same-box A/B.  `gpurun -- python tools/cnt_ab.py`."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
for (L, S) in ((100000, 10000), (12500, 10000), (400000, 1000)):
    sb = SynthBatch(eng, L, S, seed=20260931, planes=())
    res = eng.alloc_stats(sb.batch)
    for R in (('1',) if S > 4096 else ('1', '4')):
        os.environ['TRK_CNT_R'] = R
        eng.profile(True)
        for it in range(13):
            if it == 3:
                eng.sync(); eng.profile_reset()
            eng.locus_stats(sb.batch, out=res, count_only=True)
        eng.sync()
        k, ms = eng.profile_get()['k_locus_count']
        print("%s %dx%d R=%s: %.4f ms  %.0f GB/s" % (os.environ.get('TRK_LIBTRK', 'new'), L, S, R, ms / k, L * S * 4 / (ms / k) / 1e6), flush=True)
    sb = None
