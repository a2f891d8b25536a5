#!/usr/bin/env python3
"""statSTR --samples a,b (sample groups, SURVEY 8d "stratified variant") at 100k x 10k: time of the count and
finalise kernels with G = 1, 2, 3 group masks (2: disjoint 40 % / 60 %; 3: the reference's layout for two sample
lists = the two lists plus everything)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
eng = Engine(0)
eng.profile(True)
Lc, S = int(os.environ.get('L', 100000)), int(os.environ.get('S', 10000))
sb = SynthBatch(eng, Lc, S, seed=20260928 + 3, planes=())
rng = np.random.default_rng(3)
a = rng.random(S) < 0.4
g8 = (np.uint8(1) << rng.integers(0, 8, size=S).astype(np.uint8)).astype(np.uint8)        # eight disjoint groups
o5 = rng.integers(1, 32, size=S).astype(np.uint8)                                           # five overlapping groups
cases = [(1, None, 'one group'), (2, (a * 1 + (~a) * 2).astype(np.uint8), '2 disjoint'),
         (3, (a * 1 + (~a) * 2 + 4).astype(np.uint8), '2 lists + everyone'), (8, g8, '8 disjoint'), (5, o5, '5 overlapping')]
for G, gb, what in cases:
  for sort in ((False,) if gb is None else (False, True)):
    if sort:
        eng.sync(); t0 = time.perf_counter()
        b = sb.batch.sorted_by_class(eng, gb, G)
        eng.sync(); tsort = time.perf_counter() - t0
    else:
        b = sb.batch if gb is None else sb.batch.with_groups(eng, gb, G)
        tsort = 0.0
    if not sort and G > 3 and Lc * S > 2e8:
        print("G=%d (%s) per-call group kernel: skipped at this size (19-23 ms at 100k x 10k, r01 notes)" % (G, what)); continue
    res = eng.alloc_stats(b)
    for it in range(6):
        if it == 1:
            eng.sync(); eng.profile_reset(); t0 = time.perf_counter()
        eng.locus_stats(b, out=res)
    eng.sync(); w = (time.perf_counter() - t0) / 5
    pg = eng.profile_get()
    c = pg['k_locus_count'][1] / pg['k_locus_count'][0]
    f = pg['k_locus_finalize'][1] / pg['k_locus_finalize'][0]
    print("G=%d (%s)%s: %.3f ms/pass = %.2e loci/s; count %.3f ms = %.0f GB/s (%.2f of 8 TB/s), finalize+hwe %.3f ms%s" % (
        G, what, " class-sorted columns" if sort else "", w * 1e3, Lc / w, c, Lc * S * 4 / (c * 1e-3) / 1e9,
        Lc * S * 4 / (c * 1e-3) / 8e12, f, ("; one-off device gather %.2f ms" % (tsort * 1e3)) if sort else ""), flush=True)
    for x in (res.allele_count, res.locus_int, res.locus_f64):
        x.free()
    if sort:
        b.arrays['gt'].free()
