import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd import synth, _lib as TL
import test_gpu_assoc as T
eng = Engine(0)
lens, gt, lp, traits, keep = T.make_case(1, 120, 512, 2, 1, False)
off, lc, sc, cv = synth.pack_alleles(lens, None)
alen, rcls = synth.pack_assoc_tables(lens, 2)
vec = (traits[:, :1].T - traits[:, 0].mean()) / traits[:, 0].std()
b = eng.make_batch(gt, off, lc, sc, cv)
r = eng.assoc_scan(b, vec, alen, rcls, non_major_cutoff=3.0)
li = r.locus_int.get(); cnt = r.allele_count.get()
for l in range(8):
    g = gt[l]
    called = ~np.any(g == -1, axis=1)
    sel = g[called]
    exp = np.bincount(sel[sel >= 0].astype(int), minlength=len(lens[l]))
    print(l, 'dev n', li[l, 0], 'exp n', called.sum(), 'A', len(lens[l]), 'dev cnt', cnt[off[l]:off[l+1]], 'exp', exp,
          'n -2 rows', int(np.any(g == -2, axis=1).sum()), 'partial', int((np.any(g == -1, axis=1) & ~np.all(g == -1, axis=1)).sum()))
