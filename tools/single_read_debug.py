"""Debug aid: single-read step (trk_call_out.count_*) against count-then-filter on one of the test's cases."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_callfilters as T
from trtools_amd.engine import Engine
from trtools_amd import _lib as L
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng = Engine(0)
b, planes, Lc = T._single_read_case(eng, seed)
filters = [dict(op=L.F_LT, plane_a=0, thr=20), dict(op=L.F_GT, plane_a=0, thr=50), dict(op=L.F_LT, plane_a=1, thr=0.9)]
st_gt = eng.locus_stats(b, count_only=True)
st_ref = eng.locus_stats(b, count_only=True)
ref = eng.call_filters(b, planes, filters, dp_plane=0, out=eng.alloc_call_out(b, 3, place=False), delta_stats=st_ref)
st_cnt, st_dl = eng.alloc_stats(b), eng.alloc_stats(b)
res = eng.call_filters(b, planes, filters, dp_plane=0, out=eng.alloc_call_out(b, 3, place=False), delta_stats=st_dl, count_stats=st_cnt)
off = b.arrays['allele_off'].get()
for name, got, want in (('count', st_cnt, st_gt), ('delta', st_dl, st_ref)):
    a, r = got.allele_count.get()[0], want.allele_count.get()[0]
    bad = np.flatnonzero(a != r)
    print(name, 'allele_count diffs', len(bad), 'of', len(a), 'sum got', a.sum(), 'want', r.sum())
    for l in range(min(3, Lc)):
        print('  locus', l, 'got', a[off[l]:off[l + 1]], 'want', r[off[l]:off[l + 1]])
    a, r = got.locus_int.get()[0], want.locus_int.get()[0]
    bad = np.argwhere(a != r)
    print(name, 'locus_int diffs', len(bad), 'cols', np.unique(bad[:, 1]) if len(bad) else [])
    for l, c in bad[:5]:
        print('  ', l, c, a[l], r[l])
