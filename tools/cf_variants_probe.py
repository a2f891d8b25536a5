#!/usr/bin/env python3
"""The call-filter pass of the headline cohort (100k x 10k, HipSTR shape) in the builds that matter, a few launches
each -- the command tools/sq_counters.sh and rocprofv3 wrap to get SQ counters / durations per build:
  full      three filters, masked genotypes + 32-bit mask (20 B per call)         k_call_filter_v4<3,1,true,false,3,0>
  compact   three filters, the one-byte mask alone (13 B: dumpSTR's command line) k_call_filter_v4<3,1,true,false,3,2>
  hipstr5   five filters incl. the two ratio filters, full outputs (28 B)          k_call_filter_v4<5,1,true,true,3,0>
  hipstr5c  five filters, compact (21 B)                                          k_call_filter_v4<5,1,true,true,3,2>
usage: cf_variants_probe.py [--loci N] [--samples S] [--iters K] [which ...]"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
from trtools_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--iters', type=int, default=4)
ap.add_argument('which', nargs='*', default=['full', 'compact', 'compact_cv2', 'compact', 'compact_cv2', 'hipstr5', 'hipstr5c'])
a = ap.parse_args()
eng = Engine(0)
sb = SynthBatch(eng, a.loci, a.samples, seed=20260931, planes=('dp', 'q', 'dstutter', 'dflankindel'))
sb.pad_rows(32)
planes = [sb.dev['dp'], sb.dev['q'], sb.dev['dstutter'], sb.dev['dflankindel']]
f3 = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=60), dict(op=L.F_LT, plane_a=1, thr=0.9)]
f5 = f3 + [dict(op=L.F_RATIO_GT, plane_a=2, plane_b=0, thr=0.15), dict(op=L.F_RATIO_GT, plane_a=3, plane_b=0, thr=0.15)]
st0 = eng.locus_stats(sb.batch, count_only=True)
st = eng.alloc_stats(sb.batch)
cells = a.loci * a.samples
_o = eng.alloc_call_out(sb.batch, 3)
print("bare stream of the full pass on this box (k_stream_probe, reserved pair %s): %.3f ms" % (
    (type(eng).last_placement or {}).get('reserved'),
    eng.stream_probe(sb.dev['gt'], sb.dev['dp'], sb.dev['q'], _o.gt_out, _o.filter_mask, a.loci, sb.batch.n_samples, reps=3)), flush=True)
_o.gt_out.free(); _o.filter_mask.free()
for w in a.which:
    filters = f5 if w.startswith('hipstr5') else f3
    compact = w in ('compact', 'hipstr5c', 'compact_cv2')
    L.set_option('TRK_CF_CV2', '1' if w == 'compact_cv2' else None)
    out = eng.alloc_call_out(sb.batch, len(filters), want_gt=not compact, want_mask=not compact, want_mask8=compact)
    eng.profile(True)
    for it in range(a.iters + 1):
        if it == 1:
            eng.sync(); eng.profile_reset()
        st.allele_count.copy_from(st0.allele_count)
        st.locus_int.copy_from(st0.locus_int)
        eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st)
    eng.sync()
    n, ms = eng.profile_get()['k_call_filter']
    eng.profile(False)
    bpc = 4 + 4 * (2 if len(filters) == 3 else 4) + (1 if compact else 8)
    print("%-11s %d filters, %2d B per call: %.3f ms = %.0f GB/s = %.3f of 8 TB/s%s" %
          (w, len(filters), bpc, ms / n, cells * bpc / (ms / n * 1e-3) / 1e9, cells * bpc / (ms / n * 1e-3) / 8e12,
           '' if compact else '   placement %s' % (type(eng).last_placement,)), flush=True)
    for x in (out.gt_out, out.filter_mask, out.filter_mask8):
        if x is not None:
            x.free()
eng.close()
