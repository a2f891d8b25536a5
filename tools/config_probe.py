#!/usr/bin/env python3
"""Timing of BASELINE configs[1] (statSTR, 10k x 1k) and configs[2] (dumpSTR, GangSTR shape, 50k x 5k, nine call
filters + four locus filters) on the GPU box, inputs resident; kernel times from the library's HIP-event brackets."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
from trtools_amd import _lib as L
eng = Engine(0)
eng.profile(True)
# ---- configs[1] ----
sb = SynthBatch(eng, 10000, 1000, seed=20260928 + 1, planes=())
res = eng.alloc_stats(sb.batch)
for it in range(21):
    if it == 1:
        eng.sync(); eng.profile_reset(); t0 = time.perf_counter()
    eng.locus_stats(sb.batch, out=res)
eng.sync(); w = (time.perf_counter() - t0) / 20
pg = eng.profile_get()
print("configs[1] statSTR 10k x 1k: %.3f ms/pass = %.2e loci/s (%.2e calls/s); count %.3f ms, finalize+hwe %.3f ms" % (
    w * 1e3, 10000 / w, 1e7 / w, pg['k_locus_count'][1] / pg['k_locus_count'][0], pg['k_locus_finalize'][1] / pg['k_locus_finalize'][0]))
# ---- configs[2] ----
Lc, S = 50000, 5000
sb = SynthBatch(eng, Lc, S, seed=20260928 + 2, planes=('dp', 'q'), pure_repeats=True)
sb.add_gangstr_planes()
if os.environ.get('TRK_C2_PAD', '1') != '0':    # rows on 128-byte boundaries, as compute.DeviceCompute uploads a cohort
    sb.pad_rows(32)
planes = [sb.dev['dp'], sb.dev['q'], sb.dev['qexp'], sb.dev['rc'], sb.dev['repcn'], sb.dev['repci']]
filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=60), dict(op=L.F_LT, plane_a=1, thr=0.9),
           dict(op=L.F_CALLED_LT, plane_a=2, col_a=1, thr=0.05), dict(op=L.F_CALLED_LT, plane_a=2, col_a=2, thr=0.05),
           dict(op=L.F_CALLED_SUM_LT, plane_a=2, col_a=1, col_a2=2, thr=0.2),
           dict(op=L.F_CALLED_EQ, plane_a=3, col_a=1, plane_b=0, col_b=0),
           dict(op=L.F_CALLED_SUM_EQ, plane_a=3, col_a=1, col_a2=3, plane_b=0, col_b=0),
           dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=4, plane_b=5)]
bpc = 4 + 4 + 4 + 12 + 16 + 8 + 16 + 8
ref_bits = None
for layout in ('interleaved', 'planar'):
    if layout == 'planar':
        planes = [eng.planarize(p) for p in planes]
    st = eng.alloc_stats(sb.batch)
    out = eng.alloc_call_out(sb.batch, len(filters))
    for it in range(6):
        if it == 1:
            eng.sync(); eng.profile_reset(); t0 = time.perf_counter()
        eng.locus_stats(sb.batch, out=st, count_only=True)
        eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st)
        eng.locus_finalize(sb.batch, st)
        eng.locus_filters(Lc, st, min_callrate=0.8, min_hwep=1e-3, min_het=0.05, max_het=0.9)
    eng.sync(); w = (time.perf_counter() - t0) / 5
    pg = eng.profile_get()
    cf = pg['k_call_filter'][1] / pg['k_call_filter'][0]
    print("configs[2] dumpSTR GangSTR 50k x 5k [%s]: %.3f ms/pass = %.2e loci/s (%.2e calls/s); call filter %.3f ms = %.0f GB/s "
          "(%d B/call), count %.3f ms, finalize+hwe %.3f ms" % (layout, w * 1e3, Lc / w, Lc * S / w, cf,
          Lc * S * (bpc if layout == 'interleaved' else 60) / (cf * 1e-3) / 1e9, bpc if layout == 'interleaved' else 60, pg['k_locus_count'][1] / pg['k_locus_count'][0],
          pg['k_locus_finalize'][1] / pg['k_locus_finalize'][0]))
    got = [out.sample_counters.get(), out.filter_mask.get(), out.gt_out.get(), st.locus_int.get()]
    if ref_bits is None:
        ref_bits = got
    else:
        print("planar == interleaved:", all(np.array_equal(a, b) for a, b in zip(ref_bits, got)))
# ---- HipSTR five-filter set (flank indel, stutter, min/max DP, min Q: dumpSTR.py:792-804) at 100k x 10k ----
del sb, planes, st, out
Lc, S = 100000, 10000
sb = SynthBatch(eng, Lc, S, seed=20260928 + 3, planes=('dp', 'q', 'dstutter', 'dflankindel'))
if os.environ.get('TRK_C2_PAD', '1') != '0':
    sb.pad_rows(32)
planes = [sb.dev['dp'], sb.dev['q'], sb.dev['dstutter'], sb.dev['dflankindel']]
filters = [dict(op=L.F_RATIO_GT, plane_a=3, plane_b=0, thr=0.15), dict(op=L.F_RATIO_GT, plane_a=2, plane_b=0, thr=0.15),
           dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=55), dict(op=L.F_LT, plane_a=1, thr=0.9)]
if os.environ.get('PROBE_NORATIO'):   # experiment: what the two float64 divisions cost
    filters[0] = dict(op=L.F_GT, plane_a=3, thr=4)
    filters[1] = dict(op=L.F_GT, plane_a=2, thr=4)
st = eng.alloc_stats(sb.batch)
out = eng.alloc_call_out(sb.batch, len(filters))
for it in range(6):
    if it == 1:
        eng.sync(); eng.profile_reset(); t0 = time.perf_counter()
    eng.locus_stats(sb.batch, out=st, count_only=True)
    eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st)
    eng.locus_finalize(sb.batch, st)
    eng.locus_filters(Lc, st, min_callrate=0.8, min_hwep=1e-3, min_het=0.05, max_het=0.9)
eng.sync(); w = (time.perf_counter() - t0) / 5
pg = eng.profile_get()
cf = pg['k_call_filter'][1] / pg['k_call_filter'][0]
bpc = 4 + 16 + 8
print("HipSTR 5 call filters 100k x 10k: %.3f ms/pass = %.2e loci/s; call filter %.3f ms = %.0f GB/s (%d B/call), count %.3f ms, "
      "finalize+hwe %.3f ms" % (w * 1e3, Lc / w, cf, Lc * S * bpc / (cf * 1e-3) / 1e9, bpc,
      pg['k_locus_count'][1] / pg['k_locus_count'][0], pg['k_locus_finalize'][1] / pg['k_locus_finalize'][0]))
