#!/usr/bin/env python3
"""Timing of trk_qc_reduce (qcSTR's reductions) on the GPU box: 100k loci x 10k samples, genotypes + quality plane
resident (8 B per call)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--ab', action='store_true', help="round 6: k_qc_scan4 against k_qc_scan<4> (library option TRK_QC_OLD), outputs compared bit for bit")
a = ap.parse_args()
eng = Engine(0)
sb = SynthBatch(eng, a.loci, a.samples, seed=20260928 + 7, planes=('q',))
cells = a.loci * a.samples
for label, q, ign in (('calls only (4 B/call)', None, False), ('calls + quality (8 B/call)', sb.dev['q'], False),
                      ('calls + quality, ignore no-calls', sb.dev['q'], True)):
    res = eng.qc_reduce(sb.batch, q, None, ign)
    eng.sync()
    eng.timer_start(0)
    for _ in range(a.iters):
        res = eng.qc_reduce(sb.batch, q, None, ign)
    eng.timer_stop(0)
    ms = eng.timer_ms(0) / a.iters
    bpc = 4 if q is None else 8
    print("%-36s %.3f ms = %.0f GB/s  (calls %d)" % (label, ms, cells * bpc / ms / 1e6, int(res['sample_calls'].get().sum())),
          flush=True)

if a.ab:
    from trtools_amd import _lib as L
    for wgcu in (2, 4, 6, 8, 12, 16):
        L.set_option('TRK_QC_WGCU', str(wgcu))
        res = eng.qc_reduce(sb.batch, sb.dev['q'], None, False)
        eng.sync()
        eng.timer_start(0)
        for _ in range(a.iters):
            res = eng.qc_reduce(sb.batch, sb.dev['q'], None, False)
        eng.timer_stop(0)
        print("  workgroups per CU %2d: calls + quality %.3f ms" % (wgcu, eng.timer_ms(0) / a.iters), flush=True)
    L.set_option('TRK_QC_WGCU', None)
    keys = ('sample_calls', 'sample_qual_n', 'sample_qual_sum', 'locus_calls', 'locus_qual_n', 'locus_qual_sum')
    for label, q, ign in (('calls only', None, False), ('calls + quality', sb.dev['q'], False), ('ignore no-calls', sb.dev['q'], True)):
        got = {}
        for tag, opt in (('k_qc_scan4', None), ('k_qc_scan<4>', '1'), ('k_qc_scan4 ', None)):
            L.set_option('TRK_QC_OLD', opt)
            res = eng.qc_reduce(sb.batch, q, None, ign)
            eng.sync()
            eng.timer_start(0)
            for _ in range(a.iters):
                res = eng.qc_reduce(sb.batch, q, None, ign)
            eng.timer_stop(0)
            got[tag] = {k: res[k].get() for k in keys if res.get(k) is not None}
            print("  %-18s %-12s %.3f ms" % (label, tag, eng.timer_ms(0) / a.iters), flush=True)
        L.set_option('TRK_QC_OLD', None)
        for k in got['k_qc_scan4']:
            x, y = got['k_qc_scan4'][k], got['k_qc_scan<4>'][k]
            assert x.tobytes() == y.tobytes(), (label, k)
        print("  %-18s outputs identical (%s)" % (label, ', '.join(got['k_qc_scan4'])), flush=True)
