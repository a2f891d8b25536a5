#!/usr/bin/env python3
"""Ad-hoc kernel timing on the GPU box (not the bench contract)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch
from trtools_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--dump', action='store_true')
a = ap.parse_args()
eng = Engine(0)
t = time.time()
sb = SynthBatch(eng, a.loci, a.samples, seed=20260928 + 4)
print("synth build %.1fs  maxA=%d meanA=%.1f" % (time.time() - t, np.max(np.diff(sb.tables[0])), np.mean(np.diff(sb.tables[0]))))
cells = a.loci * a.samples
res = eng.alloc_stats(sb.batch)
eng.profile(True)
for it in range(a.iters + 1):
    if it == 1:
        eng.profile_reset()
    eng.locus_stats(sb.batch, out=res)
eng.sync()
for k, (n, ms) in eng.profile_get().items():
    if n:
        print("%-18s n=%d avg %.3f ms" % (k, n, ms / n))
n, ms = eng.profile_get()['k_locus_count']
print("k_locus_count: %.1f GB/s algorithmic (4 B/cell)  %.3e cells/s" % (cells * 4 / (ms / n * 1e-3) / 1e9, cells / (ms / n * 1e-3)))
# dumpSTR-style call filters
planes = [sb.dev['dp'], sb.dev['q']]
filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=60), dict(op=L.F_LT, plane_a=1, thr=0.9)]
out = eng.alloc_call_out(sb.batch, len(filters))
eng.profile_reset()
for it in range(a.iters):
    eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out)
eng.sync()
n, ms = eng.profile_get()['k_call_filter']
print("k_call_filter: avg %.3f ms  %.1f GB/s algorithmic (20 B/cell)" % (ms / n, cells * 20 / (ms / n * 1e-3) / 1e9))
