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
