#!/usr/bin/env python3
"""Timing of trk_assoc_scan for growing designs on the GPU box (100k loci x 10k samples resident): one pass up to
31 trait columns, pairs of 15-row groups above (TRK_ASSOC_MAX_VEC_WIDE)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch, pack_assoc_tables
ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--m', type=int, nargs='*', default=[1, 15, 31, 32, 45, 62])
a = ap.parse_args()
eng = Engine(0)
sb = SynthBatch(eng, a.loci, a.samples, seed=20260928 + 7, planes=())
alen, rcls = pack_assoc_tables(sb.loci.allele_lens, 2)
alen_d, rcls_d = eng.upload(alen, np.float64), eng.upload(rcls, np.uint16)
rng = np.random.default_rng(5)
for M in a.m:
    v = rng.normal(size=(M, a.samples))
    v = (v - v.mean(axis=1, keepdims=True)) / v.std(axis=1, keepdims=True)
    vec_d = eng.upload(v, np.float64)
    res = eng.assoc_scan(sb.batch, vec_d, alen_d, rcls_d, non_major_cutoff=20.0)
    eng.sync()
    eng.timer_start(0)
    for _ in range(a.iters):
        res = eng.assoc_scan(sb.batch, vec_d, alen_d, rcls_d, non_major_cutoff=20.0, out=res)
    eng.timer_stop(0)
    ms = eng.timer_ms(0) / a.iters
    li = res.locus_int.get()
    print("M = %2d   %.2f ms per pass   regressed loci %d" % (M, ms, int((li[:, 1] == 0).sum())), flush=True)
    for d in (vec_d, res.locus_int, res.locus_f64, res.allele_count):
        d.free()
