#!/usr/bin/env python3
"""Ad-hoc timing of the associaTR scan on the GPU box (not the bench contract):
100k loci x 10k samples (BASELINE configs[4] shape), M = 1 outcome + covariates."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch, pack_assoc_tables

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--vecs', type=str, default='1,2,4,8')
ap.add_argument('--subset', action='store_true')
ap.add_argument('--opt', action='append', default=[], help='library option KEY=VALUE (include/trk_test.h)')
a = ap.parse_args()
eng = Engine(0)
from trtools_amd import _lib as _L
for kv in a.opt:
    _L.set_option(*kv.split('=', 1))
sb = SynthBatch(eng, a.loci, a.samples, seed=20260928 + 5, planes=())
alen, rcls = pack_assoc_tables(sb.loci.allele_lens, 2)
alen_d, rcls_d = eng.upload(alen, np.float64), eng.upload(rcls, np.uint16)
cells = a.loci * a.samples
rng = np.random.default_rng(5)
sin = None
if a.subset:
    sin = (rng.random(a.samples) < 0.9).astype(np.uint8)
eng.profile(True)
for M in [int(x) for x in a.vecs.split(',')]:
    vec = rng.normal(size=(M, a.samples))
    vec -= vec.mean(axis=1, keepdims=True)
    vec /= vec.std(axis=1, keepdims=True)
    vec_d = eng.upload(vec, np.float64)
    res = None
    for it in range(a.iters + 1):
        if it == 1:
            eng.profile_reset()
            eng.sync(); t0 = time.time()
        res = eng.assoc_scan(sb.batch, vec_d, alen_d, rcls_d, sample_in=sin, non_major_cutoff=20.0, out=res)
    eng.sync(); wall = (time.time() - t0) / a.iters
    pg = eng.profile_get()
    n, ms = pg['k_assoc_scan']; nf, msf = pg['k_assoc_finalize']
    st = res.locus_int.get()[:, 1]
    print("M=%2d  scan %.3f ms = %.0f GB/s (4 B/call)  finalize %.3f ms  wall %.3f ms  %.2e loci/s  tested %d/%d" % (
        M, ms / n, cells * 4 / (ms / n * 1e-3) / 1e9, msf / nf, wall * 1e3, a.loci / wall, int((st == 0).sum()), a.loci),
        flush=True)
