#!/usr/bin/env python3
"""Ad-hoc timing of trk_assoc_scan_dosage on the GPU box: synthetic AP1/AP2 planes over the bench's call set."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch, pack_assoc_tables, pack_dosage_tables

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=20000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--max-alleles', type=int, default=2)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--vecs', type=str, default='1,4')
a = ap.parse_args()
eng = Engine(0)
from trtools_amd import synth
loci = synth.make_loci(a.loci, a.samples, seed=7, max_alleles=a.max_alleles)
sb = SynthBatch(eng, a.loci, a.samples, seed=7, planes=(), loci=loci)
K = max(len(x) for x in loci.allele_lens) - 1
rng = np.random.default_rng(1)
ap1 = rng.dirichlet(np.full(K + 1, 0.5), size=(a.loci, a.samples))[:, :, 1:].astype(np.float32)
ap2 = rng.dirichlet(np.full(K + 1, 0.5), size=(a.loci, a.samples))[:, :, 1:].astype(np.float32)
ap1_d, ap2_d = eng.upload(ap1), eng.upload(ap2)
alen, rcls = pack_assoc_tables(loci.allele_lens, 2)
tabs = [eng.upload(t) for t in pack_dosage_tables(loci.allele_lens, 2)]
alen_d, rcls_d = eng.upload(alen), eng.upload(rcls)
cells = a.loci * a.samples
bytes_cell = 4 + 8 * K
for M in [int(x) for x in a.vecs.split(',')]:
    vec = rng.normal(size=(M, a.samples)); vec -= vec.mean(axis=1, keepdims=True); vec /= vec.std(axis=1, keepdims=True)
    vec_d = eng.upload(vec)
    for it in range(a.iters + 1):
        if it == 1:
            eng.sync(); t0 = time.time()
        res, cs, ls = eng.assoc_scan_dosage(sb.batch, vec_d, alen_d, rcls_d, ap1_d, ap2_d, *tabs)
        for d in (res.locus_int, res.locus_f64, res.allele_count, cs, ls):
            d.free()
    eng.sync(); wall = (time.time() - t0) / a.iters
    print("K=%d M=%2d  %.3f ms/pass  %.2e loci/s  %.0f GB/s (%d B/call)" % (K, M, wall * 1e3, a.loci / wall, cells * bytes_cell / wall / 1e9, bytes_cell), flush=True)
