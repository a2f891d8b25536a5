#!/usr/bin/env python3
"""BASELINE configs[2] end to end through text at reduced scale: a synthetic GangSTR VCF (GT:DP:Q:REPCN:REPCI:RC:QEXP)
-> native reader (RC / REPCI pre-parsed) -> packed batch -> GPU (nine call filters, four locus filters) -> output VCF."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=400)
ap.add_argument('--samples', type=int, default=2000)
ap.add_argument('--out', default='/tmp/e2e')
a = ap.parse_args()
from trtools_amd import synth
os.makedirs(a.out, exist_ok=True)
path = os.path.join(a.out, 'gangstr_%dx%d.vcf' % (a.loci, a.samples))
if not os.path.exists(path):
    t = time.time()
    loci = synth.make_loci(a.loci, a.samples, seed=7, pure_repeats=True)
    idx = np.arange(a.loci)
    rows = synth.cells_numpy(7, loci, idx, a.samples)
    extra = synth.gangstr_planes_numpy(7, loci, idx, a.samples, rows['gt'], rows['dp'], 0)
    synth.render_vcf(path, loci, rows, caller='gangstr', extra=extra)
    print("generated %s (%.1f MB) in %.1fs" % (path, os.path.getsize(path) / 1e6, time.time() - t))
from trtools_amd.dumpSTR import dumpSTR
argv = ['dumpSTR', '--vcf', path, '--out', os.path.join(a.out, 'gdump'), '--vcftype', 'gangstr',
        '--gangstr-min-call-DP', '10', '--gangstr-max-call-DP', '60', '--gangstr-min-call-Q', '0.9',
        '--gangstr-expansion-prob-het', '0.05', '--gangstr-expansion-prob-hom', '0.05',
        '--gangstr-expansion-prob-total', '0.2', '--gangstr-filter-span-only', '--gangstr-filter-spanbound-only',
        '--gangstr-filter-badCI', '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05',
        '--max-locus-het', '0.9']
old = sys.argv; sys.argv = argv; args = dumpSTR.getargs(); sys.argv = old
cells = a.loci * a.samples
for rep in range(2):
    t = time.time(); rc = dumpSTR.main(args); dt = time.time() - t
    print("dumpSTR CLI, GangSTR nine call + four locus filters: rc=%d %.3fs  %.0f loci/s  %.2e cells/s" % (rc, dt, a.loci / dt, cells / dt))
    print("   path / phases:", {k: (v if not isinstance(v, dict) else {p: round(x, 4) for p, x in v.items()})
                                for k, v in dumpSTR.LAST_RUN.items()})
if os.environ.get('E2E_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); dumpSTR.main(args); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
