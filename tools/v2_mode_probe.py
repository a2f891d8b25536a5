#!/usr/bin/env python3
"""Same-box A/B of the call-filter kernel's build variants on the headline step (bench.Workload):
TRK_V2_MODE values given on the command line (the dispatch reads the variable at every launch), alternating rounds.
usage: python tools/v2_mode_probe.py [--loci L] [--rounds R] mode [mode ...]"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from trtools_amd.engine import Engine
from trtools_amd.synth import make_loci

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--var', default='TRK_V2_MODE')
ap.add_argument('--check', action='store_true')
ap.add_argument('modes', nargs='+')
a = ap.parse_args()
eng = Engine(0)
loci = make_loci(a.loci, a.samples, 20260931)
wl = bench.Workload(eng, 20260931, a.samples, loci, 0, 1, use_comm=False)
res = {m: [] for m in a.modes}
for r in range(a.rounds):
    for m in a.modes:
        os.environ[a.var] = m
        el, prof = wl.run(a.steps, 2)
        n, ms = prof['k_call_filter']
        cn, cms = prof['k_locus_count']
        res[m].append((el / a.steps * 1e3, ms / n, cms / max(cn, 1)))
        if a.check and r == 0:
            c = bench.exhaustive_check(wl, True)
            print("mode %s: parity %d loci, %d calls bit for bit, worst float %.2e" %
                  (m, c['loci'], c['calls_bit_for_bit'], c['worst_float_rel']), flush=True)
for m in a.modes:
    print("%s=%-4s ms/step %s   k_call_filter %s   k_locus_count %s" % (
        a.var, m, ' '.join('%.3f' % x[0] for x in res[m]), ' '.join('%.3f' % x[1] for x in res[m]),
        ' '.join('%.3f' % x[2] for x in res[m])), flush=True)
eng.close()
