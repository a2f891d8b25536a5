#!/usr/bin/env python3
"""The headline step's call-filter pass (bench.Workload, 100k x 10k) under launch-geometry settings, with the placement
of the two output planes PINNED: one process, candidate mask planes classified with the bare stream probe against the
masked-genotype plane (profiles/r03_notes.md section 22), then every setting timed on a fast pair AND on a slow pair.
Settings are environment assignments the dispatch reads at every launch (TRK_CF_MAP / TRK_CF_WGCU / TRK_CF_LPB /
TRK_CF_NO_PERSIST / TRK_V2_MODE), given as comma-separated groups:
    python tools/cf_geometry_probe.py "TRK_CF_MAP=0,TRK_CF_NO_PERSIST=1" "TRK_CF_MAP=2,TRK_CF_WGCU=3" ...
--check: every locus of the step against the compiled oracle under the FIRST and the LAST setting."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TRK_PLACE_OUTPUTS'] = '0'
import numpy as np
import bench
from trtools_amd.engine import Engine, CallResult
from trtools_amd.synth import make_loci

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=100000)
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--rounds', type=int, default=2)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--cands', type=int, default=8)
ap.add_argument('--check', action='store_true')
ap.add_argument('settings', nargs='+')
a = ap.parse_args()
eng = Engine(0)
loci = make_loci(a.loci, a.samples, 20260931)
wl = bench.Workload(eng, 20260931, a.samples, loci, 0, 1, use_comm=False)
ins = [wl.sb.dev['gt'], wl.sb.dev['dp'], wl.sb.dev['q']]
g = wl.call_outs[0].gt_out
Lc, S = wl.n_loci, wl.n_samples
cands = [(eng.stream_probe(ins[0], ins[1], ins[2], g, wl.call_outs[0].filter_mask, Lc, S, reps=3), wl.call_outs[0].filter_mask)]
spacers = []
while len(cands) < a.cands:
    p = eng.empty((Lc, S), np.uint32)
    cands.append((eng.stream_probe(ins[0], ins[1], ins[2], g, p, Lc, S, reps=3), p))
    ts = [t for t, _ in cands]
    if max(ts) >= 1.06 * min(ts):
        break
    if len(cands) >= 4:
        try:
            spacers.append(eng.empty((16 << 30,), np.uint8))
        except Exception:
            pass
print("bare stream (k_stream_probe, the r03 product shape) of masked-genotype plane + candidate mask planes, ms:",
      ' '.join('%.3f' % t for t, _ in cands), flush=True)
cands.sort(key=lambda c: c[0])
pairs = [('fast', cands[0])]
if cands[-1][0] >= 1.06 * cands[0][0]:
    pairs.append(('slow', cands[-1]))
else:
    print("only ONE level among the candidates", flush=True)


def use_mask(m):
    for i, co in enumerate(wl.call_outs):
        wl.call_outs[i] = CallResult(co.gt_out, m, co.sample_counters, co.sample_totaldp, co.sample_dp_missing, co.error,
                                     co.sample_totaldp_f64)


def apply(setting):
    keys = []
    for kv in setting.split(','):
        if not kv or kv == '-':
            continue
        k, v = kv.split('=', 1)
        os.environ[k] = v
        keys.append(k)
    return keys


res = {}
for r in range(a.rounds):
    for tag, (t_probe, m) in pairs:
        use_mask(m)
        for si, st in enumerate(a.settings):
            keys = apply(st)
            el, prof = wl.run(a.steps, 2)
            n, ms = prof['k_call_filter']
            rn, rms = prof.get('k_cf_reduce', (0, 0.0))
            res.setdefault((tag, st), []).append((el / a.steps * 1e3, ms / n, rms / max(rn, 1)))
            if a.check and r == 0 and tag == pairs[0][0] and si in (0, len(a.settings) - 1):
                c = bench.exhaustive_check(wl, True)
                print("setting %s: parity %d loci, %d calls bit for bit, worst float %.2e" %
                      (st, c['loci'], c['calls_bit_for_bit'], c['worst_float_rel']), flush=True)
            for k in keys:
                del os.environ[k]
for tag, (t_probe, _) in pairs:
    print("== %s pair (bare r03-shape stream %.3f ms)" % (tag, t_probe))
    for st in a.settings:
        v = res[(tag, st)]
        print("  %-58s k_call_filter %s   k_cf_reduce %s   ms/step %s" % (
            st, ' '.join('%.3f' % x[1] for x in v), ' '.join('%.3f' % x[2] for x in v), ' '.join('%.3f' % x[0] for x in v)),
            flush=True)
eng.close()
