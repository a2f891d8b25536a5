#!/usr/bin/env python3
"""The strong-scaling shard regime on one GPU: the bench step at --loci loci under launch-geometry knobs
(TRK_CF_LPB ...), kernel times from the library's HIP-event brackets.  `gpurun -- python tools/shard_probe.py`."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from trtools_amd.engine import Engine
from trtools_amd.synth import make_loci

ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, nargs='+', default=[12500])
ap.add_argument('--samples', type=int, default=10000)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--knobs', nargs='*', default=['', 'TRK_CF_LPB=64', 'TRK_CF_LPB=49', 'TRK_CF_LPB=32', 'TRK_CF_LPB=24',
                                               'TRK_CF_LPB=16', 'TRK_CF_LPB=8'])
a = ap.parse_args()
eng = Engine(0)
eng.comm_init(0, 1, eng.comm_unique_id())
loci = make_loci(max(a.loci), a.samples, 20260931)
for n in a.loci:
    wl = bench.Workload(eng, 20260931, a.samples, loci.slice(0, n), 0, 1, use_comm=True)
    wl.pipeline_count = bool(os.environ.get('PIPE'))
    for knob in a.knobs:
        kv = [k.split('=') for k in knob.split(',') if k]
        for k, v in kv:
            os.environ[k] = v
        wl.pipeline_count = bool(os.environ.get('PIPE'))
        el, prof = wl.run(a.steps, 3)
        for k, v in kv:
            del os.environ[k]
        ms = el / a.steps * 1e3
        km = {k: round(v[1] / v[0], 4) for k, v in prof.items() if v[0]}
        cf = km.get('k_call_filter', 0)
        print("loci %6d %-28s step %.3f ms  call_filter %.3f (%.2f of peak)  count %.3f  fin %.3f  locf %.3f" % (
            n, knob or 'default', ms, cf, n * a.samples * 20 / (cf * 1e-3) / 8e12 if cf else 0, km.get('k_locus_count', 0),
            km.get('k_locus_finalize', 0), km.get('k_locus_filter', 0)), flush=True)
    # the call-filter kernel alone, with and without the delta outputs
    b = wl.sb.batch
    for delta in (True, False):
        eng.profile(True); eng.profile_reset()
        for _ in range(20):
            eng.call_filters(b, wl.planes, wl.filters, dp_plane=0, out=wl.call_outs[0],
                             delta_stats=wl.stats_b[0] if delta else None)
        eng.sync()
        n_, ms_ = eng.profile_get()['k_call_filter']
        eng.profile(False)
        print("loci %6d call filter alone, delta=%s: %.3f ms" % (n, delta, ms_ / n_), flush=True)
    wl.free()
