#!/usr/bin/env python3
"""Experiment: the headline step on one GPU as two 50k-locus halves run alternately (count pass of one half beside the
call filters of the other), against one 100k-locus pass.  Not the bench contract."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from trtools_amd.engine import Engine
from trtools_amd.synth import make_loci
eng = Engine(0)
loci = make_loci(100000, 10000, 20260931)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def timed(wls):
    for _ in range(3):
        for w in wls: w.step()
    for w in wls: w.flush()
    eng.sync()
    eng.profile(True); eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(K):
        for w in wls: w.step()
    for w in wls: w.flush()
    eng.sync()
    dt = (time.perf_counter() - t0) / K
    p = eng.profile_get(); eng.profile(False)
    return dt, {k: (v[1] / v[0] if v[0] else None) for k, v in p.items() if k in ('k_call_filter', 'k_locus_count')}
if os.environ.get('MODE', 'halves') == 'halves':
    parts = int(os.environ.get('PARTS', '2'))
    per = 100000 // parts
    wls = []
    for h in range(parts):
        w = bench.Workload(eng, 20260931, 10000, loci.slice(h * per, (h + 1) * per), h * per, 1, use_comm=False,
                           pipeline_count=True, n_sets=2, ev_base=8 * (h & 1))
        wls.append(w)
    dt, p = timed(wls)
    print("%d parts alternating: %.3f ms per 100k loci  %s" % (parts, dt * 1e3, p))
else:
    w = bench.Workload(eng, 20260931, 10000, loci, 0, 1, use_comm=False, pipeline_count=os.environ.get('PIPE') == '1')
    dt, p = timed([w])
    print("one pass (pipe=%s): %.3f ms per 100k loci  %s" % (os.environ.get('PIPE'), dt * 1e3, p))
