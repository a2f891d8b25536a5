#!/usr/bin/env python3
"""Kernel timeline of the 12.5k-locus shard step (what one of eight ranks runs): `rocprofv3 --kernel-trace -d DIR --
python tools/timeline_probe.py run`, then `python tools/timeline_probe.py show DIR` prints, for the last steps, every
kernel's start / end relative to the step's call-filter kernel and the idle gap of the call-filter queue."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import glob, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == 'run':
    import bench
    from trtools_amd.engine import Engine
    from trtools_amd.synth import make_loci
    eng = Engine(0)
    eng.comm_init(0, 1, eng.comm_unique_id())
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
    loci = make_loci(n, 10000, 20260931)
    wl = bench.Workload(eng, 20260931, 10000, loci, 0, 1, use_comm=True, pipeline_count=True, gather_loci=n)
    el, _ = wl.run(12, 3)
    print("step %.3f ms" % (el / 12 * 1e3))
else:
    db = [f for f in glob.glob(os.path.join(sys.argv[2], '**', '*.db'), recursive=True)][0]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = con.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    import re
    ev = [((re.search(r'(k_\w+)', n) or re.search(r'(\w+)', n)).group(1), st, en, q) for n, st, en, q in rows]
    cfs = [e for e in ev if e[0].startswith('k_call_filter')]
    for a, b in zip(cfs[-5:-1], cfs[-4:]):
        print("--- call filter %.1f us, gap to the next one %.1f us" % ((a[2] - a[1]) / 1e3, (b[1] - a[2]) / 1e3))
        for e in ev:
            if a[1] <= e[1] < b[1] and e is not a:
                print("   %-22s q%-3s start +%7.1f us  dur %6.1f us" % (e[0][:22], e[3], (e[1] - a[1]) / 1e3, (e[2] - e[1]) / 1e3))
