#!/usr/bin/env python3
"""How often does trk_dev_alloc_pair end on the fast level?  One fresh process per trial (the driver's allocation state
is per process): three 4 GB input planes as the bench holds them, then the placed pair; prints trk_pair_info."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from trtools_amd.engine import Engine
    eng = Engine(0)
    L, S = 100000, 10016

    class B:
        n_loci, n_samples, ploidy = L, S, 2
    ins = [eng.empty((L, S), np.uint32) for _ in range(3)]
    g, m = eng.placed_output_pair(B)
    print(Engine.last_placement, flush=True)
    eng.close()
else:
    for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        subprocess.run([sys.executable, __file__, 'child'])
