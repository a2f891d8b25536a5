"""associaTR's command line on the file tools/e2e_probe.py generated (/tmp/e2e): seconds per run (round 6: where the
command line of BASELINE configs[4] stands next to its scan kernel)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.associaTR import associaTR as at
from trtools_amd import vcfnative
path = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
r = vcfnative.NativeVCFReader(path)
S = len(r.samples)
r.close()
rng = np.random.default_rng(5)
tr = '/tmp/e2e/traits.npy'
np.save(tr, np.column_stack([rng.normal(size=S), rng.normal(size=S)]))
old = sys.argv
sys.argv = ['associaTR', '/tmp/e2e/assoc.tsv', path, 'pheno', tr, '--same-samples', '--vcftype', 'hipstr']
args = at.getargs()
sys.argv = old
for i in range(runs):
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        at.main(args)
    print("run %d: %.3f s  (%d rows)" % (i, time.time() - t, sum(1 for _ in open('/tmp/e2e/assoc.tsv')) - 1), flush=True)
if os.environ.get('E2E_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        at.main(args)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(25)
