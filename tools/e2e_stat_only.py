"""statSTR's command line on the file tools/e2e_probe.py generated: wall time of three runs, then one run with the
reader's per-batch timing (TRK_VCF_TIMING) and a cProfile of the Python side."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.statSTR import statSTR
path = sys.argv[1]
ns = argparse.Namespace(vcf=path, out='/tmp/e2e/stat', vcftype='hipstr', samples=None, sample_prefixes=None,
                        plot_afreq=False, region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True,
                        entropy=True, mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
                        nalleles=True, nalleles_thresh=0.01, only_passing=False)
if os.environ.get('E2E_BATCH_CELLS'):        # experiment: smaller batches (and, E2E_HOOK_RUN_MB, smaller runs of members)
    statSTR.BATCH_CELLS = int(os.environ['E2E_BATCH_CELLS'])
if os.environ.get('E2E_HOOK_RUN_MB'):
    from trtools_amd import _lib as _L1
    _L1.set_option('TRK_VCF_HOOK_RUN_MB', int(os.environ['E2E_HOOK_RUN_MB']))
for i in range(4):
    t = time.time(); statSTR.main(ns); print("run %d: %.3f s" % (i, time.time() - t), flush=True)
from trtools_amd import _lib as _L
_L.set_option('TRK_VCF_TIMING', 1); _L.set_option('TRK_INFLATE_TIMING', 1)
t = time.time(); statSTR.main(ns); print("timed run: %.3f s" % (time.time() - t), flush=True)
_L.set_option('TRK_VCF_TIMING', None); _L.set_option('TRK_INFLATE_TIMING', None)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); statSTR.main(ns); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
