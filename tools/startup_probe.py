"""Where does a fresh statSTR process spend its time before and after the work?  (The 1 GB command line is 0.09 s inside
a warm process and 0.75 s as a process: tools/e2e_identity.py.)  usage: startup_probe.py file.vcf.gz"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')
import sys, time
T0 = time.perf_counter()
marks = []
def mark(what):
    marks.append((what, time.perf_counter()))
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np
mark('import numpy')
from trtools_amd import _lib
mark('import trtools_amd._lib')
lib = _lib.load()
mark('dlopen libtrk.so')
from trtools_amd.statSTR import statSTR
mark('import statSTR')
from trtools_amd import runtime
comp = runtime.get_compute()
mark('runtime.get_compute() (trk_init: HIP runtime, context, queues)')
if _os.environ.get('STARTUP_TIMING'):
    _lib.set_option('TRK_VCF_TIMING', 1); _lib.set_option('TRK_INFLATE_TIMING', 1)
import argparse
ns = argparse.Namespace(vcf=sys.argv[1], out='/tmp/e2e/startup', vcftype='hipstr', samples=None, sample_prefixes=None,
                        plot_afreq=False, region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True,
                        entropy=True, mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
                        nalleles=True, nalleles_thresh=0.01, only_passing=False)
if _os.environ.get('STARTUP_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); statSTR.main(ns); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
else:
    statSTR.main(ns)
mark('statSTR.main, first run in the process')
statSTR.main(ns)
mark('statSTR.main, second run')
t = T0
for what, at in marks:
    print('%8.1f ms  %s' % ((at - t) * 1e3, what))
    t = at
print('%8.1f ms  total (the interpreter itself started before)' % ((t - T0) * 1e3))
