#!/usr/bin/env python3
"""cProfile of the statSTR command line on the synthetic file of tools/e2e_probe.py (host-side hot spots)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--vcf', default='/tmp/e2e/synth_2000x5000.vcf.gz')
a = ap.parse_args()
from trtools_amd.statSTR import statSTR
ns = argparse.Namespace(vcf=a.vcf, out='/tmp/e2e/stat_prof', vcftype='hipstr', samples=None, sample_prefixes=None,
                        plot_afreq=False, region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True,
                        entropy=True, mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
                        nalleles=True, nalleles_thresh=0.01, only_passing=False)
statSTR.main(ns)      # warm (library load, first batch)
cProfile.run('statSTR.main(ns)', '/tmp/e2e/stat.prof')
pstats.Stats('/tmp/e2e/stat.prof').sort_stats('tottime').print_stats(18)
