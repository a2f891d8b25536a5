"""statSTR --samples a,b on the file tools/e2e_probe.py generated: the class-ordered columns the reader lays out while
parsing (default) against the grouped kernel on file-order columns (TRK_CLASS_SORT=0): wall time, count-kernel time,
and the two tables must be equal."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd import runtime, vcfnative
from trtools_amd.statSTR import statSTR
path = sys.argv[1]
r = vcfnative.NativeVCFReader(path)
names = list(r.samples)
r.close()
rng = np.random.default_rng(1)
pick = rng.random(len(names)) < 0.4
fa, fb = '/tmp/e2e/grp_a.txt', '/tmp/e2e/grp_b.txt'
open(fa, 'w').write('\n'.join(n for n, p in zip(names, pick) if p) + '\n')
open(fb, 'w').write('\n'.join(n for n, p in zip(names, pick) if not p) + '\n')
eng = runtime.get_compute().eng
tabs = {}
for mode in ('dev', '1', '0', 'dev', '1', '0'):      # dev (round 6): the columns parsed on the device, grouped kernel on file order
    os.environ['TRK_GROUPS_DEVICE_PARSE'] = '1' if mode == 'dev' else '0'
    os.environ['TRK_CLASS_SORT'] = mode if mode != 'dev' else '1'
    ns = argparse.Namespace(vcf=path, out='/tmp/e2e/statg' + mode, vcftype='hipstr', samples=fa + ',' + fb, sample_prefixes=None,
                            plot_afreq=False, region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True,
                            entropy=True, mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
                            nalleles=True, nalleles_thresh=0.01, only_passing=False)
    eng.profile(True); eng.profile_reset()
    t = time.time(); rc = statSTR.main(ns); dt = time.time() - t
    pg = eng.profile_get(); eng.profile(False)
    c = pg['k_locus_count']
    print("mode %s (dev: device parse; 1 / 0: host parse with / without the class-ordered columns): rc %d  %.3f s   k_locus_count %d launches, %.3f ms each  %s" % (mode, rc, dt, c[0], c[1] / max(c[0], 1), {k: statSTR.LAST_RUN.get(k) for k in ('device_parse', 'device_inflate')}), flush=True)
    tabs[mode] = open(ns.out + '.tab').read()
print("tables equal:", tabs['1'] == tabs['0'] == tabs['dev'], " rows:", tabs['1'].count('\n') - 1)
