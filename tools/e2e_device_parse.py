"""statSTR's 1 GB command line with the sample columns parsed on the host (default) and on the device
(TRK_DEVICE_PARSE=1): wall / CPU seconds, best of three, and the two tables compared.
usage: e2e_device_parse.py /tmp/e2e/synth_17000x5000.vcf.gz"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, hashlib, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.statSTR import statSTR
path = sys.argv[1]
devnull = open(os.devnull, 'w')


def run(out, dev):
    ns = argparse.Namespace(vcf=path, out=out, vcftype='hipstr', samples=None, sample_prefixes=None, plot_afreq=False,
                            region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True, mean=True,
                            mode=True, var=True, numcalled=True, use_length=False, precision=4, nalleles=True,
                            nalleles_thresh=0.01, only_passing=False)
    os.environ['TRK_DEVICE_PARSE'] = '1' if dev else '0'
    best = None
    for i in range(4):
        so = sys.stdout; sys.stdout = devnull
        r0 = resource.getrusage(resource.RUSAGE_SELF); t = time.time()
        try:
            statSTR.main(ns)
        finally:
            sys.stdout = so
        dt = time.time() - t; r1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime
        if i and (best is None or dt < best[0]):
            best = (dt, cpu)
    return best, hashlib.sha256(open(out + '.tab', 'rb').read()).hexdigest()[:16], dict(statSTR.LAST_RUN)


for rep in range(2):
    (th, ch), hh, lh = run('/tmp/e2e/stat_host', False)
    (td, cd), hd, ld = run('/tmp/e2e/stat_dev', True)
    print("host parse   %.3f s wall, %.2f CPU-s   table %s" % (th, ch, hh))
    print("device parse %.3f s wall, %.2f CPU-s   table %s  %s  (%s)" % (td, cd, hd, 'identical' if hh == hd else 'DIFFERENT', ld), flush=True)
if os.environ.get('E2E_PROFILE'):
    import cProfile, pstats
    os.environ['TRK_DEVICE_PARSE'] = '1'
    ns = argparse.Namespace(vcf=path, out='/tmp/e2e/stat_dev', vcftype='hipstr', samples=None, sample_prefixes=None, plot_afreq=False,
                            region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True, mean=True,
                            mode=True, var=True, numcalled=True, use_length=False, precision=4, nalleles=True,
                            nalleles_thresh=0.01, only_passing=False)
    pr = cProfile.Profile(); pr.enable(); statSTR.main(ns); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(16)
