"""Wall and CPU seconds (user + sys, all threads) of the two 1 GB command lines per reader / formatter thread count.
usage: e2e_cpu.py file.vcf.gz [threads ...]"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.statSTR import statSTR
from trtools_amd.dumpSTR import dumpSTR
path = sys.argv[1]
threads = [int(x) for x in sys.argv[2:]] or [0]
ns = argparse.Namespace(vcf=path, out='/tmp/e2e/stat', vcftype='hipstr', samples=None, sample_prefixes=None, plot_afreq=False,
                        region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True, mean=True,
                        mode=True, var=True, numcalled=True, use_length=False, precision=4, nalleles=True,
                        nalleles_thresh=0.01, only_passing=False)
old = sys.argv
sys.argv = ['dumpSTR', '--vcf', path, '--out', '/tmp/e2e/dump', '--vcftype', 'hipstr', '--hipstr-min-call-DP', '10',
            '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9', '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001',
            '--min-locus-het', '0.05', '--max-locus-het', '0.9']
dargs = dumpSTR.getargs()
sys.argv = old
devnull = open(os.devnull, 'w')


def timed(f, arg):
    best = None
    for _ in range(3):
        for g in os.listdir('/tmp/e2e'):
            if g.startswith('dump.'):
                os.remove(os.path.join('/tmp/e2e', g))
        so = sys.stdout; sys.stdout = devnull
        r0 = resource.getrusage(resource.RUSAGE_SELF); t = time.time()
        try:
            f(arg)
        finally:
            sys.stdout = so
        dt = time.time() - t; r1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime
        if best is None or dt < best[0]:
            best = (dt, cpu)
    return best


statSTR.main(ns); dumpSTR.main(dargs)          # warm-up
for n in threads:
    if n:
        os.environ['TRK_VCF_THREADS'] = os.environ['TRK_FMT_THREADS'] = str(n)
    s, d = timed(statSTR.main, ns), timed(dumpSTR.main, dargs)
    print("threads %-7s statSTR %.3f s wall, %.2f CPU-s    dumpSTR %.3f s wall, %.2f CPU-s" % (n or 'default', s[0], s[1], d[0], d[1]), flush=True)
