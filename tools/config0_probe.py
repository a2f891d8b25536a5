#!/usr/bin/env python3
"""BASELINE configs[0]: statSTR --afreq --het --mean on the real trio HipSTR file (9532 loci x 3 samples), and dumpSTR
with the HipSTR call filters + four locus filters on the same file, end to end through the CLIs.  The reference's
own time for the statSTR command in the build container was 3.93 s."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vcf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'data', 'dumpSTR',
                   'trio_chr21_hipstr.sorted.vcf.gz')
out = '/tmp/cfg0'
os.makedirs(out, exist_ok=True)
from trtools_amd.statSTR import statSTR
from trtools_amd.dumpSTR import dumpSTR


def cli(mod, argv):
    old = sys.argv
    sys.argv = argv
    try:
        a = mod.getargs()
    finally:
        sys.argv = old
    t = time.time()
    rc = mod.main(a)
    return rc, time.time() - t


for rep in range(2):   # the first run pays library load + context creation
    rc, t = cli(statSTR, ['statSTR', '--vcf', vcf, '--out', out + '/stat', '--vcftype', 'hipstr', '--afreq', '--het', '--mean'])
    print("statSTR --afreq --het --mean: rc=%d %.2f s = %.0f loci/s" % (rc, t, 9532 / t))
    rc, t = cli(dumpSTR, ['dumpSTR', '--vcf', vcf, '--out', out + '/dump', '--vcftype', 'hipstr',
                          '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '1000', '--hipstr-min-call-Q', '0.9',
                          '--hipstr-max-call-flank-indel', '0.15', '--hipstr-max-call-stutter', '0.15',
                          '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05',
                          '--max-locus-het', '0.9'])
    print("dumpSTR 5 call + 4 locus filters: rc=%d %.2f s = %.0f loci/s" % (rc, t, 9532 / t))
if os.environ.get('CFG0_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    cli(dumpSTR, ['dumpSTR', '--vcf', vcf, '--out', out + '/dump', '--vcftype', 'hipstr', '--hipstr-min-call-DP', '10',
                  '--hipstr-min-call-Q', '0.9', '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001'])
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
