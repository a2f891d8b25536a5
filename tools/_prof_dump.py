import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
from trtools_amd.dumpSTR import dumpSTR
path='/tmp/e2e/synth_2000x5000.vcf.gz'
old = sys.argv
sys.argv = ['dumpSTR', '--vcf', path, '--out', '/tmp/e2e/dump', '--vcftype', 'hipstr',
            '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9',
            '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05', '--max-locus-het', '0.9']
dargs = dumpSTR.getargs(); sys.argv = old
for i in range(2):
    t=time.time(); dumpSTR.main(dargs); print('dumpSTR wall', time.time()-t)
os.environ['TRK_VCF_TIMING']='1'
pr = cProfile.Profile(); pr.enable(); dumpSTR.main(dargs); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
