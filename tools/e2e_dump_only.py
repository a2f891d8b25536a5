"""dumpSTR's command line on the file tools/e2e_probe.py generated (/tmp/e2e), three runs: seconds and phases."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.dumpSTR import dumpSTR
path = sys.argv[1]
old = sys.argv
sys.argv = ['dumpSTR', '--vcf', path, '--out', '/tmp/e2e/dump', '--vcftype', 'hipstr',
            '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9',
            '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05', '--max-locus-het', '0.9']
if os.environ.get('E2E_ZIP'):          # round 6: --zip (libtrk's BGZF members, the index from the places the writer noted)
    sys.argv.append('--zip')
dargs = dumpSTR.getargs()
sys.argv = old
import glob
if os.environ.get('E2E_SWITCH'):       # the interpreter's thread switch interval (default 5 ms): what a thread that wants the GIL waits for
    sys.setswitchinterval(float(os.environ['E2E_SWITCH']))
for i in range(3):
    for f in glob.glob('/tmp/e2e/dump.*'):
        os.remove(f)        # (truncating last run's 1.5 GB output is 0.15 s of open(): not the command line's time)
    import resource
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t = time.time(); rc = dumpSTR.main(dargs); dt = time.time() - t
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    print("run %d: %.3f s  (CPU %.2f s)  phases %s" % (i, dt, (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime),
                                                     {p: round(x, 3) for p, x in dumpSTR.LAST_RUN['seconds'].items()}), flush=True)
    if os.environ.get('E2E_ZIP'):
        print("   output %.0f MB + index %.0f KB  (TRK_ZIP_LEVEL %s, TRK_DEVICE_DEFLATE %s)" % (os.path.getsize('/tmp/e2e/dump.vcf.gz') / 1e6, os.path.getsize('/tmp/e2e/dump.vcf.gz.tbi') / 1e3, os.environ.get('TRK_ZIP_LEVEL', '6'), os.environ.get('TRK_DEVICE_DEFLATE', '0')), flush=True)
if os.environ.get('E2E_FMT_TIMING'):
    from trtools_amd import _lib as _L
    _L.set_option('TRK_FMT_TIMING', 1)
    t = time.time(); rc = dumpSTR.main(dargs); print("timed run: %.3f s" % (time.time() - t), flush=True)
    _L.set_option('TRK_FMT_TIMING', None)
if os.environ.get('E2E_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); dumpSTR.main(dargs); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
