"""Per-batch seconds and minor page faults of the reader's and the record writer's native calls in one dumpSTR run of
tools/e2e_dump_only.py's command line (who re-faults 150 MB every other batch?)."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time, resource, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd import vcfnative
from trtools_amd.dumpSTR import dumpSTR

def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        r0 = resource.getrusage(resource.RUSAGE_SELF); t = time.perf_counter()
        out = f(*a, **k)
        dt = time.perf_counter() - t; r1 = resource.getrusage(resource.RUSAGE_SELF)
        print("  %-14s %6.1f ms  cpu %6.1f ms (sys %5.1f)  minflt %6d  nvcsw %d nivcsw %d" % (label, dt * 1e3,
              (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) * 1e3, (r1.ru_stime - r0.ru_stime) * 1e3,
              r1.ru_minflt - r0.ru_minflt, r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw), flush=True)
        return out
    setattr(obj, name, g)

wrap(vcfnative.NativeVCFReader, '_read_raw_batch', 'read')
wrap(vcfnative.RawBatch, 'dumpstr_lines', 'records')
wrap(vcfnative.RawBatch, 'harmonize', 'harmonize')
from trtools_amd import vcfio
wrap(vcfio.VCFWriter, 'write_bytes', 'write_bytes')
path = sys.argv[1]
old = sys.argv
sys.argv = ['dumpSTR', '--vcf', path, '--out', '/tmp/e2e/dump', '--vcftype', 'hipstr',
            '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9',
            '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05', '--max-locus-het', '0.9']
dargs = dumpSTR.getargs()
sys.argv = old
for i in range(3):
    for f in glob.glob('/tmp/e2e/dump.*'):
        os.remove(f)
    print("run", i, flush=True)
    t = time.time(); dumpSTR.main(dargs); print("run %d: %.3f s" % (i, time.time() - t), flush=True)
