"""qcSTR's reductions (trtools_amd.qcSTR.qc_reductions) on the file tools/e2e_probe.py generated: the batch road (native reader ->
native batch harmoniser -> trk_qc_reduce / trk_locus_stats over the batch's tables) and, E2E_QC_RECORDS=n, the loop over record
objects it replaced on the first n records."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd.qcSTR import reductions
path = sys.argv[1]
for i in range(3):
    t = time.time(); r = reductions.qc_reductions(path, vcftype='hipstr', quality=['per-locus', 'per-sample'], batch_loci=3355); dt = time.time() - t
    print("batch road, run %d: %.3f s for %d records x %d samples  %s  (calls %d, alleles %d)" % (
        i, dt, r['numrecords'], len(r['samples']), dict(reductions.LAST_RUN), int(r['sample_calls'].sum()), r['n_alleles']), flush=True)
n = int(os.environ.get('E2E_QC_RECORDS', '0'))
if n:
    os.environ['TRK_QC_BATCH'] = '0'
    t = time.time(); r2 = reductions.qc_reductions(path, vcftype='hipstr', quality=['per-locus', 'per-sample'], numrecords=n, batch_loci=1024); dt = time.time() - t
    print("record objects: %.3f s for %d records (%.1f s for the file at that rate)  %s" % (dt, n, dt / n * r['numrecords'], dict(reductions.LAST_RUN)), flush=True)
    del os.environ['TRK_QC_BATCH']
    r1 = reductions.qc_reductions(path, vcftype='hipstr', quality=['per-locus', 'per-sample'], numrecords=n, batch_loci=1024)
    import numpy as np
    def eq(x, y):
        if isinstance(x, dict) or isinstance(x, (int, str)) or x is None:
            return x == y
        x, y = np.asarray(x), np.asarray(y)
        return np.array_equal(x, y, equal_nan=True) if x.dtype.kind == 'f' else np.array_equal(x, y)
    same = all(eq(r1[k], r2[k]) for k in r1)
    print("the two roads on those records: equal =", same)
