#!/usr/bin/env python3
"""Record what the REAL reference's qcSTR accumulates in its per-record loop (build container only).

    python tools/gen_golden_qcstr.py     # rewrites tests/golden/qcstr/

qcSTR's products are PDF plots; the numbers behind them are what its main loop hands to the plotting functions
(trtools/qcSTR/qcSTR.py:529-561, 587-660).  This script imports the reference (cyvcf2 through tools/refshim),
replaces the plotting functions by recorders and runs ``qcSTR.main`` on the reference's own test VCFs: per-sample
call counts, per-chromosome call counts, per-sample quality means and per-locus quality means are written to
tests/golden/qcstr/cases.json, and the input VCFs (reference test data) are copied next to it as fixtures.
"""
import argparse
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'tests', 'golden', 'qcstr')
VCFS = '/root/reference/trtools/testsupport/sample_vcfs'

# name, vcf (relative to the reference's sample_vcfs), vcftype, samples file or None, quality kinds, ignore no-calls
CASES = [
    ('many_samples_per_locus', 'many_samples.vcf.gz', 'auto', None, ['per-locus'], False),
    ('many_samples_per_sample', 'many_samples.vcf.gz', 'auto', None, ['per-sample', 'per-locus'], False),
    ('many_samples_ignore', 'many_samples.vcf.gz', 'auto', None, ['per-sample', 'per-locus'], True),
    ('many_samples_subset', 'many_samples.vcf.gz', 'auto', 'many_samples_subsample1.txt', ['per-sample', 'per-locus'], False),
    ('many_samples_subset_ignore', 'many_samples.vcf.gz', 'auto', 'many_samples_subsample2.txt', ['per-sample', 'per-locus'], True),
    ('multi_chrom', 'many_samples_multiple_chroms.vcf.gz', 'auto', None, ['per-sample', 'per-locus'], False),
    ('few_samples', 'few_samples_few_loci.vcf.gz', 'auto', None, ['per-sample', 'per-locus'], True),
    ('gangstr', 'test_gangstr.vcf', 'gangstr', None, ['per-sample', 'per-locus'], False),
    ('gangstr_ignore', 'test_gangstr.vcf', 'gangstr', None, ['per-sample', 'per-locus'], True),
    ('popstr_no_quality', 'qc_vcfs/test_popstr.vcf', 'popstr', None, [], False),
]


def main():
    sys.path.insert(0, '/root/reference')
    import numpy as np
    import trtools.qcSTR.qcSTR as rq     # the reference
    rec = {}

    def grab(name, keep):
        def f(*a, **k):
            rec[name] = keep(*a, **k)
        return f
    rq.OutputDiffRefBias = grab('diffref_bias', lambda d, r, *a, **k: dict(n=len(d), sum_diffs=float(np.sum(d)), sum_reflens=float(np.sum(r))))
    rq.OutputDiffRefHistogram = grab('diffref_hist', lambda d, *a, **k: dict(n=len(d), sum=float(np.sum(d))))
    rq.OutputSampleCallrate = grab('sample_calls', lambda c, s, *a, **k: dict(calls=[float(x) for x in c], samples=list(s)))
    rq.OutputChromCallrate = grab('chrom_calls', lambda c, *a, **k: {str(kk): float(v) for kk, v in c.items()})
    rq.OutputQualityPerSample = grab('per_sample_quality', lambda q, *a, **k: [float(x) for x in q])
    rq.OutputQualityPerLocus = grab('per_locus_quality', lambda q, *a, **k: [float(x) for x in q])
    for n in ('OutputQualitySampleStrat', 'OutputQualityLocusStrat', 'OutputQualityPerCall'):
        setattr(rq, n, lambda *a, **k: None)
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    tmp = tempfile.mkdtemp()
    cases = []
    copied = set()
    for name, vcf, vcftype, samples, quality, ignore in CASES:
        src = os.path.join(VCFS, vcf)
        args = argparse.Namespace(vcf=src, out=os.path.join(tmp, name), vcftype=vcftype,
                                  samples=os.path.join(VCFS, samples) if samples else None, period=None,
                                  quality=list(quality), quality_ignore_no_call=ignore, refbias_metric='mean',
                                  refbias_mingts=100, refbias_xrange_min=0, refbias_xrange_max=100, refbias_binsize=5,
                                  numrecords=None)
        rec.clear()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            rc = int(rq.main(args))
        for f in [vcf] + ([samples] if samples else []):
            if f not in copied:
                shutil.copy(os.path.join(VCFS, f), os.path.join(OUT, os.path.basename(f)))
                copied.add(f)
        cases.append(dict(name=name, vcf=os.path.basename(vcf), vcftype=vcftype,
                          samples=os.path.basename(samples) if samples else None, quality=quality, ignore_no_call=ignore,
                          rc=rc, recorded={k: v for k, v in rec.items()}))
        print('%-28s rc=%d  %s' % (name, rc, sorted(rec)))
    with open(os.path.join(OUT, 'cases.json'), 'w') as fh:
        json.dump({'generator': 'tools/gen_golden_qcstr.py', 'cases': cases}, fh, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
