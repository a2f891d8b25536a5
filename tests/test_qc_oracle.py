"""qcSTR's reductions: the oracle restatement (oracle/trtools_oracle.py qc_record / qc_accumulate) against what the
REAL reference's qcSTR main loop handed to its plotting functions (tests/golden/qcstr/cases.json, written by
tools/gen_golden_qcstr.py from the reference's own test VCFs)."""
import json
import os

import numpy as np
import pytest

from oracle import trtools_oracle as orc
from trtools_amd.batch import genotype_matrix
from trtools_amd.utils import tr_harmonizer as trh
from trtools_amd.utils import utils

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden', 'qcstr')
CASES = json.load(open(os.path.join(GOLD, 'cases.json')))['cases']


def load_case(case):
    """(records [(chrom, gt [S, P], quality [S, 1] or None)], sample_index, selected names) of a golden case, read
    with this repository's own VCF reader and harmoniser."""
    invcf = utils.LoadSingleReader(os.path.join(GOLD, case['vcf']), checkgz=False)
    vt = case['vcftype']
    harmonizer = trh.TRRecordHarmonizer(invcf, vt) if vt != 'auto' else trh.TRRecordHarmonizer(invcf)
    names = np.array(invcf.samples)
    if case['samples']:
        wanted = [x.strip() for x in open(os.path.join(GOLD, case['samples'])).readlines()]
        sample_index = np.isin(names, wanted)
    else:
        sample_index = np.ones(len(names), dtype=bool)
    use_q = len(case['quality']) > 0
    recs = []
    for r in harmonizer:
        m, _ = genotype_matrix(r.vcfrecord)
        recs.append((r.chrom, m, np.asarray(r.GetQualityScores(), dtype=np.float32) if use_q else None))
    return recs, sample_index, list(names[sample_index])


def check_against_golden(case, got, samples, float_tol):
    want = case['recorded']
    if 'sample_calls' in want:      # (skipped by the reference when there is one sample)
        assert want['sample_calls']['samples'] == samples
        assert np.array_equal(np.asarray(want['sample_calls']['calls']), np.asarray(got['sample_calls']))
    if 'chrom_calls' in want:
        assert {k: float(v) for k, v in got['chrom_calls'].items()} == want['chrom_calls']
    if 'per_sample_quality' in want:
        np.testing.assert_allclose(got['per_sample_quality'], want['per_sample_quality'], rtol=float_tol, atol=0,
                                   equal_nan=True)
    if 'per_locus_quality' in want:
        np.testing.assert_allclose(np.asarray(got['per_locus_quality'], dtype=np.float64),
                                   want['per_locus_quality'], rtol=float_tol, atol=0, equal_nan=True)


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_oracle_equals_reference_accumulators(case):
    assert case['rc'] == 0
    recs, sample_index, samples = load_case(case)
    got = orc.qc_accumulate(recs, sample_index, case['ignore_no_call'])
    # the same numpy operations on the same arrays: bit for bit
    check_against_golden(case, got, samples, float_tol=0)


def test_half_missing_call_counts_and_all_missing_does_not():
    gt = np.array([[0, 1], [-1, 1], [-1, -1], [-1, -2], [-2, -2]])
    q = np.array([[0.5], [0.25], [0.75], [np.nan], [1.0]], dtype=np.float32)
    calls, q0, mean0 = orc.qc_record(gt, q)
    assert calls.tolist() == [True, True, False, True, True]
    assert q0.reshape(-1).tolist() == [0.5, 0.25, 0.0, 0.0, 1.0] and mean0 == np.float32(1.75 / 5)
    _, q1, mean1 = orc.qc_record(gt, q, ignore_no_call=True)
    assert np.isnan(q1.reshape(-1)[[2, 3]]).all() and mean1 == np.float32(1.75 / 3)
    sel = np.array([True, False, True, False, True])
    calls, q2, mean2 = orc.qc_record(gt, q, sel)
    assert calls.tolist() == [True, False, True] and mean2 == np.float32(1.5 / 3)


@pytest.mark.parametrize('case', [c for c in CASES if c['name'] in ('many_samples_subset_ignore', 'multi_chrom', 'gangstr',
                                                                     'popstr_no_quality')],
                         ids=lambda c: c['name'])
def test_product_host_logic_with_the_oracle_behind_the_compute_seam(case):
    """trtools_amd.qcSTR.qc_reductions (reader, harmoniser, batching, accumulation) with OracleCompute standing in
    for the device: the host logic alone must reproduce the reference's accumulators."""
    from oracle_compute import OracleCompute
    from trtools_amd import runtime
    from trtools_amd.qcSTR import qc_reductions
    old = runtime.set_compute(OracleCompute())
    try:
        got = qc_reductions(os.path.join(GOLD, case['vcf']), vcftype=case['vcftype'],
                            samples=os.path.join(GOLD, case['samples']) if case['samples'] else None,
                            quality=case['quality'], quality_ignore_no_call=case['ignore_no_call'], batch_loci=257)
    finally:
        runtime.set_compute(old)
    assert got is not None
    check_against_golden(case, got, got['samples'], float_tol=2e-6)
    want = case['recorded']
    assert got['n_alleles'] == want['diffref_hist']['n']
    np.testing.assert_allclose(got['sum_diff_unit'], want['diffref_hist']['sum'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(got['sum_diff_bp'], want['diffref_bias']['sum_diffs'], rtol=1e-9, atol=1e-6)


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray) or isinstance(y, np.ndarray) or (isinstance(x, list) and x and isinstance(x[0], float)):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True), k
        else:
            assert x == y, k


_ROADS = [('many_samples_subset_ignore', dict()), ('many_samples_subset_ignore', dict(period=2)),
          ('many_samples_subset_ignore', dict(period=4, numrecords=3)), ('multi_chrom', dict()), ('multi_chrom', dict(numrecords=7)),
          ('gangstr_ignore', dict(numrecords=60)), ('gangstr_ignore', dict(period=2)), ('few_samples', dict(numrecords=7)),
          ('popstr_no_quality', dict(numrecords=40))]


@pytest.mark.parametrize('case, extra', [(next(c for c in CASES if c['name'] == n), e) for n, e in _ROADS],
                         ids=['%s %s' % (n, ' '.join('%s=%s' % kv for kv in e.items()) or 'all') for n, e in _ROADS])
def test_the_batch_road_equals_the_record_objects(case, extra):
    """qc_reductions a batch at a time (native reader -> native batch harmoniser -> the device passes over the batch's
    tables) against the loop over record objects it replaced (TRK_QC_BATCH=0): every accumulator the same, bit for bit --
    with --period (batches it cuts go through the record objects) and --numrecords (nothing read beyond them)."""
    from helpers import lab_env
    from oracle_compute import OracleCompute
    from trtools_amd import runtime
    from trtools_amd.qcSTR import reductions
    kw = dict(vcftype=case['vcftype'], samples=os.path.join(GOLD, case['samples']) if case['samples'] else None,
              quality=case['quality'], quality_ignore_no_call=case['ignore_no_call'], batch_loci=5, **extra)
    old = runtime.set_compute(OracleCompute())
    try:
        got = reductions.qc_reductions(os.path.join(GOLD, case['vcf']), **kw)
        road = dict(reductions.LAST_RUN)
        with lab_env(TRK_QC_BATCH='0'):
            want = reductions.qc_reductions(os.path.join(GOLD, case['vcf']), **kw)
            assert reductions.LAST_RUN['path'] == 'per-record'
    finally:
        runtime.set_compute(old)
    assert road['path'] in ('batch', 'mixed') and road['batches'] > 0, road
    if 'period' not in extra:
        assert road['path'] == 'batch' and road['fallback_batches'] == 0, road
    _same(got, want)
