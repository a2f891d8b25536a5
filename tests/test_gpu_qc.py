"""trk_qc_reduce (qcSTR's reductions) on the device against the oracle restatement and,
through trtools_amd.qcSTR.qc_reductions, against the reference-generated goldens of tests/test_qc_oracle.py."""
import numpy as np
import pytest

from oracle import trtools_oracle as orc
from test_qc_oracle import CASES, GOLD, check_against_golden

pytestmark = pytest.mark.gpu


def random_batch(rng, L, S, P, haploid_frac=0.0, miss=0.1):
    gt = rng.integers(0, 4, size=(L, S, P)).astype(np.int16)
    gt[rng.random((L, S, P)) < miss] = -1
    gt[rng.random((L, S)) < miss / 2] = -1          # fully missing calls
    lp = np.full(L, P, dtype=np.int32)
    if P > 1 and haploid_frac:
        for l in np.flatnonzero(rng.random(L) < haploid_frac):
            lp[l] = 1
            gt[l, :, 1:] = -2
    if P > 1:   # haploid samples inside diploid records (chrX): -2 padding in the last column
        hs = rng.random((L, S)) < 0.03
        gt[..., 1][hs] = -2
    q = rng.random((L, S)).astype(np.float32)
    q[rng.random((L, S)) < 0.05] = np.nan
    return gt, lp, q


def oracle_reduce(gt, lp, q, sample_index, ignore):
    L, S, _ = gt.shape
    sel = np.ones(S, dtype=bool) if sample_index is None else sample_index
    out = dict(sample_calls=np.zeros(S, dtype=np.int64), locus_calls=np.zeros(L, dtype=np.int64),
               sample_qual_sum=np.zeros(S), sample_qual_n=np.zeros(S, dtype=np.int64), locus_qual_sum=np.zeros(L),
               locus_qual_n=np.zeros(L, dtype=np.int64))
    for l in range(L):
        calls, qq, _ = orc.qc_record(gt[l][:, :lp[l]], None if q is None else q[l].reshape(-1, 1), sel, ignore)
        out['sample_calls'][sel] += calls
        out['locus_calls'][l] = calls.sum()
        if q is not None:
            v = qq.reshape(-1).astype(np.float64)
            ok = ~np.isnan(v)
            out['sample_qual_sum'][np.flatnonzero(sel)[ok]] += v[ok]
            out['sample_qual_n'][np.flatnonzero(sel)[ok]] += 1
            out['locus_qual_sum'][l] = v[ok].sum()
            out['locus_qual_n'][l] = ok.sum()
    return out


@pytest.mark.parametrize('L,S,P', [(37, 64, 2), (300, 1000, 2), (129, 203, 2), (50, 77, 1), (40, 90, 3), (2100, 512, 2)])
@pytest.mark.parametrize('mode', ['plain', 'ignore', 'subset', 'subset_ignore', 'noquality'])
def test_device_equals_oracle(L, S, P, mode):
    from trtools_amd.engine import Engine
    rng = np.random.default_rng(L * 1000 + S + len(mode))
    gt, lp, q = random_batch(rng, L, S, P, haploid_frac=0.2 if P == 2 else 0.0)
    sel = (rng.random(S) < 0.6) if 'subset' in mode else None
    ignore = 'ignore' in mode
    if mode == 'noquality':
        q = None
    eng = Engine(0)
    A = 4
    off = np.arange(L + 1, dtype=np.int32) * A
    cls = np.tile(np.arange(A, dtype=np.uint16), L)
    b = eng.make_batch(gt, off, cls, cls, np.tile(np.arange(A, dtype=np.float64), L),
                       locus_ploidy=None if np.all(lp == P) else lp, max_alleles=A)
    res = eng.qc_reduce(b, q, None if sel is None else sel.astype(np.uint8), ignore)
    want = oracle_reduce(gt, lp, q, sel, ignore)
    for k in ('sample_calls', 'locus_calls'):
        assert np.array_equal(res[k].get(), want[k]), k
    if q is not None:
        for k in ('sample_qual_n', 'locus_qual_n'):
            assert np.array_equal(res[k].get(), want[k]), k
        for k in ('sample_qual_sum', 'locus_qual_sum'):   # float32 values added in float64: order changes the last bits
            np.testing.assert_allclose(res[k].get(), want[k], rtol=1e-12, atol=1e-12, err_msg=k)
    eng.close()


def test_empty_batch_and_repeatability():
    from trtools_amd.engine import Engine
    eng = Engine(0)
    rng = np.random.default_rng(7)
    gt, lp, q = random_batch(rng, 500, 2000, 2)
    off = np.arange(501, dtype=np.int32) * 4
    cls = np.tile(np.arange(4, dtype=np.uint16), 500)
    b = eng.make_batch(gt, off, cls, cls, np.tile(np.arange(4, dtype=np.float64), 500), max_alleles=4)
    r1 = eng.qc_reduce(b, q)
    r2 = eng.qc_reduce(b, q)
    for k in ('sample_qual_sum', 'locus_qual_sum', 'sample_calls', 'locus_calls'):
        assert np.array_equal(r1[k].get(), r2[k].get()), k        # fixed summation order: bit-identical reruns
    eng.close()


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_product_reductions_equal_reference_goldens(case):
    import os
    from trtools_amd.qcSTR import qc_reductions
    got = qc_reductions(os.path.join(GOLD, case['vcf']), vcftype=case['vcftype'],
                        samples=os.path.join(GOLD, case['samples']) if case['samples'] else None,
                        quality=case['quality'], quality_ignore_no_call=case['ignore_no_call'], batch_loci=300)
    assert got is not None
    # counts exact; the quality means to float32 rounding (the reference takes a float32 pairwise mean per locus and
    # adds float32 scores into a float64 total per sample; the device sums in float64)
    check_against_golden(case, got, got['samples'], float_tol=2e-6)
    want = case['recorded']
    assert got['n_alleles'] == want['diffref_hist']['n'] == want['diffref_bias']['n']
    np.testing.assert_allclose(got['sum_diff_unit'], want['diffref_hist']['sum'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(got['sum_diff_bp'], want['diffref_bias']['sum_diffs'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(got['sum_reflen_bp'], want['diffref_bias']['sum_reflens'], rtol=1e-9, atol=1e-6)
    # the joint (reference length, period, difference) distribution reproduces the recorded sums exactly: it is the
    # data of both diff-from-reference plots (four scalars could not rebuild the per-bin medians)
    h = got['diff_ref_histogram']
    assert sum(h.values()) == got['n_alleles']
    np.testing.assert_allclose(sum(d * k for (_, _, d), k in h.items()), want['diffref_hist']['sum'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(sum(d * p * k for (_, p, d), k in h.items()), want['diffref_bias']['sum_diffs'],
                               rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(sum(r * k for (r, _, _), k in h.items()), want['diffref_bias']['sum_reflens'],
                               rtol=1e-9, atol=1e-6)
