"""Pin the oracle (oracle/trtools_oracle.py) against vectors produced by the
REAL reference (tools/gen_golden.py) and against the literal known answers in
the reference's own unit tests.  CPU only."""
import math

import numpy as np
import pytest

from oracle import trtools_oracle as orc
from helpers import load_golden, unjf, close


@pytest.fixture(scope='module')
def trvec():
    return load_golden('trrecord_vectors.json')['cases']


def _si(c):
    return None if c['sample_index'] is None else np.array(c['sample_index'], dtype=bool)


def _strs(c):
    return [c['ref']] + list(c['alts'])


def _lens(c):
    return [unjf(x) for x in c['allele_lens']]


def test_called_callrate_ploidy(trvec):
    for c in trvec:
        gt = np.array(c['gt'])
        assert orc.get_called_samples(gt).tolist() == c['called']
        assert orc.get_called_samples(gt, strict=False).tolist() == c['called_nonstrict']
        assert orc.get_call_rate(gt) == unjf(c['callrate'])
        assert orc.get_sample_ploidies(gt).tolist() == c['ploidies']


def test_allele_counts_freqs(trvec):
    for c in trvec:
        gt, si = np.array(c['gt']), _si(c)
        got = orc.get_allele_counts(gt, _lens(c), si)
        assert [[repr(float(k)), int(v)] for k, v in got.items()] == c['counts_len']
        got = orc.get_allele_counts(gt, _lens(c), si, index=True)
        assert [[str(int(k)), int(v)] for k, v in got.items()] == c['counts_idx']
        got = orc.get_allele_counts(gt, _strs(c), si)
        assert [[str(k), int(v)] for k, v in got.items()] == c['counts_str']
        got = orc.get_allele_freqs(gt, _lens(c), si)
        assert [[repr(float(k)), float(v)] for k, v in got.items()] == \
            [[k, unjf(v)] for k, v in c['freqs_len']]
        got = orc.get_allele_freqs(gt, _strs(c), si)
        assert [[str(k), float(v)] for k, v in got.items()] == \
            [[k, unjf(v)] for k, v in c['freqs_str']]
        mx = orc.get_max_allele(gt, _lens(c), si)
        assert close(mx, unjf(c['maxallele']), 0, 0)


def test_genotype_counts(trvec):
    for c in trvec:
        gt, si = np.array(c['gt']), _si(c)
        got = orc.get_genotype_counts(gt, _lens(c), si)
        assert [[[float(x) for x in k], int(v)] for k, v in got.items()] == \
            [[[unjf(x) for x in k], v] for k, v in c['gcounts_len']]
        got = orc.get_genotype_counts(gt, _strs(c), si)
        assert [[[str(x) for x in k], int(v)] for k, v in got.items()] == c['gcounts_str']


def test_statstr_columns(trvec):
    n_err = 0
    for c in trvec:
        gt, si, st = np.array(c['gt']), _si(c), c['statstr']
        for ul in (True, False):
            tag = 'len' if ul else 'str'
            o = orc.locus_stats(gt, _lens(c), _strs(c), si, use_length=ul, nalleles_thresh=0.1)
            assert orc.format_afreq(o['afreq']) == st['afreq_' + tag]
            assert orc.format_afreq(o['acount'], count=True) == st['acount_' + tag]
            assert o['nalleles'] == st['nalleles_' + tag]
            ref_h = st['hwep_' + tag]
            if 'raises' in ref_h:
                n_err += 1
                want = orc.HWE_VALUE_ERROR if ref_h['raises'] == 'ValueError' else orc.HWE_INDEX_ERROR
                assert o['hwep_status'] == want
            else:
                assert o['hwep_status'] == orc.HWE_OK
                assert close(o['hwep'], unjf(ref_h['ok']), 0, 0)
            assert close(o['het'], unjf(st['het_' + tag]), 0, 0)
            assert close(o['entropy'], unjf(st['entropy_' + tag]), 0, 0)
        assert close(o['thresh'], unjf(st['thresh']), 0, 0)
        assert close(o['mean'], unjf(st['mean']), 0, 0)
        assert close(o['mode'], unjf(st['mode']), 0, 0)
        assert close(o['var'], unjf(st['var']), 0, 0)
        assert o['numcalled'] == st['numcalled']
    assert n_err > 0  # the golden set covers the exception paths


# ---- literal known answers from the reference's own unit tests -------------

def test_reference_test_utils_known_answers():
    # utils/tests/test_utils.py:21-99
    assert orc.validate_allele_freqs({0: 0.5, 1: 0.5})
    assert not orc.validate_allele_freqs({})
    assert not orc.validate_allele_freqs({0: 0.5, 1: 0.6})
    assert orc.get_heterozygosity({0: 1}) == 0
    assert orc.get_heterozygosity({0: 0.5, 1: 0.5}) == 0.5
    assert math.isnan(orc.get_heterozygosity({0: 0.5, 1: 0.6}))
    assert round(orc.get_heterozygosity({0: 0.5, 1: 0.2, 2: 0.3}), 2) == 0.62
    assert orc.get_entropy({0: 1}) == 0
    assert orc.get_entropy({0: 0.5, 1: 0.5}) == 1
    assert math.isnan(orc.get_entropy({}))
    assert orc.get_mean({0: 1}) == 0
    assert orc.get_mean({0: 0.5, 1: 0.5}) == 0.5
    assert orc.get_mode({0: 1}) == 0
    assert orc.get_mode({0: 0.1, 1: 0.9}) == 1
    assert orc.get_variance({0: 1}) == 0
    assert orc.get_variance({0: 0.5, 1: 0.5}) == 0.25
    # HWE: test_utils.py:81-99
    afreqs = {0: 0.5, 1: 0.5}
    assert round(orc.get_hwe_binomial_test(afreqs, {(0, 0): 0, (0, 1): 100, (1, 1): 0}), 2) == 0.0
    assert round(orc.get_hwe_binomial_test(afreqs, {(0, 0): 50, (0, 1): 0, (1, 1): 50}), 2) == 0.0
    assert math.isnan(orc.get_hwe_binomial_test({0: 0.5, 1: 0.6}, {(0, 0): 1}))
    assert math.isnan(orc.get_hwe_binomial_test({0: 0.5, 1: 0.5}, {(0, 3): 1}))


def test_reference_test_trharmonizer_known_answers():
    # utils/tests/test_trharmonizer.py:53-137 dummy records, expected dicts 441-715
    gts = np.array([[0, 1], [1, 1], [1, 1], [1, 2], [2, 2], [0, -1]])
    lens = [3.0, 4.0, 6.0]
    strs = ["CAGCAGCAG", "CAGCAGCAGCAG", "CAGCAGCAGCAGCAGCAG"]
    assert orc.get_genotype_counts(gts, lens) == {(3, 4): 1, (4, 4): 2, (4, 6): 1, (6, 6): 1}
    assert orc.get_allele_counts(gts, lens) == {3: 2, 4: 6, 6: 3}
    assert orc.get_allele_counts(gts, strs) == {strs[0]: 2, strs[1]: 6, strs[2]: 3}
    assert orc.get_max_allele(gts, lens) == 6
    assert orc.get_called_samples(gts).tolist() == [True] * 5 + [False]
    assert orc.get_call_rate(gts) == pytest.approx(5 / 6)
    tri = np.array([[0, 0, -2], [0, 0, -2], [0, 0, -2], [0, 0, 0]])
    assert orc.get_genotype_counts(tri, [3.0]) == {(-2, 3, 3): 3, (3, 3, 3): 1}
    assert orc.get_allele_counts(tri, [3.0]) == {3: 9}
    assert orc.get_sample_ploidies(tri).tolist() == [2, 2, 2, 3]
    assert orc.get_allele_counts(np.array([[-1, -1]]), [3.0]) == {}
    assert math.isnan(orc.get_max_allele(np.array([[-1, -1]]), [3.0]))
    # sample_index subsets (test_trharmonizer.py:494-508)
    si = [0, 2, 3]
    assert orc.get_allele_counts(gts, lens, sample_index=si) == {3: 1, 4: 4, 6: 1}


def test_binomtest_vectors_are_scipy():
    import scipy.stats
    v = load_golden('binomtest_vectors.json')['cases']
    for k, n, p, pv in v[::7]:
        assert close(scipy.stats.binomtest(k, n=n, p=p).pvalue, unjf(pv), 1e-12, 0)
