"""Pin the C half of the oracle against the numpy/scipy oracle (which is pinned
against the real reference).  CPU only."""
import collections
import math

import numpy as np
import pytest

from oracle import trtools_oracle as orc
from oracle import oracle_c
from helpers import load_golden, unjf, close


def test_c_binomtest_matches_scipy_vectors():
    lib = oracle_c.load()
    for k, n, p, pv in load_golden('binomtest_vectors.json')['cases']:
        got = lib.orc_binomtest(k, n, p)
        assert close(got, unjf(pv), 1e-9, 1e-300), (k, n, p, got, pv)


def test_c_batch_stats_match_numpy_oracle():
    from trtools_amd.synth import make_loci, cells_numpy, pack_alleles
    for S, Lc in ((400, 60), (33, 40)):
        loci = make_loci(Lc, S, seed=5 + S)
        h = cells_numpy(5, loci, np.arange(Lc), S)
        off, lc, sc, cv = pack_alleles(loci.allele_lens, loci.allele_strs)
        cnt, oi, of = oracle_c.batch_stats(h['gt'], None, off, lc, sc, cv)
        for l in range(Lc):
            ol = orc.locus_stats(h['gt'][l], loci.allele_lens[l], loci.allele_strs[l], None, True)
            os_ = orc.locus_stats(h['gt'][l], loci.allele_lens[l], loci.allele_strs[l], None, False)
            assert np.array_equal(cnt[off[l]:off[l + 1]], ol['index_counts'])
            assert oi[l, 0] == ol['numcalled']
            for j, (o, key) in enumerate(((ol, 'thresh'), (ol, 'mean'), (ol, 'mode'), (ol, 'var'), (ol, 'het'),
                                          (os_, 'het'), (ol, 'entropy'), (os_, 'entropy'), (ol, 'hwep'),
                                          (os_, 'hwep'))):
                if key == 'hwep' and o['hwep_status'] != orc.HWE_OK:
                    assert oi[l, 5 if o is ol else 6] == o['hwep_status']
                    continue
                assert close(of[l, j], o[key], 1e-9, 1e-300), (l, key, of[l, j], o[key])


def test_c_call_filters_match_numpy_oracle():
    from trtools_amd.synth import make_loci, cells_numpy
    S, Lc = 120, 50
    loci = make_loci(Lc, S, seed=9)
    h = cells_numpy(9, loci, np.arange(Lc), S)
    gout, mask, counters, totaldp, dpmiss = oracle_c.call_filters_dpq(h['gt'], h['dp'], h['q'], 10, 55, 0.9)
    info = collections.OrderedDict([('numcalls', np.zeros(S, dtype=int)), ('totaldp', np.zeros(S)),
                                    ('a', np.zeros(S, dtype=int)), ('b', np.zeros(S, dtype=int)),
                                    ('c', np.zeros(S, dtype=int))])
    for l in range(Lc):
        dp, q = h['dp'][l].reshape(-1, 1), h['q'][l].reshape(-1, 1)
        outs = [('a', orc.filt_min_value(dp, 10)), ('b', orc.filt_max_value(dp, 55)), ('c', orc.filt_min_value(q, 0.9))]
        g2, _ = orc.apply_call_filters(h['gt'][l], outs, info, dp=dp)
        assert np.array_equal(g2, gout[l])
    assert np.array_equal(counters[0], info['numcalls'])
    for k, n in enumerate('abc'):
        assert np.array_equal(counters[1 + k], info[n])
    tot = totaldp.astype(float)
    tot[dpmiss > 0] = np.nan
    assert np.array_equal(np.isnan(tot), np.isnan(info['totaldp']))
    assert np.array_equal(tot[~np.isnan(tot)], info['totaldp'][~np.isnan(tot)])


def _assoc_case(seed, S, n_cov, subset):
    rng = np.random.default_rng(seed)
    Lc = 60
    lens, rows = [], []
    for l in range(Lc):
        A = int(rng.integers(1, 9))
        base = float(rng.integers(5, 30))
        al = [base] + [base + float(rng.integers(-6, 7)) + float(rng.choice([0.0, 0.0, 0.5, 0.004, 0.001, 0.25]))
                       for _ in range(A - 1)]
        lens.append(al)
        p = rng.dirichlet(np.full(A, 0.4))
        if l % 11 == 3:
            p = np.eye(A)[0] * 0.999 + 0.001 / A           # nearly monomorphic: the non-major-allele filter
            p /= p.sum()
        g = rng.choice(A, size=(S, 2), p=p).astype(np.int16)
        g[rng.random(S) < 0.05] = -1
        g[rng.random(S) < 0.01, 1] = -1                    # half-missing calls are not called
        if l % 17 == 5:
            g[:] = -1
        rows.append(g)
    gt = np.stack(rows)
    off = np.concatenate([[0], np.cumsum([len(a) for a in lens])]).astype(np.int32)
    alen = np.concatenate([np.array(a) for a in lens])
    cov = rng.normal(size=(S, n_cov))
    y = rng.normal(size=S) + 0.1 * gt[7, :, 0]
    sample_in = (rng.random(S) < 0.7) if subset else np.ones(S, dtype=bool)
    # standardisation over the regression set, as associaTR.py:192-204 does
    full = np.concatenate([y[:, None], cov], axis=1)[sample_in]
    full = (full - full.mean(axis=0)) / full.std(axis=0)
    x = np.zeros((S, 2 + n_cov))
    x[:, 1] = 1.0
    x[sample_in, 2:] = full[:, 1:]
    yy = np.zeros(S)
    yy[sample_in] = full[:, 0]
    return gt, off, alen, lens, x, yy, sample_in


@pytest.mark.parametrize('S,n_cov,subset,cutoff', [(300, 0, False, 20.0), (257, 2, True, 5.0), (64, 3, False, 1.0),
                                                   (1500, 1, True, 40.0)])
def test_c_association_scan_equals_the_numpy_oracle(S, n_cov, subset, cutoff):
    """oracle_c.c's orc_assoc_locus against oracle/associatr_oracle.py (itself pinned to the reference-generated
    tables and the plink fixtures): tested-sample counts and filter reasons exactly, p / coefficient / se / R^2 to
    1e-10 -- so that the GPU scan can be checked at EVERY locus of BASELINE configs[4]."""
    from oracle import associatr_oracle as ao
    gt, off, alen, lens, x, y, sample_in = _assoc_case(1000 + S + n_cov, S, n_cov, subset)
    oi, of = oracle_c.assoc_scan(gt, off, alen, x, y, sample_in=None if not subset else sample_in,
                                 non_major_cutoff=cutoff, n_threads=2)
    covars = x[sample_in]
    outcome = y[sample_in]
    reasons = {None: 0, 'No called samples': 1, 'Only one called allele': 2, 'n covars >= n samples': 4}
    n_reg = 0
    for l in range(gt.shape[0]):
        r = ao.scan_locus(gt[l], lens[l], sample_in, covars, outcome, 1.0, cutoff, 2)
        assert oi[l, 0] == r['n_tested'], l
        why = r['locus_filtered']
        want = 3 if (why and why.startswith('non-major')) else reasons[why]
        if oi[l, 1] == 5:
            continue                       # constant genotype over the tested samples: nothing to compare
        assert oi[l, 1] == want, (l, oi[l], why)
        if want == 0:
            n_reg += 1
            for j, key in enumerate(('pval', 'coef_std', 'se_std', 'rsquared')):
                # (R^2 = 1 - ssr / sst cancels near 0: float64 noise of the numpy side, 1e-15 absolute)
                assert abs(of[l, j] - r[key]) <= 1e-10 * abs(r[key]) + (1e-13 if key == 'rsquared' else 1e-300), \
                    (l, key, of[l, j], r[key])
    assert n_reg >= 20


def test_c_student_t_tail_against_scipy():
    import ctypes as C
    from scipy.stats import t as student
    lib = oracle_c.load()
    lib.orc_t_two_sided.restype = C.c_double
    lib.orc_t_two_sided.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(4)
    for df in (1, 2, 5, 30, 997, 9998, 250000):
        for tv in np.concatenate([[0.0, 1e-9, 0.3, 1.0, 2.5, 8.0, 20.0, 37.0], rng.uniform(0, 40, 20)]):
            want = 2 * student.sf(tv, df)
            got = lib.orc_t_two_sided(float(tv), float(df))
            assert abs(got - want) <= 2e-11 * want + 1e-300, (df, tv, got, want)
