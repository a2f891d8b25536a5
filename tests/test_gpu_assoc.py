"""GPU parity: trk_assoc_scan (HIP, through the C ABI) vs the associaTR oracle on seeded synthetic
batches.  Integers / filter decisions bit-exact, floats within 1e-9 (p-values: 1e-9 relative)."""
import os
import sys

import numpy as np
import pytest

from trtools_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def make_case(seed, L, S, P=2, M=1, subset=False, max_alleles=12, miss=0.04, frac_len=True, locus_ploidy=False):
    """Random loci: lengths with duplicates (same length, different sequence) and rounding collisions,
    missing / partial calls, an all-missing locus, a monomorphic locus, an all-heterozygous locus."""
    rng = np.random.default_rng(seed)
    lens, gts = [], []
    for l in range(L):
        A = int(rng.integers(1, max_alleles + 1))
        base = float(rng.integers(5, 30))
        ll = [base]
        for a in range(1, A):
            r = rng.random()
            if r < 0.15:
                ll.append(ll[int(rng.integers(0, len(ll)))])                 # same length again
            elif r < 0.3 and frac_len:
                ll.append(base + float(rng.integers(-3, 4)) + 0.001 * float(rng.integers(1, 4)))  # rounds onto a neighbour
            elif frac_len and r < 0.5:
                ll.append(base + float(rng.integers(-8, 9)) / 3.0)
            else:
                ll.append(base + float(rng.integers(-4, 8)))
        lens.append(ll)
        p = rng.dirichlet(np.full(A, 0.6))
        g = rng.choice(A, size=(S, P), p=p).astype(np.int16)
        m = rng.random((S, P)) < miss * 0.3
        g[m] = -1
        g[rng.random(S) < miss] = -1
        if P > 1 and l % 7 == 3:
            g[rng.random(S) < 0.1, P - 1] = -2                               # lower-ploidy samples
        if l == 1:
            g[:] = -1
        if l == 2:
            g[:] = 0
        if l == 3 and A > 1 and P == 2:
            g[:, 0], g[:, 1] = 0, 1
        gts.append(g)
    gt = np.stack(gts)
    lp = None
    if locus_ploidy:
        lp = rng.integers(1, P + 1, size=L).astype(np.uint8)
        for l in range(L):
            gt[l, :, lp[l]:] = -2
    traits = rng.normal(size=(S, M))
    traits[:, 0] += 0.05 * gt[min(5, L - 1), :, 0]
    if subset:
        traits[rng.random(S) < 0.05, 0] = np.nan
        keep = rng.random(S) < 0.8
    else:
        keep = None
    return lens, gt, lp, traits, keep


def run_case(eng, seed, L, S, P=2, M=1, subset=False, cutoff=3.0, precision=2, **kw):
    from oracle import associatr_oracle as ao
    from trtools_amd import synth
    from trtools_amd import _lib as TL
    lens, gt, lp, traits, keep = make_case(seed, L, S, P, M, subset, **kw)
    names = ['s%d' % i for i in range(S)]
    sub = [n for n, k in zip(names, keep) if k] if keep is not None else None
    sf, covars, outcome, pheno_std = ao.prepare_design(names, [traits], True, sub)
    # device inputs
    off, lc, sc, cv = synth.pack_alleles(lens, None)
    alen, rcls = synth.pack_assoc_tables(lens, precision)
    vec = np.zeros((M, S))
    vec[0, sf] = outcome
    for k in range(1, M):
        vec[k, sf] = covars[:, 1 + k]
    batch = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp)
    res = eng.assoc_scan(batch, vec, alen, rcls, sample_in=None if sf.all() else sf.astype(np.uint8),
                         non_major_cutoff=cutoff)
    li, lf, cnt = res.locus_int.get(), res.locus_f64.get(), res.allele_count.get()
    status_of = {None: TL.AS_OK, 'No called samples': TL.AS_NO_CALLED, 'Only one called allele': TL.AS_ONE_ALLELE,
                 'n covars >= n samples': TL.AS_N_COVARS}
    n_ok = 0
    for l in range(L):
        g = gt[l] if lp is None else gt[l][:, :lp[l]]
        gi = ao.locus_genotypes(g, lens[l], sf, cutoff, precision)
        reason = gi['locus_filtered']
        n_tested = int(np.sum(gi['called_samples_filter']))
        if not reason and covars.shape[1] >= n_tested:
            reason = 'n covars >= n samples'
        want = status_of.get(reason, TL.AS_NON_MAJOR)
        assert li[l, TL.AI_N_TESTED] == n_tested, (l, li[l], n_tested)
        # allele counts by index over the tested samples
        curr = sf & ~np.any(g == -1, axis=1)
        sel = g[curr]
        exp_cnt = np.bincount(sel[sel >= 0].astype(int), minlength=len(lens[l]))
        assert np.array_equal(cnt[off[l]:off[l + 1]], exp_cnt), (l, cnt[off[l]:off[l + 1]], exp_cnt)
        assert li[l, TL.AI_N_RALLELES] == len(gi['allele_frequency']), l
        if want == TL.AS_OK:
            summed = np.sum(gi['gts'], axis=1)
            if np.std(summed) <= 1e-9 * max(1.0, abs(np.mean(summed))):
                assert li[l, TL.AI_STATUS] == TL.AS_ZERO_VARIANCE, (l, li[l])
                continue
        assert li[l, TL.AI_STATUS] == want, (l, li[l], reason)
        if want != TL.AS_OK:
            assert np.isnan(lf[l, TL.AF_PVAL])
            continue
        r = ao.locus_regression(gi['gts'], gi['called_samples_filter'], covars, outcome, pheno_std)
        for col, key, tol in ((TL.AF_COEF, 'coef_std', 1e-9), (TL.AF_SE, 'se_std', 1e-9), (TL.AF_RSQUARED, 'rsquared', 1e-9),
                              (TL.AF_GT_STD, 'std', 1e-12), (TL.AF_TVALUE, 'tvalue', 1e-9), (TL.AF_PVAL, 'pval', 1e-9)):
            a, b = lf[l, col], r[key]
            assert abs(a - b) <= tol * max(abs(b), 1e-300) + (1e-12 if key in ('coef_std', 'tvalue', 'rsquared') else 0), \
                (l, key, a, b)
        assert lf[l, TL.AF_DF_RESID] == r['df_resid'], l
        assert abs(lf[l, TL.AF_GT_MEAN] - np.mean(np.sum(gi['gts'], axis=1))) < 1e-9
        n_ok += 1
    return n_ok


def test_fast_path_single_trait(eng):
    assert run_case(eng, 1, 120, 512, M=1) > 60


def test_fast_path_covariates_and_subset(eng):
    assert run_case(eng, 2, 90, 1000, M=4, subset=True) > 40
    assert run_case(eng, 3, 60, 768, M=10, subset=True, cutoff=0.0) > 30


def test_fast_path_sample_chunks(eng):
    # 16 vectors x 4096 samples do not fit one LDS chunk -> partial records over several chunks
    assert run_case(eng, 4, 40, 4096, M=16, subset=True, miss=0.02) > 20


def test_mfma_path_many_covariates(eng):
    # M >= 3 goes through k_assoc_scan_mfma (16 loci x 16/32 vector rows per MFMA tile)
    assert run_case(eng, 21, 70, 1000, M=3, subset=True) > 30
    assert run_case(eng, 22, 50, 2048, M=15) > 25              # 16 rows with the ones row: one tile
    assert run_case(eng, 23, 50, 1500, M=16, subset=True, miss=0.08) > 25   # two tiles
    assert run_case(eng, 24, 37, 772, M=31, subset=True) > 15  # loci not a multiple of 16, S not of 256
    assert run_case(eng, 25, 20, 516, M=20, cutoff=0.0, miss=0.5) > 3      # half the calls missing


def test_gram_correction_of_mostly_missing_loci_on_long_rows(eng):
    """k_assoc_gram_miss: a round of 64 bit words covers 4096 samples; with 90 % of the calls missing it holds more
    sample indices than the wave's list (1032), so the round is taken a quarter of the lanes at a time and the list is
    consumed in between -- on rows of 8192 samples, one to three 16-row tiles, against the oracle."""
    assert run_case(eng, 61, 24, 8192, M=6, miss=0.9) > 10
    assert run_case(eng, 62, 20, 8192, M=20, subset=True, miss=0.85) > 8
    assert run_case(eng, 63, 18, 4100, M=36, miss=0.8) > 8                   # (S not a multiple of 256 either)


def test_wide_designs_pairs_of_row_groups(eng):
    """More than 31 trait columns (the reference has no bound, associaTR.py:138-204): ONE pass of three or four
    16-row tiles for diploid batches whose rows are whole 16-byte chunks (round 3), pairs of 15-row groups through
    the same scan kernels otherwise; the whole design solved by a wave per locus."""
    assert run_case(eng, 41, 50, 1024, M=32, subset=True) > 20               # groups of 15, 15, 2
    assert run_case(eng, 42, 37, 772, M=45, subset=True, miss=0.1) > 15      # three full groups; loci % 16, S % 256
    assert run_case(eng, 43, 40, 1280, M=62, cutoff=0.0, miss=0.3) > 15      # five groups, ten pairs, 64 lanes
    assert run_case(eng, 44, 30, 333, P=2, M=40, subset=True) > 10           # S % 4 != 0: the generic scan kernel
    assert run_case(eng, 45, 30, 300, P=3, M=33) > 10                        # triploid: the generic scan kernel
    from trtools_amd._lib import TrkError
    with pytest.raises(TrkError):
        run_case(eng, 46, 10, 256, M=127)


def test_designs_beyond_one_pass_of_the_matrix_pipe(eng):
    """63 to 126 trait columns (round 4): always pairs of 15-row groups (up to nine groups, 36 passes), the whole design
    solved by one wavefront per locus with TWO rows of the normal matrix per lane (k_assoc_regress_wave<2>; the first
    size is the one where row 64 appears)."""
    assert run_case(eng, 47, 24, 512, M=63, miss=0.1) > 10
    assert run_case(eng, 48, 20, 1024, M=100, subset=True) > 8
    assert run_case(eng, 49, 16, 772, M=126, cutoff=0.0, miss=0.2) > 6
    assert run_case(eng, 50, 12, 300, P=3, M=70) > 4                          # triploid: the generic scan kernel


def test_wave_parallel_regression_for_narrow_designs_too(eng):
    L.set_option('TRK_AS_WAVE_REGRESS_MIN', '2')
    try:
        assert run_case(eng, 31, 60, 512, M=2, subset=True) > 25
        assert run_case(eng, 32, 40, 640, M=9, subset=True) > 15
    finally:
        L.set_option('TRK_AS_WAVE_REGRESS_MIN', None)


def test_lds_resident_kernels_for_more_than_two_vectors(eng):
    L.set_option('TRK_AS_MFMA_MIN', '99')
    try:
        assert run_case(eng, 2, 90, 1000, M=4, subset=True) > 40
        assert run_case(eng, 3, 60, 768, M=10, subset=True, cutoff=0.0) > 30
        assert run_case(eng, 4, 40, 4096, M=16, subset=True, miss=0.02) > 20
    finally:
        L.set_option('TRK_AS_MFMA_MIN', None)


def test_many_alleles_and_rounding(eng):
    assert run_case(eng, 5, 50, 640, M=2, max_alleles=60) > 25
    assert run_case(eng, 6, 30, 512, M=1, max_alleles=400, precision=10) > 10   # generic path (LUT too big)
    # the one / two-vector kernel with so many bins that fewer than four histogram copies fit the wave's table area
    # (scalar zeroing and fold), with and without a sample subset
    assert run_case(eng, 11, 40, 2048, M=1, max_alleles=200) > 20
    assert run_case(eng, 12, 40, 1024, M=2, max_alleles=240, subset=True) > 20
    assert run_case(eng, 13, 40, 1024, M=1, max_alleles=130, subset=True, miss=0.3) > 15


def test_generic_path_ploidy_and_alignment(eng):
    assert run_case(eng, 7, 40, 333, P=2, M=3, subset=True) > 15      # S % 4 != 0
    assert run_case(eng, 8, 40, 200, P=3, M=2) > 15
    assert run_case(eng, 9, 40, 200, P=1, M=1, cutoff=0.0) > 15
    assert run_case(eng, 10, 40, 256, P=3, M=2, locus_ploidy=True, cutoff=0.0) > 10


def test_n_covars_rule_and_empty(eng):
    assert run_case(eng, 11, 30, 12, M=12, cutoff=0.0, miss=0.0) == 0   # 13 columns >= 12 samples
    from trtools_amd import synth
    off, lc, sc, cv = synth.pack_alleles([], None)
    b = eng.make_batch(np.zeros((0, 8, 2), np.int16), off, lc, sc, cv)
    r = eng.assoc_scan(b, np.zeros((1, 8)), np.zeros(0), np.zeros(0, np.uint16))
    assert r.locus_int.get().shape == (0, 8)


def run_dosage_case(eng, seed, L_, S, M, amax, min_ok=None):
    """One random batch through trk_assoc_scan_dosage and through the oracle-backed seam."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_compute import OracleCompute
    from trtools_amd.compute import DeviceCompute
    from trtools_amd.batch import HostBatch
    from trtools_amd import _lib as TL
    lens, gt, lp, traits, keep = make_case(100 + seed, L_, S, 2, M, True, max_alleles=amax)
    rng = np.random.default_rng(200 + seed)
    K = max(len(x) for x in lens) - 1
    ap = []
    for _ in range(2):
        a = np.full((L_, S, max(K, 1)), np.nan, dtype=np.float32)
        for l in range(L_):
            k = len(lens[l]) - 1
            if k:
                p = rng.dirichlet(np.full(k + 1, 0.4), size=S)
                a[l, :, :k] = (p[:, 1:] * rng.uniform(0.9, 1.1, size=(S, 1))).astype(np.float32)
        ap.append(a)
    sf = keep & ~np.isnan(traits[:, 0])
    tr = traits[sf]
    tr = (tr - tr.mean(axis=0)) / tr.std(axis=0)
    vec = np.zeros((M, S))
    vec[:, sf] = tr.T
    hb = HostBatch(gt, np.full(L_, 2, dtype=np.uint8), lens, [[str(i) for i in range(len(x))] for x in lens])
    want, wcs, wls, _ = OracleCompute().assoc_dosage_batch(hb, vec, sf, ap[0], ap[1], 2)
    got, gcs, gls, _ = DeviceCompute(eng).assoc_dosage_batch(hb, vec, sf, ap[0], ap[1], 2)
    assert np.array_equal(got.locus_int[:, TL.AI_N_TESTED], want.locus_int[:, TL.AI_N_TESTED])
    assert np.array_equal(got.locus_int[:, TL.AI_STATUS], want.locus_int[:, TL.AI_STATUS])
    np.testing.assert_allclose(gcs, wcs, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gls, wls, rtol=1e-11, atol=1e-11)
    ok = want.locus_int[:, TL.AI_STATUS] == TL.AS_OK
    if min_ok is not None:
        assert ok.sum() > min_ok
    for col in (TL.AF_PVAL, TL.AF_COEF, TL.AF_SE, TL.AF_RSQUARED, TL.AF_GT_STD, TL.AF_TVALUE, TL.AF_DF_RESID):
        np.testing.assert_allclose(got.locus_f64[ok, col], want.locus_f64[ok, col], rtol=1e-9, atol=1e-12)


def test_dosage_scan_against_oracle(eng):
    """trk_assoc_scan_dosage vs the oracle on synthetic AP1/AP2 planes: many alleles (more than 8 alternates ->
    numpy's blocked float32 row sum), rounding collisions between alleles, sample subset, missing calls."""
    for seed, (L_, S, M, amax) in enumerate([(30, 300, 2, 4), (24, 257, 5, 14), (12, 128, 20, 3)]):
        run_dosage_case(eng, seed, L_, S, M, amax, min_ok=L_ // 3)
    # more than 31 columns (round 4): pairs of 15-row groups through the same kernels, two rows per lane from 63 on
    run_dosage_case(eng, 7, 16, 320, 40, 4, min_ok=4)
    run_dosage_case(eng, 8, 12, 512, 70, 3, min_ok=3)


def test_scans_on_two_queues_run_side_by_side(eng):
    """The scan's workspace (Gram, partial records, class counts, work counter) is per queue: two different cohorts
    scanned concurrently on queues 0 and 1, several rounds, return what each returns alone."""
    from trtools_amd.synth import SynthBatch, pack_assoc_tables
    cases = []
    for k, (Lc, S) in enumerate(((3000, 2000), (2500, 1000))):
        sb = SynthBatch(eng, Lc, S, seed=500 + k, planes=())
        alen, rcls = pack_assoc_tables(sb.loci.allele_lens, 2)
        y = np.random.default_rng(9 + k).normal(size=S)
        y = (y - y.mean()) / y.std()
        cases.append((sb, eng.upload(y[None, :].copy(), np.float64), eng.upload(alen, np.float64),
                      eng.upload(rcls, np.uint16)))
    alone = []
    for sb, v, al, rc in cases:
        r = eng.assoc_scan(sb.batch, v, al, rc, non_major_cutoff=5.0)
        alone.append((r.locus_int.get().copy(), r.locus_f64.get().copy()))
    outs = [None, None]
    for _ in range(6):
        for q, (sb, v, al, rc) in enumerate(cases):
            with eng.on_queue(q):
                outs[q] = eng.assoc_scan(sb.batch, v, al, rc, non_major_cutoff=5.0, out=outs[q])
    eng.sync()
    for q in range(2):
        assert np.array_equal(outs[q].locus_int.get(), alone[q][0]), q
        assert np.array_equal(outs[q].locus_f64.get(), alone[q][1], equal_nan=True), q
