"""GPU parity: trk_call_filters / trk_locus_filters (HIP through the C ABI) vs
the oracle restatement of dumpSTR.ApplyCallFilters / ApplyLocusFilters
(dumpSTR.py:613-973, filters.py).  Integer outputs bit-exact."""
import collections
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT_MIN = -2147483648


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _oracle_run(orc, gt, filt_fn, names, dp, locus_ploidy=None):
    """Loop the oracle over loci; returns sample_info, masked gt, mask bits."""
    Lc, S, P = gt.shape
    info = collections.OrderedDict()
    info['numcalls'] = np.zeros(S, dtype=int)
    info['totaldp'] = np.zeros(S, dtype=float)
    for n in names:
        info[n] = np.zeros(S, dtype=int)
    gout = np.empty_like(gt)
    mask = np.zeros((Lc, S), dtype=np.uint32)
    for l in range(Lc):
        pl = P if locus_ploidy is None else int(locus_ploidy[l])
        g = gt[l][:, :pl]
        outs = filt_fn(l, g)
        for k, (_, o) in enumerate(outs):
            mask[l] |= (~np.isnan(o)).astype(np.uint32) << np.uint32(k)
        mask[l] |= (~orc.get_called_samples(g)).astype(np.uint32) << np.uint32(31)
        mg, _ = orc.apply_call_filters(g, outs, info, dp=None if dp is None else dp[l].reshape(-1, 1))
        gout[l] = gt[l]
        gout[l][:, :pl] = mg
    return info, gout, mask


def _compare(info, gout, mask, res, names, S):
    cnt = res.sample_counters.get()
    assert np.array_equal(cnt[0], info['numcalls'])
    for k, n in enumerate(names):
        assert np.array_equal(cnt[1 + k], info[n]), n
    tot = res.sample_totaldp.get().astype(float)
    tot[res.sample_dp_missing.get() > 0] = np.nan
    assert np.array_equal(np.isnan(tot), np.isnan(info['totaldp']))
    ok = ~np.isnan(tot)
    assert np.array_equal(tot[ok], info['totaldp'][ok])
    assert np.array_equal(res.gt_out.get(), gout)
    assert np.array_equal(res.filter_mask.get(), mask)
    assert res.error.get()[0] == 0


@pytest.mark.parametrize("n_loci,n_samples", [(150, 1000), (33, 52), (20, 1003)])
def test_hipstr_shape_filters(eng, n_loci, n_samples):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, n_loci, n_samples, seed=77 + n_samples, planes=('dp', 'q', 'dstutter', 'dflankindel'))
    h = sb.host_rows(np.arange(n_loci))
    planes = [sb.dev['dp'], sb.dev['q'], sb.dev['dstutter'], sb.dev['dflankindel']]
    # BuildCallFilters order (dumpSTR.py:792-804): flank indel, stutter, min DP, max DP, min Q
    filters = [dict(op=L.F_RATIO_GT, plane_a=3, plane_b=0, thr=0.15),
               dict(op=L.F_RATIO_GT, plane_a=2, plane_b=0, thr=0.15),
               dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=55),
               dict(op=L.F_LT, plane_a=1, thr=0.9)]
    names = ['flank', 'stutter', 'mindp', 'maxdp', 'minq']

    def filt(l, g):
        dp = h['dp'][l].reshape(-1, 1)
        return [('flank', orc.filt_ratio_gt(h['dflankindel'][l].reshape(-1, 1), dp, 0.15)),
                ('stutter', orc.filt_ratio_gt(h['dstutter'][l].reshape(-1, 1), dp, 0.15)),
                ('mindp', orc.filt_min_value(dp, 10)), ('maxdp', orc.filt_max_value(dp, 55)),
                ('minq', orc.filt_min_value(h['q'][l].reshape(-1, 1), 0.9))]
    info, gout, mask = _oracle_run(orc, h['gt'], filt, names, h['dp'])
    res = eng.call_filters(sb.batch, planes, filters, dp_plane=0)
    _compare(info, gout, mask, res, names, n_samples)
    assert info['mindp'].sum() > 0 and info['minq'].sum() > 0 and info['stutter'].sum() > 0

    # locus filters on the masked genotypes (dumpSTR.py:917-973)
    b2 = sb.batch.with_gt(res.gt_out)
    st = eng.locus_stats(b2)
    for use_length in (False, True):
        bits, counters = eng.locus_filters(n_loci, st, min_callrate=0.9, min_hwep=0.01, min_het=0.1,
                                           max_het=0.8, use_length=use_length)
        loc = collections.OrderedDict((k, 0) for k in ['totalcalls', 'PASS', 'NO_CALLS_REMAINING',
                                                       'CALLRATE0.9', 'HWE0.01', 'HETLOW0.1', 'HETHIGH0.8'])
        want_bits = np.zeros(n_loci, dtype=np.uint32)
        bitof = {'CALLRATE0.9': 0, 'HWE0.01': 1, 'HETLOW0.1': 2, 'HETHIGH0.8': 3, 'NO_CALLS_REMAINING': 31}
        n_raise = 0
        for l in range(n_loci):
            try:
                _, nm = orc.apply_locus_filters(gout[l], sb.loci.allele_lens[l], sb.loci.allele_strs[l], loc,
                                                use_length=use_length, min_callrate=0.9, min_hwep=0.01,
                                                min_het=0.1, max_het=0.8)
            except ValueError:
                n_raise += 1   # the reference would crash here; the device reports it
                continue
            for x in nm:
                want_bits[l] |= np.uint32(1) << np.uint32(bitof[x])
        c = counters.get()
        if n_raise == 0:
            assert np.array_equal(bits.get(), want_bits)
            assert c[L.LC_TOTALCALLS] == loc['totalcalls'] and c[L.LC_PASS] == loc['PASS']
            assert c[L.LC_NO_CALLS] == loc['NO_CALLS_REMAINING']
            for nm, b in (('CALLRATE0.9', 0), ('HWE0.01', 1), ('HETLOW0.1', 2), ('HETHIGH0.8', 3)):
                assert c[L.LC_FILTER0 + b] == loc[nm], nm
        assert c[L.LC_HWE_ERRORS] == n_raise


def run_gangstr_popstr_case(eng, seed, Lc, S, layout, keep=None, thr=None, delta=False, require_hits=False,
                            with_low=False):
    """QEXP / RC / REPCN+REPCI / AD filters (filters.py:573-867) on a random diploid batch against the oracle.
    keep: which of the nine filters to apply (order kept); thr: (mindp, maxdp, het, hom, total, support)."""
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    rng = np.random.default_rng(seed)
    P, A = 2, 5
    t_mindp, t_maxdp, t_het, t_hom, t_total, t_supp = thr if thr is not None else (5, 35, 0.2, 0.2, 0.5, 2)
    gt = rng.integers(0, A, size=(Lc, S, P)).astype(np.int16)
    gt[rng.random((Lc, S)) < 0.1] = -1
    gt[rng.random((Lc, S)) < 0.03, 1] = -1
    gt[min(3, Lc - 1)] = -1                      # a locus without any call
    lp = None
    if with_low:                                 # haploid records inside the diploid batch
        lp = rng.integers(1, 3, size=Lc).astype(np.uint8)
        gt[lp == 1, :, 1] = -2
    lens = [[float(i + 2) for i in range(A)]] * Lc
    strs = [['AC' * (i + 2) for i in range(A)]] * Lc
    off, lc, sc, cv = pack_alleles(lens, strs)
    dp = rng.integers(0, 40, size=(Lc, S)).astype(np.int32)
    qexp = rng.dirichlet(np.ones(3), size=(Lc, S)).astype(np.float32)
    qexp[rng.random((Lc, S)) < 0.1] = -1
    rc = np.zeros((Lc, S, 4), dtype=np.int32)
    rem = dp.copy()
    for j in range(3):
        x = (rng.random((Lc, S)) * (rem + 1)).astype(np.int32)
        x[rng.random((Lc, S)) < 0.2] = 0
        rc[:, :, j] = np.minimum(x, rem)
        rem = rem - rc[:, :, j]
    rc[:, :, 3] = rem
    perm = rng.permuted(np.tile(np.arange(4), (Lc, S, 1)), axis=2)
    rc = np.take_along_axis(rc, perm, axis=2)
    repcn = rng.integers(2, 12, size=(Lc, S, 2)).astype(np.int32)
    lo = repcn - rng.integers(-1, 3, size=(Lc, S, 2))
    hi = repcn + rng.integers(-1, 3, size=(Lc, S, 2))
    repci = np.stack([lo[:, :, 0], hi[:, :, 0], lo[:, :, 1], hi[:, :, 1]], axis=2).astype(np.int32)
    ad = rng.integers(0, 8, size=(Lc, S, A)).astype(np.int32)
    nocall = np.any(gt == -1, axis=2)
    dp[nocall & (rng.random((Lc, S)) < 0.8)] = INT_MIN
    b = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp)
    if layout == 'planar':
        def up(a):   # Engine.upload_plane with the transposition forced (by default planes of <= 4 columns stay as they are)
            os.environ['TRK_CF_PLANARIZE'] = '1'
            try:
                return eng.upload_plane(a)
            finally:
                del os.environ['TRK_CF_PLANARIZE']
    elif layout == 'planarize':
        up = lambda a: eng.planarize(eng.upload(a))
    else:
        up = eng.upload
    planes = [up(dp), up(qexp), up(rc), up(repcn), up(repci), up(ad)]
    assert getattr(planes[2], 'planar', False) == (layout != 'interleaved')
    # BuildCallFilters order (dumpSTR.py:819-836, 867-872)
    filters = [dict(op=L.F_LT, plane_a=0, thr=t_mindp), dict(op=L.F_GT, plane_a=0, thr=t_maxdp),
               dict(op=L.F_CALLED_LT, plane_a=1, col_a=1, thr=t_het),
               dict(op=L.F_CALLED_LT, plane_a=1, col_a=2, thr=t_hom),
               dict(op=L.F_CALLED_SUM_LT, plane_a=1, col_a=1, col_a2=2, thr=t_total),
               dict(op=L.F_CALLED_EQ, plane_a=2, col_a=1, plane_b=0, col_b=0),
               dict(op=L.F_CALLED_SUM_EQ, plane_a=2, col_a=1, col_a2=3, plane_b=0, col_b=0),
               dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=3, plane_b=4),
               dict(op=L.F_AD_SUPPORT_LT, plane_a=5, thr=t_supp)]
    names = ['mindp', 'maxdp', 'het', 'hom', 'total', 'span', 'spanbound', 'badci', 'support']
    keep = list(range(9)) if keep is None else sorted(keep)
    filters = [filters[k] for k in keep]
    names = [names[k] for k in keep]

    def filt(l, g):
        d = dp[l].reshape(-1, 1)
        rcs = np.array([','.join(map(str, r)) for r in rc[l]])
        cis = np.array(['%d-%d,%d-%d' % tuple(r) for r in repci[l]])
        allf = [('mindp', lambda: orc.filt_min_value(d, t_mindp)), ('maxdp', lambda: orc.filt_max_value(d, t_maxdp)),
                ('het', lambda: orc.filt_gangstr_qexp(g, qexp[l], t_het, 'het')),
                ('hom', lambda: orc.filt_gangstr_qexp(g, qexp[l], t_hom, 'hom')),
                ('total', lambda: orc.filt_gangstr_qexp(g, qexp[l], t_total, 'total')),
                ('span', lambda: orc.filt_gangstr_span_only(g, rcs, d)),
                ('spanbound', lambda: orc.filt_gangstr_spanbound_only(g, rcs, d)),
                ('badci', lambda: orc.filt_gangstr_bad_ci(g, repcn[l], cis)),
                ('support', lambda: orc.filt_popstr_require_support(g, ad[l], t_supp))]
        return [(allf[k][0], allf[k][1]()) for k in keep]
    info, gout, mask = _oracle_run(orc, gt, filt, names, dp, locus_ploidy=lp)
    st = eng.locus_stats(b, count_only=True) if delta else None
    res = eng.call_filters(b, planes, filters, dp_plane=0, delta_stats=st)
    _compare(info, gout, mask, res, names, S)
    if delta:
        recount = eng.locus_stats(b.with_gt(res.gt_out), count_only=True)
        assert np.array_equal(st.allele_count.get(), recount.allele_count.get())
        cols = [L.LI_N_CALLED, L.LI_N_LOWPLOIDY, L.LI_N_HOM_LEN, L.LI_N_HOM_STR]
        assert np.array_equal(st.locus_int.get()[0][:, cols], recount.locus_int.get()[0][:, cols])
    if require_hits:
        for n in names:
            assert info[n].sum() > 0, n


@pytest.mark.parametrize("layout,S", [('interleaved', 96), ('planar', 96), ('interleaved', 1000), ('planar', 1000),
                                      ('planarize', 1003)])
def test_gangstr_and_popstr_shape_filters(eng, layout, S):
    """QEXP / RC / REPCN+REPCI / AD filters (filters.py:573-867).  Multi-column planes in both device layouts:
    interleaved [L, S, k] and planar [k, L, S] (TRK_DT_PLANAR; uploaded that way or transposed on the device by
    trk_planarize)."""
    run_gangstr_popstr_case(eng, 11, 40, S, layout, require_hits=True)


def test_general_ploidy_call_filters(eng):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    rng = np.random.default_rng(12)
    Lc, S, P, A = 17, 61, 3, 4
    gt = rng.integers(0, A, size=(Lc, S, P)).astype(np.int16)
    lp = rng.integers(1, P + 1, size=Lc).astype(np.uint8)
    for l in range(Lc):
        gt[l][:, lp[l]:] = -2
    gt[rng.random((Lc, S)) < 0.1, 0] = -1
    dp = rng.integers(0, 40, size=(Lc, S)).astype(np.int32)
    q = rng.random((Lc, S)).astype(np.float32)
    off, lc, sc, cv = pack_alleles([[1.0, 2.0, 3.0, 4.5]] * Lc, [['A', 'AA', 'AAA', 'AAAAC']] * Lc)
    b = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp)
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_LT, plane_a=1, thr=0.5)]

    def filt(l, g):
        return [('mindp', orc.filt_min_value(dp[l].reshape(-1, 1), 10)),
                ('minq', orc.filt_min_value(q[l].reshape(-1, 1), 0.5))]
    info, gout, mask = _oracle_run(orc, gt, filt, ['mindp', 'minq'], dp, locus_ploidy=lp)
    res = eng.call_filters(b, [eng.upload(dp), eng.upload(q)], filters, dp_plane=0)
    _compare(info, gout, mask, res, ['mindp', 'minq'], S)


def test_negative_dp_is_reported(eng):
    """dumpSTR.py:698-706: a PASS call with negative DP is a ValueError."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    gt = np.zeros((2, 8, 2), dtype=np.int16)
    dp = np.full((2, 8), 20, dtype=np.int32)
    dp[1, 5] = -3
    off, lc, sc, cv = pack_alleles([[1.0]] * 2, [['A']] * 2)
    b = eng.make_batch(gt, off, lc, sc, cv)
    res = eng.call_filters(b, [eng.upload(dp)], [dict(op=L.F_GT, plane_a=0, thr=100)], dp_plane=0)
    err = res.error.get()
    assert err[0] == 1 and err[1] == 1 and err[2] == 5


def test_float32_threshold_semantics(eng):
    """numpy compares a float32 FORMAT array with the threshold in float32:
    float32(0.9) < 0.9 is False there but True in float64."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    gt = np.zeros((1, 4, 2), dtype=np.int16)
    q = np.array([[0.9, 0.89999, 0.90001, np.nan]], dtype=np.float32)
    off, lc, sc, cv = pack_alleles([[1.0]], [['A']])
    b = eng.make_batch(gt, off, lc, sc, cv)
    res = eng.call_filters(b, [eng.upload(q)], [dict(op=L.F_LT, plane_a=0, thr=0.9)], dp_plane=-1)
    assert (res.filter_mask.get()[0] & 1).tolist() == [0, 1, 0, 0]
    assert (q[0] < 0.9).tolist() == [False, True, False, False]


def test_missing_dp_poisons_totaldp(eng):
    """dumpSTR.py:710-711: a PASS call whose DP is missing turns the sample's totaldp into nan."""
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    rng = np.random.default_rng(3)
    Lc, S = 12, 40
    gt = rng.integers(0, 2, size=(Lc, S, 2)).astype(np.int16)
    gt[rng.random((Lc, S)) < 0.1] = -1
    dp = rng.integers(1, 40, size=(Lc, S)).astype(np.int32)
    dp[rng.random((Lc, S)) < 0.05] = INT_MIN
    dp[rng.random((Lc, S)) < 0.05] = 0
    q = rng.random((Lc, S)).astype(np.float32)
    off, lc, sc, cv = pack_alleles([[1.0, 2.0]] * Lc, [['A', 'AA']] * Lc)
    b = eng.make_batch(gt, off, lc, sc, cv)

    def filt(l, g):
        return [('minq', orc.filt_min_value(q[l].reshape(-1, 1), 0.3))]
    info, gout, mask = _oracle_run(orc, gt, filt, ['minq'], dp)
    res = eng.call_filters(b, [eng.upload(dp), eng.upload(q)], [dict(op=L.F_LT, plane_a=1, thr=0.3)], dp_plane=0)
    _compare(info, gout, mask, res, ['minq'], S)
    assert np.isnan(info['totaldp']).any() and (~np.isnan(info['totaldp'])).any()


@pytest.mark.parametrize("n_samples", [1000, 1003])
def test_delta_counts_equal_recount(eng, n_samples):
    """trk_call_out.delta_*: counts(GT) minus the filtered calls == counts(GT') recomputed from gt_out
    (streaming kernel with the LDS delta table for S % 4 == 0, per-call kernel otherwise)."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, 300, n_samples, seed=5 + n_samples)
    planes = [sb.dev['dp'], sb.dev['q']]
    filters = [dict(op=L.F_LT, plane_a=0, thr=25), dict(op=L.F_LT, plane_a=1, thr=0.95)]
    st = eng.locus_stats(sb.batch, count_only=True)
    res = eng.call_filters(sb.batch, planes, filters, dp_plane=0, delta_stats=st)
    eng.locus_finalize(sb.batch, st)
    ref = eng.locus_stats(sb.batch.with_gt(res.gt_out))
    assert np.array_equal(st.allele_count.get(), ref.allele_count.get())
    assert np.array_equal(st.locus_int.get(), ref.locus_int.get())
    a, b = st.locus_f64.get(), ref.locus_f64.get()
    assert np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))
    assert (res.filter_mask.get() & np.uint32(3)).any()


def test_no_filters_gives_per_sample_call_counts_and_float_depth(eng):
    """n_filters == 0: sample_counters[0] is the per-sample number of calls (qcSTR's sample_calls), the genotypes
    pass through; a Float depth plane (ExpansionHunter's LC) is summed in float64 (sample_totaldp_f64)."""
    from trtools_amd import synth
    rng = np.random.default_rng(9)
    Lc, S = 37, 130
    lens = [[10.0, 11.0, 12.0]] * Lc
    off, lc, sc, cv = synth.pack_alleles(lens, None)
    gt = rng.integers(-1, 3, size=(Lc, S, 2)).astype(np.int16)
    lcov = rng.uniform(0, 60, size=(Lc, S, 1)).astype(np.float32)
    lcov[rng.random((Lc, S, 1)) < 0.05] = np.nan
    b = eng.make_batch(gt, off, lc, sc, cv)
    res = eng.call_filters(b, [eng.upload(lcov)], [], dp_plane=0)
    called = ~np.any(gt == -1, axis=2)
    assert np.array_equal(res.sample_counters.get()[0], called.sum(axis=0))
    assert np.array_equal(res.gt_out.get(), gt)
    assert np.array_equal(res.filter_mask.get() >> 31, (~called).astype(np.uint32))
    want = np.where(called & (lcov[:, :, 0] > 0), lcov[:, :, 0].astype(np.float64), 0.0).sum(axis=0)
    assert np.allclose(res.sample_totaldp_f64.get(), want, rtol=1e-14, atol=0)
    assert not res.sample_totaldp.get().any()


def test_ratio_filters_decide_like_the_float64_division(eng):
    """HipSTR flank-indel / stutter filters (filters.py:415-484: DSTUTTER / DP > thr in float64) on the lean
    streaming kernel: exact ties (3/20 vs 0.15), zero and missing depths, missing numerators, negative values and odd
    thresholds must all come out as numpy has them."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    rng = np.random.default_rng(5)
    Lc, S = 24, 4096
    gt = rng.integers(0, 2, size=(Lc, S, 2)).astype(np.int16)
    off, lc, sc, cv = pack_alleles([[2.0, 3.0]] * Lc, [['ACAC', 'ACACAC']] * Lc)
    b = eng.make_batch(gt, off, lc, sc, cv)
    # (round 5: the streaming kernel decides by the sign of x - t |d| and divides only inside one ulp above the threshold:
    # thresholds ON the rounded quotient of a ratio the data is full of, and one ulp either side of it)
    cases = [(t, 3, 20) for t in (0.15, 0.2, 1.0 / 3.0, 0.0, -0.5, 1.0, 1e-300, 3.0, float('inf'), float('-inf'),
                                  0.15000000000000002, float('nan'))]
    for n_, d_ in ((7, 13), (1, 3), (33, 97), (5, 7), (1, 10), (-4, 9), (0, 5)):
        q = n_ / d_
        cases += [(q, n_, d_), (float(np.nextafter(q, np.inf)), n_, d_), (float(np.nextafter(q, -np.inf)), n_, d_)]
    for thr, n_, d_ in cases:
        dp = rng.integers(1, 200, size=(Lc, S)).astype(np.int32)
        num = rng.integers(0, 40, size=(Lc, S)).astype(np.int32)
        # exact ties and near ties of the threshold
        k = rng.integers(1, 9, size=(Lc, S))
        k[rng.random((Lc, S)) < 0.2] *= -1               # (-n) / (-d): the same quotient through a negative depth
        tie = rng.random((Lc, S)) < 0.3
        dp[tie] = (d_ * k)[tie]
        num[tie] = (n_ * k)[tie]
        dp[rng.random((Lc, S)) < 0.02] = 0
        dp[rng.random((Lc, S)) < 0.02] = INT_MIN
        num[rng.random((Lc, S)) < 0.02] = INT_MIN
        num[rng.random((Lc, S)) < 0.02] *= -1
        dp[rng.random((Lc, S)) < 0.01] *= -1
        with np.errstate(divide='ignore', invalid='ignore'):
            want = (num.astype(np.float64) / dp.astype(np.float64)) > thr
        filters = [dict(op=L.F_RATIO_GT, plane_a=1, plane_b=0, thr=thr), dict(op=L.F_LT, plane_a=0, thr=-1e9)]
        res = eng.call_filters(b, [eng.upload(dp), eng.upload(num)], filters, dp_plane=0)
        got = (res.filter_mask.get() & np.uint32(1)).astype(bool)
        assert np.array_equal(got, want), (thr, n_, d_)


@pytest.mark.parametrize("max_alt,S", [(130, 512), (900, 256), (3000, 64)])
def test_delta_outputs_with_very_large_allele_sets(eng, max_alt, S):
    """Loci with hundreds / thousands of alleles: the LDS delta tables of the streaming kernels do not fit, the
    interpreter with smaller blocks or the per-call kernel with global atomics takes over -- delta-corrected counts
    must still equal a recount of the masked genotypes."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    rng = np.random.default_rng(max_alt)
    Lc = 12
    lens, strs, gts = [], [], []
    for l in range(Lc):
        A = max_alt + 1 if l % 3 == 0 else int(rng.integers(1, 6))
        strs.append(['AC' * (i + 1) for i in range(A)])
        lens.append([float(i + 1) for i in range(A)])
        g = rng.integers(0, A, size=(S, 2)).astype(np.int16)
        g[rng.random(S) < 0.1] = -1
        gts.append(g)
    gt = np.stack(gts)
    off, lc, sc, cv = pack_alleles(lens, strs)
    b = eng.make_batch(gt, off, lc, sc, cv)
    dp = rng.integers(0, 50, size=(Lc, S)).astype(np.int32)
    q = rng.random((Lc, S)).astype(np.float32)
    filters = [dict(op=L.F_LT, plane_a=0, thr=15), dict(op=L.F_LT, plane_a=1, thr=0.4)]
    st = eng.locus_stats(b, count_only=True)
    res = eng.call_filters(b, [eng.upload(dp), eng.upload(q)], filters, dp_plane=0, delta_stats=st)
    recount = eng.locus_stats(b.with_gt(res.gt_out), count_only=True)
    assert np.array_equal(st.allele_count.get(), recount.allele_count.get())
    cols = [L.LI_N_CALLED, L.LI_N_LOWPLOIDY, L.LI_N_HOM_LEN, L.LI_N_HOM_STR]
    assert np.array_equal(st.locus_int.get()[0][:, cols], recount.locus_int.get()[0][:, cols])
    want_mask = ((dp < 15).astype(np.uint32) | ((q < np.float32(0.4)).astype(np.uint32) << 1))
    want_mask |= np.any(gt == -1, axis=2).astype(np.uint32) << np.uint32(31)
    assert np.array_equal(res.filter_mask.get(), want_mask)


@pytest.mark.parametrize("n_samples,n_filters", [(1000, 3), (1000, 7), (1003, 2), (1000, 9)])
def test_compact_mask_output(eng, n_samples, n_filters):
    """trk_call_out.filter_mask8: one byte per call (bit k = filter k, bit 7 = no-call) next to -- or instead of -- the
    32-bit mask and the masked genotypes, on the threshold kernel, the interpreter kernel (a GangSTR-style filter in
    the set) and the per-call kernel (unaligned rows); counters and delta outputs do not depend on which outputs are
    asked for.  More than 7 filters: the byte mask is not written."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, 200, n_samples, seed=77 + n_samples + n_filters, planes=('dp', 'q', 'dstutter'))
    planes = [sb.dev['dp'], sb.dev['q'], sb.dev['dstutter']]
    pool = [dict(op=L.F_LT, plane_a=0, thr=12), dict(op=L.F_GT, plane_a=0, thr=50), dict(op=L.F_LT, plane_a=1, thr=0.93),
            dict(op=L.F_CALLED_LT, plane_a=2, thr=1), dict(op=L.F_GT, plane_a=2, thr=2),
            dict(op=L.F_CALLED_EQ, plane_a=2, plane_b=0), dict(op=L.F_LT, plane_a=0, thr=20),
            dict(op=L.F_GT, plane_a=1, thr=0.99), dict(op=L.F_LT, plane_a=2, thr=1)]
    filters = pool[:n_filters]
    st = eng.locus_stats(sb.batch, count_only=True)
    full = eng.call_filters(sb.batch, planes, filters, dp_plane=0, delta_stats=st)
    mask = full.filter_mask.get()
    st2 = eng.locus_stats(sb.batch, count_only=True)
    out = eng.alloc_call_out(sb.batch, n_filters, want_gt=False, want_mask=False, want_mask8=True)
    out.filter_mask8.set(np.full((200, n_samples), 0x55, dtype=np.uint8))
    eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st2)
    m8 = out.filter_mask8.get()
    if n_filters <= 7:
        want = ((mask & np.uint32(0x7f)) | ((mask >> np.uint32(24)) & np.uint32(0x80))).astype(np.uint8)
        assert np.array_equal(m8, want) and (m8 & 0x80).any() and (m8 & 0x7f).any()
    else:
        assert np.all(m8 == 0x55)
    assert np.array_equal(out.sample_counters.get(), full.sample_counters.get())
    assert np.array_equal(out.sample_totaldp.get(), full.sample_totaldp.get())
    assert np.array_equal(st2.allele_count.get(), st.allele_count.get())
    assert np.array_equal(st2.locus_int.get()[..., :6], st.locus_int.get()[..., :6])


@pytest.mark.parametrize("shape", ["hipstr3", "hipstr5", "gangstr", "percall"])
def test_in_place_masked_genotypes(eng, shape):
    """gt_out == trk_batch.gt: the genotype tensor is updated where it lies (the reference sets the filtered calls of
    its record to no-call, dumpSTR.py:721-727).  The streaming kernels store only the chunks that hold a filtered call;
    every cell, the mask, the counters and the delta counts equal the two-plane run's -- at every locus."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    S = 1003 if shape == "percall" else 2052
    n_loci = 300
    if shape == "gangstr":
        sb = SynthBatch(eng, n_loci, S, seed=5, planes=('dp', 'q'), pure_repeats=True)
        sb.add_gangstr_planes()
        planes = [eng.planarize(sb.dev[n]) for n in ('dp', 'q', 'qexp', 'rc', 'repcn', 'repci')]
        filters = [dict(op=L.F_LT, plane_a=0, thr=12), dict(op=L.F_GT, plane_a=0, thr=55), dict(op=L.F_LT, plane_a=1, thr=0.9),
                   dict(op=L.F_CALLED_LT, plane_a=2, col_a=1, thr=0.05), dict(op=L.F_CALLED_EQ, plane_a=3, col_a=1, plane_b=0, col_b=0),
                   dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=4, plane_b=5)]
    else:
        sb = SynthBatch(eng, n_loci, S, seed=5, planes=('dp', 'q', 'dstutter', 'dflankindel'))
        planes = [sb.dev['dp'], sb.dev['q'], sb.dev['dstutter'], sb.dev['dflankindel']]
        filters = [dict(op=L.F_LT, plane_a=0, thr=12), dict(op=L.F_GT, plane_a=0, thr=55), dict(op=L.F_LT, plane_a=1, thr=0.9)]
        if shape == "hipstr5":
            filters += [dict(op=L.F_RATIO_GT, plane_a=2, plane_b=0, thr=0.12), dict(op=L.F_RATIO_GT, plane_a=3, plane_b=0, thr=0.08)]
    st = eng.locus_stats(sb.batch, count_only=True)
    two = eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=eng.alloc_call_out(sb.batch, len(filters), place=False),
                           delta_stats=st)
    want_gt, want_mask = two.gt_out.get(), two.filter_mask.get()
    before = sb.batch.arrays['gt'].get()
    assert (want_gt != before).any()
    st2 = eng.locus_stats(sb.batch, count_only=True)
    out = eng.alloc_call_out(sb.batch, len(filters), in_place=True)
    assert out.gt_out.ptr == sb.batch.arrays['gt'].ptr
    eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out, delta_stats=st2)
    assert np.array_equal(sb.batch.arrays['gt'].get(), want_gt)
    assert np.array_equal(out.filter_mask.get(), want_mask)
    assert np.array_equal(out.sample_counters.get(), two.sample_counters.get())
    assert np.array_equal(out.sample_totaldp.get(), two.sample_totaldp.get())
    assert np.array_equal(st2.allele_count.get(), st.allele_count.get())
    assert np.array_equal(st2.locus_int.get()[..., :6], st.locus_int.get()[..., :6])
    # the counts of the updated tensor are the delta-corrected counts (dumpSTR.py:748-774 rebuilds its record)
    recount = eng.locus_stats(sb.batch, count_only=True)
    assert np.array_equal(recount.allele_count.get(), st2.allele_count.get())
    # a second pass over the updated tensor filters nothing further among the calls that are left
    out3 = eng.alloc_call_out(sb.batch, len(filters), in_place=True)
    eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out3)
    assert np.array_equal(sb.batch.arrays['gt'].get(), want_gt)


def test_placed_output_planes_hold_the_same_results(eng):
    """Engine.alloc_call_out from 256 MB planes on (trk_dev_alloc_pair): the second big output plane is the best of at
    most three candidates (write-only probe of the pass's stream shape with the first plane), at most two spare planes
    exist during the search and none after it; the pass writes into the placed planes what it writes into plainly
    allocated ones."""
    from trtools_amd import _lib as L
    from trtools_amd.engine import Engine
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, 8192, 8192, seed=11, planes=('dp', 'q'))
    planes = [sb.dev['dp'], sb.dev['q']]
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=55), dict(op=L.F_LT, plane_a=1, thr=0.9)]
    plain = eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=eng.alloc_call_out(sb.batch, len(filters), place=False))
    eng.trim()                                        # (an empty pool: the placed allocation is a fresh one)
    live_before = len(eng._live)
    Engine.last_placement = None
    out = eng.alloc_call_out(sb.batch, len(filters))
    seen = Engine.last_placement
    assert seen and 1 <= len(seen["probe_ms"]) <= 7 and seen['kept_ms'] == min(seen['probe_ms'])
    assert seen['peak_extra_bytes'] <= max(2 * seen['plane_bytes'], (16 << 30) + seen['plane_bytes']) and seen['seconds'] < 5.0
    assert len(eng._live) - live_before == 7          # the seven arrays of one CallResult: the other candidates are gone
    tuned = eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out)
    assert np.array_equal(tuned.gt_out.get(), plain.gt_out.get())
    assert np.array_equal(tuned.filter_mask.get(), plain.filter_mask.get())
    assert np.array_equal(tuned.sample_counters.get(), plain.sample_counters.get())


def _reserved_pair_checks():
    """trk_reserve_pair (Engine(reserve_pair_gb=...)): the context takes the two output planes as its first device
    allocations and trk_dev_alloc_pair lends them to whoever asks for a pair that fits; a freed plane goes back to the
    context (not to the driver, not to the engine's pool), a second pair while the first is out is searched as before;
    the pass writes into the lent planes what it writes into plain ones."""
    from trtools_amd import _lib as L
    from trtools_amd.engine import Engine
    from trtools_amd.synth import SynthBatch
    e2 = Engine(0, reserve_pair_gb=0.375)
    try:
        assert Engine.last_reservation['plane_bytes'] == 384 << 20 and len(Engine.last_reservation['probe_ms']) <= 8
        # a pair that is not fast is not kept (round 6): nothing of the eight candidate planes stays behind
        assert e2.reserved_pair_bytes == ((384 << 20) if Engine.last_reservation['fast'] else 0)
        sb = SynthBatch(e2, 8192, 8192, seed=11, planes=('dp', 'q'))
        planes = [sb.dev['dp'], sb.dev['q']]
        filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=55), dict(op=L.F_LT, plane_a=1, thr=0.9)]
        plain = e2.call_filters(sb.batch, planes, filters, dp_plane=0, out=e2.alloc_call_out(sb.batch, len(filters), place=False))
        out = e2.alloc_call_out(sb.batch, len(filters))
        if not Engine.last_reservation['fast']:
            # (this process's first allocations were made long ago: no fast pair among eight planes is the usual outcome
            # here) -- the pair is searched as if nothing had been reserved, and the pass writes the same
            assert not Engine.last_placement['reserved']
            srch = e2.call_filters(sb.batch, planes, filters, dp_plane=0, out=out)
            assert np.array_equal(srch.gt_out.get(), plain.gt_out.get()) and np.array_equal(srch.filter_mask.get(), plain.filter_mask.get())
            return 'searched'
        assert Engine.last_placement['reserved'] and len(Engine.last_placement['probe_ms']) == 1
        ptrs = (out.gt_out.ptr, out.filter_mask.ptr)
        lent = e2.call_filters(sb.batch, planes, filters, dp_plane=0, out=out)
        assert np.array_equal(lent.gt_out.get(), plain.gt_out.get())
        assert np.array_equal(lent.filter_mask.get(), plain.filter_mask.get())
        other = e2.alloc_call_out(sb.batch, len(filters))          # the pair is out: a searched pair
        assert not Engine.last_placement['reserved'] and other.gt_out.ptr not in ptrs
        pool_before = e2._pool_bytes
        out.gt_out.free()
        out.filter_mask.free()
        assert e2._pool_bytes == pool_before                       # handed back to the context, not pooled
        again = e2.alloc_call_out(sb.batch, len(filters))
        assert Engine.last_placement['reserved'] and (again.gt_out.ptr, again.filter_mask.ptr) == ptrs
        small = SynthBatch(e2, 64, 512, seed=3, planes=('dp', 'q'))  # below 256 MB: plain allocations, pair untouched
        o3 = e2.alloc_call_out(small.batch, 3)
        assert o3.gt_out.ptr not in ptrs
    finally:
        e2.close()
    return 'lent' if Engine.last_reservation['fast'] else 'searched'


def test_reserved_pair_is_lent_and_handed_back():
    """The checks above in THIS process (its first allocations were made long ago: usually no fast pair among the eight
    candidate planes -- nothing is kept, the pair is searched) and in a FRESH one, where the reservation is what it is
    meant to be, the process's first device allocations (a fast pair within eight planes in every fresh process seen:
    profiles/r05_class_probe.txt) -- the lending path is exercised there."""
    import subprocess
    import sys
    _reserved_pair_checks()
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_callfilters as t; "
            "print('RESERVED-PAIR', t._reserved_pair_checks())" % (os.path.dirname(os.path.abspath(__file__)),
                                                                    os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, TRK_LAB='1'))
    out = r.stdout.decode()
    assert r.returncode == 0 and 'RESERVED-PAIR' in out, out[-3000:]


@pytest.mark.parametrize("seed", range(12))
def test_streaming_kernel_equals_the_per_call_kernel_on_random_filter_sets(eng, seed):
    """k_call_filter_v4 (round 4: static compare types, LT / GT folded into the operands, filters reordered by plane and
    type, queued delta updates) against the per-call kernel (TRK_CF_GENERIC=1, pinned to the oracle elsewhere) on filter
    sets it has to get right: 1-6 filters over two integer and two float planes in any order, LT / GT / called-only LT,
    fractional, negative and extreme thresholds, missing values and NaNs, with and without the delta outputs."""
    import os
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    rng = np.random.default_rng(1000 + seed)
    n_loci, S = int(rng.integers(40, 400)), int(rng.choice([256, 1000, 1024, 2052]))
    sb = SynthBatch(eng, n_loci, S, seed=50 + seed, planes=('dp', 'q'))
    dp = sb.dev['dp'].get()
    i2 = rng.integers(-5, 60, size=(n_loci, S)).astype(np.int32)
    i2[rng.random((n_loci, S)) < 0.04] = -2147483648
    f2 = (rng.normal(size=(n_loci, S)) * 3).astype(np.float32)
    f2[rng.random((n_loci, S)) < 0.04] = np.nan
    f2[rng.random((n_loci, S)) < 0.01] = -0.0
    planes = [sb.dev['dp'], sb.dev['q'], eng.upload(i2), eng.upload(f2)]
    thr_int = [0.0, 10.0, 10.5, -3.0, 54.999, 2147483000.0, -2147483000.0, 30.0]
    thr_flt = [0.9, 0.0, -0.0, 1.0, -1.5, 0.3333333, 1e30, -1e30]
    nf = int(rng.integers(1, 7))
    filters = []
    for _ in range(nf):
        p = int(rng.integers(0, 4))
        op = int(rng.choice([L.F_LT, L.F_GT, L.F_CALLED_LT]))
        filters.append(dict(op=op, plane_a=p, thr=float(rng.choice(thr_flt if p in (1, 3) else thr_int))))
    use_dp = bool(rng.integers(0, 2)) or np.any(dp < 0) is False
    outs = []
    for generic in (False, True):
        if generic:
            L.set_option('TRK_CF_GENERIC', '1')
        try:
            st = eng.locus_stats(sb.batch, count_only=True) if seed % 3 else None
            res = eng.call_filters(sb.batch, planes, filters, dp_plane=0 if use_dp else -1,
                                   out=eng.alloc_call_out(sb.batch, nf, place=False), delta_stats=st)
            outs.append((res.gt_out.get(), res.filter_mask.get(), res.sample_counters.get(), res.sample_totaldp.get(),
                         res.sample_dp_missing.get(), res.error.get()[0] != 0,
                         None if st is None else (st.allele_count.get(), st.locus_int.get()[..., :6])))
        finally:
            L.set_option('TRK_CF_GENERIC', None)
    a, b = outs
    assert a[5] == b[5]
    if a[5]:
        return                      # a negative depth on a passing call: both kernels report it, outputs are void
    for x, y, what in zip(a[:5], b[:5], ('gt_out', 'mask', 'counters', 'totaldp', 'dp_missing')):
        assert np.array_equal(x, y), (what, filters)
    if a[6] is not None:
        assert np.array_equal(a[6][0], b[6][0]) and np.array_equal(a[6][1], b[6][1]), filters
