"""Property-based GPU parity (hypothesis): random small batches -- any sample count (rows aligned or not), ploidy 1-3,
1-9 alleles with duplicate length / sequence classes, arbitrary mixes of missing (-1) and padding (-2) haplotypes,
optional sample groups -- against the numpy oracle for the statistics, and random threshold filter sets against the
oracle's call-filter restatement for masks, masked genotypes and per-sample counters."""
import collections

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from test_gpu_stats import check_against_oracle

pytestmark = pytest.mark.gpu

# TRK_PROPERTY_SCALE=k: one-off campaigns on the GPU box -- k times the examples, fresh random seeds (the suite itself
# runs the derandomised set)
import os
_SCALE = int(os.environ.get('TRK_PROPERTY_SCALE', '0'))


def _cfg(n):
    return settings(max_examples=n * max(_SCALE, 1), deadline=None, derandomize=_SCALE == 0, database=None,
                    suppress_health_check=list(HealthCheck))


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _batch(rng, n_loci, S, P, with_low):
    from trtools_amd.synth import pack_alleles
    lens, strs, gts, lp = [], [], [], []
    for l in range(n_loci):
        motif = ''.join(rng.choice(list('ACGT'), size=int(rng.integers(1, 4))))
        A = 1 + int(rng.integers(0, 9))
        ss = []
        while len(ss) < A:
            s = motif * int(rng.integers(1, 12))
            if rng.random() < 0.3:
                s += 'N' * int(rng.integers(1, 3))          # same motif count, different length class
            if s in ss and rng.random() < 0.7:
                continue
            ss.append(s)                                   # now and then a duplicate sequence
        strs.append(ss)
        lens.append([len(s) / len(motif) for s in ss])
        pl = P if not with_low else int(rng.integers(1, P + 1))
        lp.append(pl)
        g = rng.integers(0, A, size=(S, P)).astype(np.int16)
        g[rng.random(S) < rng.random() * 0.4] = -1
        g[rng.random(S) < 0.05, int(rng.integers(0, P))] = -1
        if pl < P:
            g[:, pl:] = -2
        elif P > 1 and rng.random() < 0.5:
            g[rng.random(S) < 0.15, P - 1] = -2
        if rng.random() < 0.1:
            g[:] = -1
        gts.append(g)
    return np.stack(gts), lens, strs, np.array(lp, dtype=np.uint8), pack_alleles(lens, strs)


@_cfg(200)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(1, 10),
       S=st.one_of(st.integers(1, 260), st.sampled_from([1020, 1024, 1028, 2049, 4100])), P=st.integers(1, 3),
       with_low=st.booleans(), n_groups=st.integers(0, 3))
def test_statistics_match_the_oracle(eng, seed, n_loci, S, P, with_low, n_groups):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(seed)
    gt, lens, strs, lp, (off, lc, sc, cv) = _batch(rng, n_loci, S, P, with_low)
    gb, groups = None, [None]
    if n_groups:
        gb = rng.integers(0, 1 << n_groups, size=S).astype(np.uint8)
        groups = [((gb >> g) & 1).astype(bool) for g in range(n_groups)]
    b = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp if with_low else None, group_bits=gb,
                       n_groups=max(n_groups, 1))
    res = eng.locus_stats(b, nalleles_thresh=0.05)
    gt_view = [gt[l][:, :lp[l]] for l in range(n_loci)]
    check_against_oracle(orc, L, res.allele_count.get(), res.locus_int.get(), res.locus_f64.get(), off, gt_view, lens,
                         strs, groups, 0.05)
    for a in list(b.arrays.values()) + [res.allele_count, res.locus_int, res.locus_f64]:
        a.free()


@_cfg(200)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(1, 8),
       S=st.one_of(st.integers(1, 300), st.sampled_from([1020, 1024, 1028, 2052, 4100])), n_filters=st.integers(0, 7),
       delta=st.booleans(), with_low=st.booleans())
def test_threshold_call_filters_match_the_oracle(eng, seed, n_loci, S, n_filters, delta, with_low):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(seed)
    gt, lens, strs, lp, (off, lc, sc, cv) = _batch(rng, n_loci, S, 2, with_low)
    dp = rng.integers(0, 40, size=(n_loci, S)).astype(np.int32)
    dp[rng.random((n_loci, S)) < 0.05] = -2147483648
    nocall = np.stack([np.any(gt[l][:, :lp[l]] == -1, axis=1) for l in range(n_loci)])
    dp[nocall & (rng.random((n_loci, S)) < 0.5)] = -2147483648
    q = np.round(rng.random((n_loci, S)), 2).astype(np.float32)
    q[rng.random((n_loci, S)) < 0.05] = np.nan
    num = rng.integers(0, 12, size=(n_loci, S)).astype(np.int32)       # DSTUTTER / DFLANKINDEL-like counts
    num[rng.random((n_loci, S)) < 0.05] = -2147483648
    filters, fns = [], []
    for k in range(n_filters):
        plane = int(rng.integers(0, 3))
        gt_op = bool(rng.integers(0, 2))
        if plane == 2:      # HipSTR's ratio over the depth (filters.py:415-484)
            thr = float(rng.choice([0.15, 0.1, 0.25, 1.0 / 3.0, 0.0]))
            filters.append(dict(op=L.F_RATIO_GT, plane_a=2, plane_b=0, thr=thr))
        else:
            thr = float(rng.integers(0, 40)) if plane == 0 else float(np.round(rng.random(), 2))
            filters.append(dict(op=L.F_GT if gt_op else L.F_LT, plane_a=plane, thr=thr))
        fns.append((plane, gt_op, thr))
    b = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp if with_low else None)
    st_ = eng.locus_stats(b, count_only=True) if delta else None
    res = eng.call_filters(b, [eng.upload(dp), eng.upload(q), eng.upload(num)], filters, dp_plane=0, delta_stats=st_)
    names = ['f%d' % k for k in range(n_filters)]
    info = collections.OrderedDict([('numcalls', np.zeros(S, dtype=int)), ('totaldp', np.zeros(S))] +
                                   [(n, np.zeros(S, dtype=int)) for n in names])
    gout = np.empty_like(gt)
    mask = np.zeros((n_loci, S), dtype=np.uint32)
    for l in range(n_loci):
        outs = []
        for k, (plane, gt_op, thr) in enumerate(fns):
            if plane == 2:
                with np.errstate(divide='ignore', invalid='ignore'):
                    outs.append((names[k], orc.filt_ratio_gt(num[l].reshape(-1, 1), dp[l].reshape(-1, 1), thr)))
                continue
            field = (dp if plane == 0 else q)[l].reshape(-1, 1)
            outs.append((names[k], orc.filt_max_value(field, thr) if gt_op else orc.filt_min_value(field, thr)))
        for k, (_, o) in enumerate(outs):
            mask[l] |= (~np.isnan(o)).astype(np.uint32) << np.uint32(k)
        g = gt[l][:, :lp[l]]                # the reference sees the record's own ploidy columns
        mask[l] |= (~orc.get_called_samples(g)).astype(np.uint32) << np.uint32(31)
        try:
            gout[l] = gt[l]
            gout[l][:, :lp[l]], _ = orc.apply_call_filters(g, outs, info, dp=dp[l].reshape(-1, 1))
        except ValueError:      # a passing call with negative depth: the device reports it in `error`
            assert res.error.get()[0] != 0
            return
    assert res.error.get()[0] == 0
    assert np.array_equal(res.filter_mask.get(), mask)
    assert np.array_equal(res.gt_out.get(), gout)
    cnt = res.sample_counters.get()
    assert np.array_equal(cnt[0], info['numcalls'])
    for k, n in enumerate(names):
        assert np.array_equal(cnt[1 + k], info[n])
    tot = res.sample_totaldp.get().astype(float)
    tot[res.sample_dp_missing.get() > 0] = np.nan
    assert np.array_equal(tot, info['totaldp'], equal_nan=True)
    if delta:
        recount = eng.locus_stats(b.with_gt(res.gt_out), count_only=True)
        assert np.array_equal(st_.allele_count.get(), recount.allele_count.get())
        cols = [L.LI_N_CALLED, L.LI_N_LOWPLOIDY, L.LI_N_HOM_LEN, L.LI_N_HOM_STR]
        assert np.array_equal(st_.locus_int.get()[0][:, cols], recount.locus_int.get()[0][:, cols])


@_cfg(120)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(6, 14), S=st.integers(40, 700), P=st.integers(1, 3),
       M=st.integers(1, 9), subset=st.booleans(), locus_ploidy=st.booleans(), miss=st.sampled_from([0.0, 0.04, 0.3]))
def test_association_scan_matches_the_oracle(eng, seed, n_loci, S, P, M, subset, locus_ploidy, miss):
    """trk_assoc_scan on random shapes (aligned and unaligned rows, ploidy 1-3 with and without a per-locus ploidy
    table, 1-9 trait columns -> one-trait / LDS-resident / MFMA / per-call kernels, sample subsets, missing rates)
    against the associaTR oracle, with the tolerances of tests/test_gpu_assoc.py."""
    from test_gpu_assoc import run_case
    run_case(eng, seed, n_loci, S, P=P, M=M, subset=subset, locus_ploidy=locus_ploidy and P > 1, miss=miss)


@_cfg(60)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(3, 40), S=st.integers(64, 180).map(lambda q: 4 * q),
       M=st.sampled_from([5, 12, 15, 16, 17, 24, 31, 32, 33, 40, 47, 48, 55, 62]), subset=st.booleans(),
       miss=st.sampled_from([0.0, 0.04, 0.3, 0.9]))
def test_association_scan_many_columns_matches_the_oracle(eng, seed, n_loci, S, M, subset, miss):
    """The matrix-pipe path of trk_assoc_scan (5-62 trait columns = one to four 16-row tiles in one pass, the
    missing-call Gram correction from the scan's bit per sample, the wave-per-locus solve from 16 columns) on random
    shapes: rows of 256-720 samples (whole and partial 256-sample steps), 3-40 loci (partial 16-locus tiles),
    missing rates up to 0.9 (sample lists longer than one consume round), sample subsets.
    Domain: at least ~6 tested samples per design column.  The device solves the normal equations (Cholesky on
    X'X, float64); with 58 tested samples for 35 columns (seed 365, 720 samples, 0.9 missing, M = 33) it keeps 8
    digits of `se` where statsmodels' pinv keeps 12 -- the one-pass and the pair-of-row-groups paths agree to the last
    bit there, it is the conditioning of X'X, not a kernel."""
    from hypothesis import assume
    assume(S * (1.0 - miss) * (0.75 if subset else 1.0) >= 6 * (M + 2))
    from test_gpu_assoc import run_case
    run_case(eng, seed, n_loci, S, P=2, M=M, subset=subset, locus_ploidy=False, miss=miss)


@_cfg(100)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(1, 9), S=st.integers(1, 520),
       layout=st.sampled_from(['interleaved', 'planar', 'planarize']),
       keep=st.sets(st.integers(0, 8), min_size=1), delta=st.booleans(), with_low=st.booleans(),
       thr=st.tuples(st.integers(0, 20), st.integers(10, 45), st.sampled_from([0.0, 0.1, 0.35]),
                     st.sampled_from([0.05, 0.2, 0.5]), st.sampled_from([0.2, 0.6, 1.0]), st.integers(0, 6)))
def test_gangstr_and_popstr_filters_match_the_oracle(eng, seed, n_loci, S, layout, keep, delta, thr, with_low):
    """Random subsets of the GangSTR / PopSTR call filters, random thresholds, any sample count, all three plane
    layouts, with and without the delta outputs (register interpreter, per-call path and their mixtures)."""
    from test_gpu_callfilters import run_gangstr_popstr_case
    run_gangstr_popstr_case(eng, seed, n_loci, S, layout, keep=keep, thr=thr, delta=delta, with_low=with_low)


@pytest.fixture(scope='module')
def comp(eng):
    from trtools_amd.compute import DeviceCompute
    return DeviceCompute(engine=eng)


@_cfg(150)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(1, 9), S=st.integers(1, 130), P=st.integers(1, 3),
       with_low=st.booleans(), n_groups=st.integers(0, 2), n_filters=st.integers(0, 5))
def test_compute_seam_matches_the_oracle_seam(comp, seed, n_loci, S, P, with_low, n_groups, n_filters):
    """What the CLIs call (compute.DeviceCompute: row padding, per-locus ploidy tables, planar planes, delta
    outputs, finaliser, locus filters) against the oracle-backed object the CPU tests substitute for it
    (tests/oracle_compute.py), on random host batches."""
    from oracle_compute import OracleCompute
    from trtools_amd import _lib as L
    from trtools_amd.batch import HostBatch
    rng = np.random.default_rng(seed)
    gt, lens, strs, lp, _ = _batch(rng, n_loci, S, P, with_low)
    gb = None
    if n_groups:
        gb = rng.integers(0, 1 << n_groups, size=S).astype(np.uint8)
    hb = HostBatch(gt, lp, lens, strs, group_bits=gb, n_groups=max(n_groups, 1))
    orc_c = OracleCompute()
    a, b = comp.locus_stats(hb, nalleles_thresh=0.05), orc_c.locus_stats(hb, nalleles_thresh=0.05)
    # (the integer columns the host layer reads; the oracle-backed object fills no others)
    cols = [L.LI_N_CALLED, L.LI_N_SAMPLES, L.LI_N_ALLELES, L.LI_NALLELES_LEN, L.LI_NALLELES_STR, L.LI_HWE_STATUS_LEN,
            L.LI_HWE_STATUS_STR]
    assert np.array_equal(a.allele_count, b.allele_count)
    assert np.array_equal(a.locus_int[:, :, cols], b.locus_int[:, :, cols])
    nf = L.LF_CALLRATE + 1          # the named float columns (the last one of the 12 is spare)
    assert np.allclose(a.locus_f64[..., :nf], b.locus_f64[..., :nf], rtol=1e-9, atol=1e-12, equal_nan=True)
    if n_groups:
        return                      # the dumpSTR pass is defined for one sample group
    dp = rng.integers(0, 40, size=(n_loci, S)).astype(np.int32)
    dp[rng.random((n_loci, S)) < 0.05] = -2147483648
    q = np.round(rng.random((n_loci, S)), 2).astype(np.float32)
    qexp = rng.random((n_loci, S, 3)).astype(np.float32)
    filters = []
    for k in range(n_filters):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            filters.append(dict(op=L.F_LT, plane_a=0, thr=float(rng.integers(0, 30))))
        elif kind == 1:
            filters.append(dict(op=L.F_GT, plane_a=0, thr=float(rng.integers(10, 40))))
        elif kind == 2:
            filters.append(dict(op=L.F_LT, plane_a=1, thr=float(np.round(rng.random(), 2))))
        else:
            filters.append(dict(op=L.F_CALLED_SUM_LT, plane_a=2, col_a=1, col_a2=2, thr=float(np.round(rng.random() * 1.5, 2))))
    spec = dict(min_callrate=float(rng.choice([0.0, 0.5, 0.9])), min_het=0.05, max_het=0.9)
    # (the HWE locus filter needs every tested locus to have a fully called genotype; both seams agree on the
    # error count, compared below through the counters)
    if rng.random() < 0.5:
        spec['min_hwep'] = 0.01
    (ca, sa, ba, la) = comp.dumpstr_batch(hb, [dp, q, qexp], filters, 0, spec, nalleles_thresh=0.05)
    (cb, sb, bb, lb) = orc_c.dumpstr_batch(hb, [dp, q, qexp], filters, 0, spec, nalleles_thresh=0.05)
    if ca.error[0]:
        return                      # a passing call with negative depth: the reference raises here
    assert np.array_equal(ca.mask, cb.mask)
    assert np.array_equal(ca.gt_out, cb.gt_out)
    assert np.array_equal(ca.sample_counters, cb.sample_counters)
    tot_a = ca.totaldp.astype(float)
    tot_a[ca.dp_missing > 0] = np.nan
    tot_b = np.asarray(cb.totaldp, dtype=float)
    tot_b[cb.dp_missing > 0] = np.nan
    assert np.array_equal(tot_a, tot_b, equal_nan=True)
    assert np.array_equal(sa.allele_count, sb.allele_count)
    assert np.array_equal(sa.locus_int[:, :, cols], sb.locus_int[:, :, cols])
    assert np.allclose(sa.locus_f64[..., :nf], sb.locus_f64[..., :nf], rtol=1e-9, atol=1e-12, equal_nan=True)
    assert np.array_equal(ba, bb)
    assert np.array_equal(la, lb)


@_cfg(60)
@given(seed=st.integers(0, 10**6), n_loci=st.integers(6, 16), S=st.integers(60, 420), M=st.integers(1, 8),
       amax=st.integers(1, 12))
def test_dosage_scan_matches_the_oracle(eng, seed, n_loci, S, M, amax):
    """trk_assoc_scan_dosage (--beagle-dosages) on random shapes: allele sets of 1..12 (both the single-pass and the
    two-pass kernel), 1-8 trait columns, sample subsets, missing calls."""
    from test_gpu_assoc import run_dosage_case
    run_dosage_case(eng, seed, n_loci, S, M, amax)


@_cfg(100)
@given(seed=st.integers(0, 2**31 - 1), n_loci=st.integers(1, 8), S=st.integers(1, 200),
       dtype_=st.sampled_from(['bestguess', 'bestguess_norm', 'beagleap', 'beagleap_norm']), amax=st.integers(1, 12))
def test_dosages_match_the_oracle(comp, seed, n_loci, S, dtype_, amax):
    """TRRecord.GetDosages for a batch (trk_dosages) against the oracle: four dosage types, 1-12 alleles (numpy's
    blocked float32 row sums above 8 terms), missing calls, AP rows that do not sum to one, the error conditions."""
    from oracle_compute import OracleCompute
    from test_dosages import close32
    from trtools_amd.batch import HostBatch
    rng = np.random.default_rng(seed)
    lens, strs, gts = [], [], []
    for l in range(n_loci):
        A = int(rng.integers(1, amax + 1))
        strs.append(['AC' * (i + 1) for i in range(A)])
        lens.append([float(i + 1) + (0.5 if rng.random() < 0.2 else 0.0) for i in range(A)])
        g = rng.integers(0, A, size=(S, 2)).astype(np.int16)
        g[rng.random(S) < 0.1] = -1
        if rng.random() < 0.1:
            g[:] = -1
        gts.append(g)
    hb = HostBatch(np.stack(gts), np.full(n_loci, 2, dtype=np.uint8), lens, strs)
    ap1 = ap2 = None
    if dtype_.startswith('beagle'):
        K = max(len(x) for x in lens) - 1
        aps = []
        for _ in range(2):
            a = np.zeros((n_loci, S, max(K, 1)), dtype=np.float32)
            for l in range(n_loci):
                k = len(lens[l]) - 1
                if k:
                    p = rng.dirichlet(np.full(k + 1, 0.5), size=S)
                    a[l, :, :k] = (p[:, 1:] * rng.uniform(0.85, 1.15, size=(S, 1))).astype(np.float32)
                if rng.random() < 0.05:
                    a[l, int(rng.integers(0, S)), 0] = -0.25      # negative probability -> error bit
                if rng.random() < 0.05 and k:
                    a[l, int(rng.integers(0, S)), :k] = 1.0       # row sum above 1.1 -> error bit
            aps.append(a)
        ap1, ap2 = aps
    got, gerr = comp.dosages_batch(hb, dtype_, ap1, ap2)
    want, werr = OracleCompute().dosages_batch(hb, dtype_, ap1, ap2)
    assert np.array_equal(gerr, werr)
    ok = werr == 0
    assert close32(got[ok], want[ok])
