"""GPU parity at the BASELINE.json configuration sizes.

configs[1]  statSTR full stats, synthetic HipSTR-shape, 10k loci x 1k samples:
            EVERY locus against the C half of the oracle + the fast kernel against the
            general kernel (two independent device code paths) + sampled loci against the
            numpy oracle.
configs[2]  dumpSTR call + locus filters, synthetic GangSTR-shape, 50k loci x 5k samples:
            size-independent invariants over the whole batch + sampled loci against the
            numpy oracle (rows regenerated on the host by the generator's numpy twin).
configs[3]  (100k x 10k) is what bench.py runs, with the same spot checks.
configs[4]  associaTR linear-regression scan, 100k loci x 10k samples x 1 trait (one GPU's share and more):
            size-independent properties over the whole batch (invariance of p / R^2 / counts under an affine
            change of the trait with the coefficient scaling along, sample masking == dropping the samples'
            contribution, the one-trait kernel == the MFMA kernel == the per-call kernel on the same input,
            sum of allele counts == 2 x tested samples - padding) + sampled loci against the associaTR oracle.
"""
import collections
import math

import numpy as np
import pytest

from helpers import close, lab_env

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_config1_statstr_10k_x_1k(eng):
    from oracle import oracle_c, trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    Lc, S = 10000, 1000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 1, planes=())
    res = eng.locus_stats(sb.batch, nalleles_thresh=0.01)
    cnt, li, lf = res.allele_count.get()[0], res.locus_int.get()[0], res.locus_f64.get()[0]
    off, lc, sc, cv = sb.tables
    gt = sb.dev['gt'].get()
    # (1) every locus vs the C oracle
    ccnt, oi, of = oracle_c.batch_stats(gt, None, off, lc, sc, cv)
    assert np.array_equal(cnt, ccnt)
    assert np.array_equal(li[:, L.LI_N_CALLED], oi[:, 0]) and np.array_equal(li[:, L.LI_N_LOWPLOIDY], oi[:, 1])
    assert np.array_equal(li[:, L.LI_N_HOM_LEN], oi[:, 2]) and np.array_equal(li[:, L.LI_N_HOM_STR], oi[:, 3])
    assert np.array_equal(li[:, L.LI_N_ALLELES], oi[:, 7]) and not li[:, L.LI_N_BAD].any()
    assert np.array_equal(li[:, L.LI_HWE_STATUS_LEN], oi[:, 5]) and np.array_equal(li[:, L.LI_HWE_STATUS_STR], oi[:, 6])
    cols = [L.LF_THRESH, L.LF_MEAN, L.LF_MODE, L.LF_VAR, L.LF_HET_LEN, L.LF_HET_STR, L.LF_ENTROPY_LEN,
            L.LF_ENTROPY_STR, L.LF_HWEP_LEN, L.LF_HWEP_STR]
    for j, c in enumerate(cols):
        a, b = lf[:, c], of[:, j]
        assert np.array_equal(np.isnan(a), np.isnan(b)), c
        ok = ~np.isnan(a)
        assert np.all(np.abs(a[ok] - b[ok]) <= 1e-9 * np.maximum(1.0, np.abs(b[ok])) + 1e-300), c
    # (2) fast kernel vs general kernel (one all-samples group forces the general path)
    b2 = eng.make_batch(sb.dev['gt'], sb.d_off, lc, sc, cv, group_bits=np.ones(S, dtype=np.uint8), n_groups=1,
                        max_alleles=int(np.max(np.diff(off))))
    res2 = eng.locus_stats(b2, nalleles_thresh=0.01)
    assert np.array_equal(res2.allele_count.get()[0], cnt)
    li2 = res2.locus_int.get()[0]
    assert np.array_equal(li2, li)
    lf2 = res2.locus_f64.get()[0]
    assert np.array_equal(np.nan_to_num(lf2, nan=-7.0), np.nan_to_num(lf, nan=-7.0))
    # (3) sampled loci vs the numpy oracle (incl. nalleles, call rate)
    for l in np.random.default_rng(0).choice(Lc, size=60, replace=False):
        o = orc.locus_stats(gt[l], sb.loci.allele_lens[l], sb.loci.allele_strs[l], None, use_length=False)
        assert li[l, L.LI_NALLELES_STR] == o['nalleles']
        assert close(lf[l, L.LF_CALLRATE], orc.get_call_rate(gt[l]))
        assert close(lf[l, L.LF_HET_STR], o['het']) and close(lf[l, L.LF_ENTROPY_STR], o['entropy'])
        if o['hwep_status'] == orc.HWE_OK and not math.isnan(o['hwep']):
            assert close(lf[l, L.LF_HWEP_STR], o['hwep'], 1e-9, 1e-300)


def test_config2_dumpstr_gangstr_50k_x_5k(eng):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    Lc, S = 50000, 5000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 2, planes=('dp', 'q'), pure_repeats=True)
    sb.add_gangstr_planes()
    planes = [sb.dev['dp'], sb.dev['q'], sb.dev['qexp'], sb.dev['rc'], sb.dev['repcn'], sb.dev['repci']]
    # BuildCallFilters order for GangSTR (dumpSTR.py:819-836)
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=60),
               dict(op=L.F_LT, plane_a=1, thr=0.9),
               dict(op=L.F_CALLED_LT, plane_a=2, col_a=1, thr=0.05),
               dict(op=L.F_CALLED_LT, plane_a=2, col_a=2, thr=0.05),
               dict(op=L.F_CALLED_SUM_LT, plane_a=2, col_a=1, col_a2=2, thr=0.2),
               dict(op=L.F_CALLED_EQ, plane_a=3, col_a=1, plane_b=0, col_b=0),
               dict(op=L.F_CALLED_SUM_EQ, plane_a=3, col_a=1, col_a2=3, plane_b=0, col_b=0),
               dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=4, plane_b=5)]
    names = ['mindp', 'maxdp', 'minq', 'het', 'hom', 'total', 'span', 'spanbound', 'badci']
    res = eng.call_filters(sb.batch, planes, filters, dp_plane=0)
    b2 = sb.batch.with_gt(res.gt_out)
    st = eng.locus_stats(b2)
    bits, counters = eng.locus_filters(Lc, st, min_callrate=0.8, min_hwep=1e-3, min_het=0.05, max_het=0.9)
    assert res.error.get()[0] == 0
    cnts = res.sample_counters.get()
    li = st.locus_int.get()[0]
    lf = st.locus_f64.get()[0]
    lc = counters.get()
    bits_h = bits.get()
    # ---- invariants over the whole batch ----
    mask = res.filter_mask.get()
    nocall = (mask >> np.uint32(31)).astype(bool)
    assert int(cnts[0].sum()) == int((mask == 0).sum()) == int(li[:, L.LI_N_CALLED].sum())
    assert np.array_equal(cnts[0], (mask == 0).sum(axis=0))
    for k in range(len(filters)):
        fired = ((mask >> np.uint32(k)) & np.uint32(1)).astype(bool) & ~nocall
        assert np.array_equal(cnts[1 + k], fired.sum(axis=0)), names[k]
        assert fired.any(), names[k]
    del nocall
    assert lc[L.LC_PASS] == int((bits_h == 0).sum())
    assert lc[L.LC_TOTALCALLS] == int(li[bits_h == 0, L.LI_N_CALLED].sum())
    assert lc[L.LC_NO_CALLS] == int((li[:, L.LI_N_CALLED] == 0).sum())
    for b in range(4):
        assert lc[L.LC_FILTER0 + b] == int(((bits_h >> np.uint32(b)) & 1).sum())
    cnt = st.allele_count.get()[0]
    off = sb.tables[0]
    assert np.array_equal(np.add.reduceat(cnt.astype(np.int64), off[:-1]), li[:, L.LI_N_ALLELES])
    # ---- the same planes planar ([k, L, S], every filter evaluated on register-resident vector sources), with
    # the delta outputs: bit-identical masks / genotypes / counters, and counts(GT') = counts(GT) - delta ----
    pl_planes = [eng.planarize(p) for p in planes]
    assert pl_planes[2].planar and pl_planes[2].shape == (3, Lc, S)
    st0 = eng.locus_stats(sb.batch, count_only=True)
    res_p = eng.call_filters(sb.batch, pl_planes, filters, dp_plane=0, delta_stats=st0)
    assert np.array_equal(res_p.filter_mask.get(), mask)
    assert np.array_equal(res_p.sample_counters.get(), cnts)
    assert np.array_equal(res_p.sample_totaldp.get(), res.sample_totaldp.get())
    assert np.array_equal(res_p.sample_dp_missing.get(), res.sample_dp_missing.get())
    assert np.array_equal(res_p.gt_out.get_rows(0, 64), res.gt_out.get_rows(0, 64))
    assert np.array_equal(res_p.gt_out.get_rows(Lc - 64, Lc), res.gt_out.get_rows(Lc - 64, Lc))
    assert np.array_equal(st0.allele_count.get()[0], cnt)
    for col in (L.LI_N_CALLED, L.LI_N_LOWPLOIDY, L.LI_N_HOM_LEN, L.LI_N_HOM_STR):
        assert np.array_equal(st0.locus_int.get()[0][:, col], li[:, col])
    del pl_planes, res_p, st0
    # ---- EVERY locus against the compiled oracle's generic call-filter interpreter (pinned to the numpy oracle by
    # tests/test_fullsize_checker.py): masks / masked genotypes bit for bit on the first 10k+ loci, statistics,
    # locus filter decisions, sample_info and loc_info over all 50k ----
    from oracle import fullsize
    dev = dict(cnt_a=None, cnt_b=cnt, li_b=li, lf_b=lf, bits=bits_h, sample_counters=cnts,
               totaldp=res.sample_totaldp.get(), dpmiss=res.sample_dp_missing.get(), loc_counters=lc)
    r = fullsize.check_step(lambda lo, hi: (sb.dev['gt'].get_rows(lo, hi), [p.get_rows(lo, hi) for p in planes]),
                            lambda lo, hi: (res.gt_out.get_rows(lo, hi), res.filter_mask.get_rows(lo, hi)),
                            Lc, S, sb.tables, filters, 0,
                            dict(min_callrate=0.8, min_hwep=1e-3, min_het=0.05, max_het=0.9, use_length=False), dev,
                            block=2048)
    assert r['loci'] == Lc and r['calls_bit_for_bit'] >= 10000 * S and r['worst_float_rel'] <= 1e-9
    # ---- sampled loci against the numpy oracle ----
    idx = np.sort(np.random.default_rng(1).choice(Lc, size=24, replace=False))
    h = sb.host_rows(idx)
    g = sb.host_gangstr_rows(idx, h)
    for r, l in enumerate(idx):
        l = int(l)
        gt = h['gt'][r]
        d = h['dp'][r].reshape(-1, 1)
        assert np.array_equal(sb.dev['gt'].get_rows(l, l + 1)[0], gt)
        for key in ('qexp', 'rc', 'repcn', 'repci'):
            dev_row = sb.dev[key].get_rows(l, l + 1)[0]
            if key == 'qexp':
                assert np.array_equal(dev_row.view(np.uint32), g[key][r].view(np.uint32)), key
            else:
                assert np.array_equal(dev_row, g[key][r]), key
        rcs = np.array([','.join(map(str, x)) for x in g['rc'][r]])
        cis = np.array(['%d-%d,%d-%d' % tuple(x) for x in g['repci'][r]])
        outs = [('mindp', orc.filt_min_value(d, 10)), ('maxdp', orc.filt_max_value(d, 60)),
                ('minq', orc.filt_min_value(h['q'][r].reshape(-1, 1), 0.9)),
                ('het', orc.filt_gangstr_qexp(gt, g['qexp'][r], 0.05, 'het')),
                ('hom', orc.filt_gangstr_qexp(gt, g['qexp'][r], 0.05, 'hom')),
                ('total', orc.filt_gangstr_qexp(gt, g['qexp'][r], 0.2, 'total')),
                ('span', orc.filt_gangstr_span_only(gt, rcs, d)),
                ('spanbound', orc.filt_gangstr_spanbound_only(gt, rcs, d)),
                ('badci', orc.filt_gangstr_bad_ci(gt, g['repcn'][r], cis))]
        want_mask = np.zeros(S, dtype=np.uint32)
        for k, (_, o) in enumerate(outs):
            want_mask |= (~np.isnan(o)).astype(np.uint32) << np.uint32(k)
        want_mask |= (~orc.get_called_samples(gt)).astype(np.uint32) << np.uint32(31)
        assert np.array_equal(mask[l], want_mask), l
        info = collections.OrderedDict([('numcalls', np.zeros(S, dtype=int)), ('totaldp', np.zeros(S))] +
                                       [(n, np.zeros(S, dtype=int)) for n in names])
        g2, _ = orc.apply_call_filters(gt, outs, info, dp=d)
        assert np.array_equal(res.gt_out.get_rows(l, l + 1)[0], g2), l
        lens, strs = sb.loci.allele_lens[l], sb.loci.allele_strs[l]
        loc = collections.defaultdict(int)
        try:
            _, nm = orc.apply_locus_filters(g2, lens, strs, loc, use_length=False, min_callrate=0.8,
                                            min_hwep=1e-3, min_het=0.05, max_het=0.9)
        except ValueError:
            continue
        bitof = {'CALLRATE0.8': 0, 'HWE0.001': 1, 'HETLOW0.05': 2, 'HETHIGH0.9': 3, 'NO_CALLS_REMAINING': 31}
        want = 0
        for x in nm:
            want |= 1 << bitof[x]
        assert int(bits_h[l]) == want, (l, nm)
        o = orc.locus_stats(g2, lens, strs, None, use_length=False)
        assert np.array_equal(cnt[off[l]:off[l + 1]], o['index_counts'])
        assert close(lf[l, L.LF_HET_STR], o['het'])


def test_config4_associatr_100k_x_10k(eng):
    import os
    from oracle import associatr_oracle as ao
    from trtools_amd.synth import SynthBatch, pack_assoc_tables
    from trtools_amd import _lib as TL
    Lc, S = 100000, 10000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 4, planes=())
    alen, rcls = pack_assoc_tables(sb.loci.allele_lens, 2)
    alen_d, rcls_d = eng.upload(alen, np.float64), eng.upload(rcls, np.uint16)
    rng = np.random.default_rng(44)
    y = rng.normal(size=S)
    y = (y - y.mean()) / y.std()

    def scan(vec, sample_in=None, env=None):
        with lab_env(**(env or {})):
            r = eng.assoc_scan(sb.batch, np.ascontiguousarray(vec), alen_d, rcls_d, sample_in=sample_in,
                               non_major_cutoff=20.0)
            out = (r.locus_int.get(), r.locus_f64.get(), r.allele_count.get())
            for d in (r.locus_int, r.locus_f64, r.allele_count):
                d.free()
            return out

    li, lf, cnt = scan(y[None, :])
    ok = li[:, TL.AI_STATUS] == TL.AS_OK
    assert ok.sum() > 0.8 * Lc
    # counts: every tested sample contributes its two haplotypes (the generator has no ploidy padding)
    off = sb.tables[0].astype(np.int64)
    per_locus = np.add.reduceat(cnt, off[:-1])
    assert np.array_equal(per_locus, 2 * li[:, TL.AI_N_TESTED].astype(np.int64))
    assert np.array_equal(per_locus, li[:, TL.AI_N_HAPS])
    p = lf[ok, TL.AF_PVAL]
    assert np.all((p >= 0) & (p <= 1)) and np.all(lf[ok, TL.AF_SE] > 0)
    assert np.all(lf[ok, TL.AF_DF_RESID] == li[ok, TL.AI_N_TESTED] - 2)
    t2 = lf[ok, TL.AF_TVALUE] ** 2
    df = lf[ok, TL.AF_DF_RESID]
    assert np.allclose(lf[ok, TL.AF_RSQUARED], t2 / (t2 + df), rtol=1e-9, atol=1e-13)   # one regressor + intercept
    # affine change of the trait: same statistics, coefficient and standard error scale
    li2, lf2, _ = scan((3.5 * y - 2.0)[None, :])
    assert np.array_equal(li2, li)
    for col, factor in ((TL.AF_PVAL, 1.0), (TL.AF_RSQUARED, 1.0), (TL.AF_COEF, 3.5), (TL.AF_SE, 3.5), (TL.AF_GT_STD, 1.0)):
        assert np.allclose(lf2[ok, col], factor * lf[ok, col], rtol=1e-8, atol=1e-12 if col == TL.AF_RSQUARED else 1e-300), col
    # three device code paths on the same input: one-trait kernel (above), MFMA kernel, per-call kernel
    for env in ({'TRK_AS_MFMA_MIN': '1'}, {'TRK_AS_GENERIC': '1'}):
        li3, lf3, cnt3 = scan(y[None, :], env=env)
        assert np.array_equal(li3, li) and np.array_equal(cnt3, cnt), env
        for col in (TL.AF_PVAL, TL.AF_COEF, TL.AF_SE, TL.AF_RSQUARED, TL.AF_GT_STD):
            assert np.allclose(lf3[ok, col], lf[ok, col], rtol=1e-9, atol=1e-13 if col == TL.AF_RSQUARED else 1e-300), (env, col)
    # a sample mask: masked samples count for nothing (compare with zeroing nothing else)
    mask = (rng.random(S) < 0.9).astype(np.uint8)
    ym = y.copy()
    m = mask.astype(bool)
    ym[m] = (y[m] - y[m].mean()) / y[m].std()
    lim, lfm, cntm = scan(ym[None, :], sample_in=mask)
    assert np.all(lim[:, TL.AI_N_TESTED] <= li[:, TL.AI_N_TESTED]) and lim[:, TL.AI_N_TESTED].max() <= m.sum()
    # EVERY locus of both runs against the compiled restatement of the scan (oracle_c.c orc_assoc_locus, pinned to the
    # numpy oracle by tests/test_oracle_c.py): tested samples, filter decisions, p / coefficient / se / R^2
    from oracle import fullsize
    gt_d = sb.dev['gt']
    x1 = np.zeros((S, 2))
    x1[:, 1] = 1.0
    r_all = fullsize.check_assoc(lambda lo, hi: gt_d.get_rows(lo, hi), Lc, S, sb.tables[0], alen, x1, y, li, lf)
    assert r_all['loci'] == Lc and r_all['regressed'] > 0.8 * Lc and r_all['worst_rel'] <= 1e-9
    ym_full = np.zeros(S)
    ym_full[m] = ym[m]
    r_m = fullsize.check_assoc(lambda lo, hi: gt_d.get_rows(lo, hi), Lc, S, sb.tables[0], alen, x1, ym_full, lim, lfm,
                               sample_in=mask)
    assert r_m['loci'] == Lc and r_m['worst_rel'] <= 1e-9
    # sampled loci against the numpy oracle (rows regenerated by the generator's numpy twin), full and masked runs
    idx = np.unique(np.linspace(0, Lc - 1, 9).astype(int))
    rows = sb.host_rows(idx)
    for k, l in enumerate(idx):
        for (I, F, sf, yy) in ((li, lf, np.ones(S, dtype=bool), y), (lim, lfm, m, ym[m])):
            cov = np.ones((int(sf.sum()), 2))
            r = ao.scan_locus(rows['gt'][k], sb.loci.allele_lens[l], sf, cov, yy, 1.0, 20.0, 2)
            assert I[l, TL.AI_N_TESTED] == r['n_tested'], l
            assert (I[l, TL.AI_STATUS] == TL.AS_OK) == (not r['locus_filtered']), (l, r['locus_filtered'])
            if not r['locus_filtered']:
                for col, key in ((TL.AF_PVAL, 'pval'), (TL.AF_COEF, 'coef_std'), (TL.AF_SE, 'se_std'),
                                 (TL.AF_RSQUARED, 'rsquared')):
                    assert abs(F[l, col] - r[key]) <= 1e-9 * abs(r[key]) + 1e-13, (l, key)


def test_config3_combined_100k_x_10k_two_queue_pipeline(eng):
    """configs[3] exactly as bench.py runs it (statSTR + dumpSTR on 100k x 10k, finalisers and locus filters on the
    second queue, the dumpSTR tail one step behind): three pipelined steps give the same outputs as one step with
    everything in order on queue 0, and the oracle spot checks hold."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from trtools_amd.synth import make_loci
    seed, S = 20260928 + 3, 10000
    loci = make_loci(100000, S, seed)
    ref = bench.Workload(eng, seed, S, loci, 0, 1, use_comm=False, overlap=False)
    ref.step()
    ref.flush()
    eng.sync()
    want = dict(bits=ref.bits.get(), loc=ref.loc_counters.get(), cnt=ref.call_out.sample_counters.get(),
                li_a=ref.stats_a[0].locus_int.get(), lf_a=ref.stats_a[0].locus_f64.get(),
                li_b=ref.stats_b[0].locus_int.get(), lf_b=ref.stats_b[0].locus_f64.get(),
                mask_head=ref.call_out.filter_mask.get_rows(0, 32))
    # free the big buffers of the reference run before the second workload allocates its own
    for arr in list(eng._live):
        if arr.nbytes > (1 << 28) and arr not in ref.sb.dev.values():
            arr.free()
    wl = bench.Workload(eng, seed, S, loci, 0, 1, use_comm=False)
    assert wl.overlap
    for _ in range(3):
        wl.step()
    wl.flush()
    eng.sync()
    i = wl.last
    assert np.array_equal(wl.bits.get(), want['bits'])
    assert np.array_equal(wl.loc_counters.get(), want['loc'])
    assert np.array_equal(wl.call_out.sample_counters.get(), want['cnt'])
    assert np.array_equal(wl.stats_a[i].locus_int.get(), want['li_a'])
    assert np.array_equal(wl.stats_a[i].locus_f64.get(), want['lf_a'], equal_nan=True)
    assert np.array_equal(wl.stats_b[i].locus_int.get(), want['li_b'])
    assert np.array_equal(wl.stats_b[i].locus_f64.get(), want['lf_b'], equal_nan=True)
    assert np.array_equal(wl.call_out.filter_mask.get_rows(0, 32), want['mask_head'])
    # EVERY locus of the 100k x 10k call set against the compiled oracle (statistics before and after masking, locus
    # filter decisions, sample_info, loc_info; masks and masked genotypes bit for bit on the first 10 000+ loci)
    r = bench.exhaustive_check(wl, single_rank_sums=True)
    assert r['loci'] == 100000 and r['calls_bit_for_bit'] >= 10000 * S and r['worst_float_rel'] <= 1e-9


def test_config3_locus_shards_add_up_to_the_cohort(eng):
    """BASELINE configs[3] as the multi-GPU run cuts it: contiguous locus shards of ONE cohort (dist.locus_shard).  Two
    shards run one after the other on this GPU give, put together, exactly what the unsharded run gives: per-locus
    rows concatenate in rank order, sample_info / loc_info add up (the sums RCCL forms over xGMI)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from trtools_amd.dist import locus_shard
    from trtools_amd.synth import make_loci
    seed, S, Lc = 20260928 + 3, 2000, 9001
    loci = make_loci(Lc, S, seed)
    eng.comm_init(0, 1, eng.comm_unique_id())      # 1-rank RCCL communicator: the exchange kernels really launch

    def run(lo, hi, world):
        # the shards run as a rank of the multi-GPU job does: the count pass of a step on its own queue, beside the
        # end of the previous step's call filters (pipeline_count)
        wl = bench.Workload(eng, seed, S, loci.slice(lo, hi), lo, 1, use_comm=True, gather_loci=-(-Lc // world),
                            pipeline_count=world > 1)
        for _ in range(4):
            wl.step()
        wl.flush()
        eng.sync()
        i = wl.last
        out = dict(bits=wl.bits.get()[:hi - lo], gathered=wl.gather.get()[0][:hi - lo], loc=wl.loc_counters.get(),
                   cnt=wl.call_out.sample_counters.get(), td=wl.call_out.sample_totaldp.get(),
                   li_a=wl.stats_a[i].locus_int.get()[0], lf_a=wl.stats_a[i].locus_f64.get()[0],
                   li_b=wl.stats_b[i].locus_int.get()[0], lf_b=wl.stats_b[i].locus_f64.get()[0],
                   cnt_b=wl.stats_b[i].allele_count.get()[0])
        r = bench.exhaustive_check(wl, single_rank_sums=True)
        assert r['loci'] == hi - lo
        wl.free()
        return out

    whole = run(0, Lc, 1)
    assert np.array_equal(whole['bits'], whole['gathered'])
    for world in (2, 3):
        parts = [run(*locus_shard(Lc, r, world), world) for r in range(world)]
        for key in ('bits', 'li_a', 'li_b', 'cnt_b'):
            assert np.array_equal(np.concatenate([p[key] for p in parts]), whole[key]), (world, key)
        for key in ('lf_a', 'lf_b'):
            assert np.array_equal(np.concatenate([p[key] for p in parts]), whole[key], equal_nan=True), (world, key)
        for key in ('loc', 'cnt', 'td'):
            assert np.array_equal(sum(p[key] for p in parts), whole[key]), (world, key)


def test_config3_use_length_every_locus(eng):
    """The combined step with --use-length locus filters (length alleles: HWE / heterozygosity by repeat length), a
    40 000 x 4 000 cohort, every locus against the compiled oracle (the full-size checks ran
    use_length = False only)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from trtools_amd.synth import make_loci
    seed, S, Lc = 20260928 + 13, 4000, 40000
    loci = make_loci(Lc, S, seed)
    wl = bench.Workload(eng, seed, S, loci, 0, 1, use_comm=False)
    wl.locus_args = dict(bench.LOCUS_ARGS, use_length=True, min_hwep=1e-6, min_het=0.3)
    for _ in range(2):
        wl.step()
    wl.flush()
    eng.sync()
    r = bench.exhaustive_check(wl, single_rank_sums=True)
    assert r['loci'] == Lc and r['worst_float_rel'] <= 1e-9
    # the two allele notions really decide differently on this cohort (else the test would prove nothing)
    wl2 = bench.Workload(eng, seed, S, loci, 0, 1, use_comm=False)
    wl2.locus_args = dict(wl.locus_args, use_length=False)
    wl2.step()
    wl2.flush()
    eng.sync()
    assert not np.array_equal(wl.bits.get(), wl2.bits.get())
    wl.free()
    wl2.free()


@pytest.mark.parametrize('layout', ['two_disjoint', 'five_overlapping'])
def test_statstr_sample_groups_every_locus(eng, layout):
    """statSTR --samples at 50 000 x 5 000: every locus of every group against the compiled oracle on the group's
    columns -- the one-pass grouped kernel (<= 3 groups), the class-column-range path (trk_batch.class_runs) and,
    for the two-group layout, both against each other bit for bit."""
    from oracle import fullsize
    from trtools_amd.synth import SynthBatch
    Lc, S = 50000, 5000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 21, planes=())
    rng = np.random.default_rng(21)
    if layout == 'two_disjoint':
        a = rng.random(S) < 0.4
        gb, G = (a * 1 + (~a) * 2).astype(np.uint8), 2
    else:
        gb, G = rng.integers(0, 32, size=S).astype(np.uint8), 5
    masks = [((gb >> g) & 1).astype(bool) for g in range(G)]
    gt_d = sb.dev['gt']

    def fetch(lo, hi):
        return gt_d.get_rows(lo, hi), []
    results = []
    for path in (('sorted',) if G > 3 else ('grouped', 'sorted')):
        b = sb.batch.sorted_by_class(eng, gb, G) if path == 'sorted' else sb.batch.with_groups(eng, gb, G)
        res = eng.locus_stats(b, nalleles_thresh=0.01)
        dev = dict(cnt=res.allele_count.get(), li=res.locus_int.get(), lf=res.locus_f64.get())
        r = fullsize.check_group_stats(fetch, Lc, S, sb.tables, masks, dev)
        assert r['loci'] == Lc and r['worst_float_rel'] <= 1e-9
        results.append(dev)
        for x in (res.allele_count, res.locus_int, res.locus_f64):
            x.free()
        if path == 'sorted':
            b.arrays['gt'].free()
    if len(results) == 2:
        for k in ('cnt', 'li', 'lf'):
            assert np.array_equal(results[0][k], results[1][k], equal_nan=True), k


@pytest.mark.parametrize('n_cov,subset', [(2, False), (6, True)])
def test_associatr_with_covariates_every_locus(eng, n_cov, subset):
    """The scan with covariates (the LDS-resident and the MFMA kernels: 3 / 7 trait vectors) on 30 000 x 4 000, with
    and without a sample subset: every locus against the compiled restatement."""
    from oracle import fullsize
    from trtools_amd.synth import SynthBatch, pack_assoc_tables
    Lc, S = 30000, 4000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 31 + n_cov, planes=())
    alen, rcls = pack_assoc_tables(sb.loci.allele_lens, 2)
    rng = np.random.default_rng(31 + n_cov)
    keep = (rng.random(S) < 0.85) if subset else np.ones(S, dtype=bool)
    raw = rng.normal(size=(S, 1 + n_cov))
    raw[:, 1] += 0.3 * raw[:, 0]                          # correlated covariate
    z = (raw[keep] - raw[keep].mean(axis=0)) / raw[keep].std(axis=0)
    vec = np.zeros((1 + n_cov, S))
    vec[:, keep] = z.T
    res = eng.assoc_scan(sb.batch, vec, alen, rcls, sample_in=keep.astype(np.uint8) if subset else None,
                         non_major_cutoff=20.0)
    li, lf = res.locus_int.get(), res.locus_f64.get()
    x = np.zeros((S, 2 + n_cov))
    x[:, 1] = 1.0
    x[:, 2:] = vec[1:].T
    gt_d = sb.dev['gt']
    r = fullsize.check_assoc(lambda lo, hi: gt_d.get_rows(lo, hi), Lc, S, sb.tables[0], alen, x, vec[0], li, lf,
                             sample_in=keep.astype(np.uint8) if subset else None)
    assert r['loci'] == Lc and r['regressed'] > 0.7 * Lc and r['worst_rel'] <= 1e-9
    for d in (res.locus_int, res.locus_f64, res.allele_count):
        d.free()
