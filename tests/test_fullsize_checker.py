"""CPU: pins the pieces the full-size parity runs rely on -- oracle_c's generic call-filter interpreter
(orc_call_filters), its threaded statistics (orc_batch_stats_mt) and oracle/fullsize.check_step's locus-filter /
loc_info bookkeeping -- against the numpy oracle (tests/oracle_compute.OracleCompute, pinned to the reference by
tests/test_oracle_golden.py).  The numpy oracle's results play the part of the device here; in the GPU tests and in
bench.py the same check_step sees the HIP results."""
import numpy as np
import pytest

from oracle import fullsize, oracle_c
from oracle_compute import OracleCompute
from trtools_amd import _lib as L
from trtools_amd.batch import HostBatch
from trtools_amd.synth import make_loci, cells_numpy, gangstr_planes_numpy


def _gangstr_filters():
    return [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=60), dict(op=L.F_LT, plane_a=1, thr=0.9),
            dict(op=L.F_CALLED_LT, plane_a=2, col_a=1, thr=0.05), dict(op=L.F_CALLED_LT, plane_a=2, col_a=2, thr=0.05),
            dict(op=L.F_CALLED_SUM_LT, plane_a=2, col_a=1, col_a2=2, thr=0.2),
            dict(op=L.F_CALLED_EQ, plane_a=3, col_a=1, plane_b=0, col_b=0),
            dict(op=L.F_CALLED_SUM_EQ, plane_a=3, col_a=1, col_a2=3, plane_b=0, col_b=0),
            dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=4, plane_b=5)]


def _complete(hb, gt, st):
    """OracleCompute fills the columns the host layer reads; the homozygote / low-ploidy counts the device also
    reports come from the numpy oracle's genotype counts (utils.py:327-333: sum of counts with gt[0] == gt[1])."""
    from oracle import trtools_oracle as orc
    li = st.locus_int[0]
    for l in range(hb.n_loci):
        g = gt[l]
        for col, reps in ((L.LI_N_HOM_LEN, hb.allele_lens[l]), (L.LI_N_HOM_STR, hb.allele_strs[l])):
            gc = orc.get_genotype_counts(g, list(reps), None)
            li[l, col] = sum(c for k, c in gc.items() if len(k) > 1 and k[0] == k[1])
        li[l, L.LI_N_LOWPLOIDY] = int(np.sum(~np.any(g == -1, axis=1) & np.any(g == -2, axis=1)))
    return st


def _dev_from_oracle(hb, planes, filters, dp_plane, locus_args, with_a=True):
    oc = OracleCompute()
    ch, st, bits, lc = oc.dumpstr_batch(hb, planes, filters, dp_plane, locus_args)
    _complete(hb, ch.gt_out, st)
    dev = dict(cnt_b=st.allele_count[0], li_b=st.locus_int[0], lf_b=st.locus_f64[0], bits=bits,
               sample_counters=ch.sample_counters, totaldp=np.asarray(ch.totaldp, dtype=np.int64), dpmiss=ch.dp_missing,
               loc_counters=lc, cnt_a=None)
    if with_a:
        sa = _complete(hb, hb.gt, oc.locus_stats(hb))
        dev.update(cnt_a=sa.allele_count[0], li_a=sa.locus_int[0], lf_a=sa.locus_f64[0])
    return dev, ch


@pytest.mark.parametrize('threads', [1, 3])
def test_check_step_gangstr_set_against_numpy_oracle(threads):
    Lc, S, seed = 48, 96, 31
    loci = make_loci(Lc, S, seed, pure_repeats=True)
    h = cells_numpy(seed, loci, np.arange(Lc), S)
    e = gangstr_planes_numpy(seed, loci, np.arange(Lc), S, h['gt'], h['dp'])
    planes = [h['dp'], h['q'], e['qexp'], e['rc'], e['repcn'], e['repci']]
    filters = _gangstr_filters()
    locus_args = dict(min_callrate=0.8, min_hwep=1e-3, min_het=0.05, max_het=0.9, use_length=False)
    hb = HostBatch(h['gt'], np.full(Lc, 2), loci.allele_lens, loci.allele_strs)
    dev, ch = _dev_from_oracle(hb, planes, filters, 0, locus_args)
    # dp-missing: the numpy oracle poisons totaldp (nan); OracleCompute reports it as a 0/1 flag, the C side counts
    dev['dpmiss'] = None
    tables = (hb.allele_off, hb.len_class, hb.str_class, hb.len_class_value)
    r = fullsize.check_step(lambda lo, hi: (h['gt'][lo:hi], [p[lo:hi] for p in planes]),
                            lambda lo, hi: (ch.gt_out[lo:hi], ch.mask[lo:hi]), Lc, S, tables, filters, 0, locus_args,
                            {k: v for k, v in dev.items() if k not in ('sample_counters', 'totaldp', 'dpmiss', 'loc_counters')},
                            block=20, n_threads=threads)
    assert r['loci'] == Lc and r['calls_bit_for_bit'] == Lc * S and r['worst_float_rel'] <= 1e-9
    counters, totaldp, dpmiss, loc = r['sums']
    assert np.array_equal(counters, dev['sample_counters'])
    ok = dpmiss == 0
    assert np.array_equal(totaldp[ok], dev['totaldp'][ok]) and np.array_equal(dpmiss > 0, ch.dp_missing > 0)
    assert np.array_equal(loc, dev['loc_counters'])
    # every filter of the set fires somewhere, so each opcode is really exercised
    assert all(counters[1 + k].sum() > 0 for k in range(len(filters)))


def test_check_step_reports_a_planted_difference():
    Lc, S, seed = 24, 40, 7
    loci = make_loci(Lc, S, seed)
    h = cells_numpy(seed, loci, np.arange(Lc), S)
    planes = [h['dp'], h['q']]
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=55), dict(op=L.F_LT, plane_a=1, thr=0.9)]
    locus_args = dict(min_callrate=0.8, min_hwep=1e-4, min_het=0.05, max_het=0.95, use_length=False)
    hb = HostBatch(h['gt'], np.full(Lc, 2), loci.allele_lens, loci.allele_strs)
    dev, ch = _dev_from_oracle(hb, planes, filters, 0, locus_args)
    dev = {k: v for k, v in dev.items() if k not in ('sample_counters', 'totaldp', 'dpmiss', 'loc_counters')}
    tables = (hb.allele_off, hb.len_class, hb.str_class, hb.len_class_value)
    args = (lambda lo, hi: (h['gt'][lo:hi], [p[lo:hi] for p in planes]),
            lambda lo, hi: (ch.gt_out[lo:hi], ch.mask[lo:hi]), Lc, S, tables, filters, 0, locus_args)
    fullsize.check_step(*args, dev)
    for key, poke in (('cnt_b', lambda a: a.__setitem__(5, a[5] + 1)), ('bits', lambda a: a.__setitem__(3, a[3] ^ 1)),
                      ('lf_a', lambda a: a.__setitem__((2, L.LF_HET_STR), a[2, L.LF_HET_STR] * (1 + 1e-6))),
                      ('li_b', lambda a: a.__setitem__((7, L.LI_N_HOM_STR), a[7, L.LI_N_HOM_STR] + 1))):
        bad = dict(dev)
        bad[key] = dev[key].copy()
        poke(bad[key])
        with pytest.raises(AssertionError):
            fullsize.check_step(*args, bad)
    m2 = ch.mask.copy()
    m2[11, 3] ^= 2
    with pytest.raises(AssertionError):
        fullsize.check_step(args[0], lambda lo, hi: (ch.gt_out[lo:hi], m2[lo:hi]), *args[2:], dev)


def test_popstr_support_and_ratio_opcodes_in_c():
    rng = np.random.default_rng(3)
    Lc, S, K = 12, 50, 5
    gt = rng.integers(-1, K, size=(Lc, S, 2)).astype(np.int16)
    ad = rng.integers(0, 9, size=(Lc, S, K)).astype(np.int32)
    dp = rng.integers(0, 40, size=(Lc, S)).astype(np.int32)
    st = rng.integers(0, 6, size=(Lc, S)).astype(np.int32)
    filters = [dict(op=L.F_AD_SUPPORT_LT, plane_a=0, thr=3), dict(op=L.F_RATIO_GT, plane_a=2, plane_b=1, thr=0.15)]
    lens = [[float(i) for i in range(K)]] * Lc
    hb = HostBatch(gt, np.full(Lc, 2), lens, [[str(i) for i in range(K)]] * Lc)
    ch, _, _, _ = OracleCompute().dumpstr_batch(hb, [ad, dp, st], filters, 1, {})
    g2, mask, counters, totaldp, dpmiss, err = oracle_c.call_filters(gt, [ad, dp, st], filters, dp_plane=1, n_threads=2)
    assert np.array_equal(mask, ch.mask) and np.array_equal(g2, ch.gt_out)
    assert np.array_equal(counters, ch.sample_counters) and np.array_equal(totaldp, ch.totaldp)
