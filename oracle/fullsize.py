"""ORACLE -- TEST INFRASTRUCTURE ONLY (imported by tests/ and bench.py's checker, never by trtools_amd/).

Exhaustive parity at the BASELINE.json sizes: EVERY locus of a device-resident call set against the compiled C
restatement (oracle/oracle_c.c, itself pinned to the numpy oracle and through it to the reference), run on all host
threads.  The call set is read back from the device in blocks of loci (the device generator's output IS the input
of both sides), pushed through

    statSTR  : counts + 11 statistics of GT                      tr_harmonizer.py:1420-1575, utils.py:118-338
    dumpSTR  : call filters -> GT', mask, sample_info             dumpSTR.py:613-774, filters.py:327-867
               counts + statistics of GT'                         (the rebuilt record, dumpSTR.py:748-774)
               locus filters + loc_info                           dumpSTR.py:917-973, filters.py:35-217

and compared with what the device produced: integers, masks, genotypes and filter decisions bit for bit, floats to
1e-9 relative (north_star's tolerance)."""
import numpy as np

from . import oracle_c

# columns of trk_stats_out (include/trk.h), restated here so that this module does not import the product
LI_N_CALLED, LI_N_LOWPLOIDY, LI_N_HOM_LEN, LI_N_HOM_STR, LI_N_ALLELES, LI_N_BAD, LI_HWE_STATUS_LEN, \
    LI_HWE_STATUS_STR, LI_N_SAMPLES = range(9)
LF_THRESH, LF_MEAN, LF_MODE, LF_VAR, LF_HET_LEN, LF_HET_STR, LF_ENTROPY_LEN, LF_ENTROPY_STR, LF_HWEP_LEN, \
    LF_HWEP_STR, LF_CALLRATE = range(11)
LC_TOTALCALLS, LC_PASS, LC_NO_CALLS, LC_FILTER0, LC_HWE_ERRORS = 0, 1, 2, 3, 31
FLOAT_COLS = [LF_THRESH, LF_MEAN, LF_MODE, LF_VAR, LF_HET_LEN, LF_HET_STR, LF_ENTROPY_LEN, LF_ENTROPY_STR,
              LF_HWEP_LEN, LF_HWEP_STR]
RTOL = 1e-9
P_TINY = 1e-280      # p-values are compared relative to themselves down to here


def _cmp_stats(tag, lo, cnt_d, li_d, lf_d, cnt_o, oi, of, n_samples):
    """Device rows [lo, lo + n) vs the oracle's; raises AssertionError naming the first differing locus."""
    def first(bad):
        return int(lo + np.flatnonzero(bad)[0])
    if not np.array_equal(cnt_d, cnt_o):
        raise AssertionError("%s: allele counts differ (block at locus %d)" % (tag, lo))
    for dcol, ocol, name in ((LI_N_CALLED, 0, 'n_called'), (LI_N_LOWPLOIDY, 1, 'n_lowploidy'),
                             (LI_N_HOM_LEN, 2, 'hom_len'), (LI_N_HOM_STR, 3, 'hom_str'), (LI_N_BAD, 4, 'n_bad'),
                             (LI_HWE_STATUS_LEN, 5, 'hwe_status_len'), (LI_HWE_STATUS_STR, 6, 'hwe_status_str'),
                             (LI_N_ALLELES, 7, 'n_alleles')):
        bad = li_d[:, dcol] != oi[:, ocol]
        if bad.any():
            raise AssertionError("%s: %s differs at locus %d: %d vs %d" % (tag, name, first(bad), li_d[bad, dcol][0],
                                                                           oi[bad, ocol][0]))
    worst = 0.0
    for j, c in enumerate(FLOAT_COLS):
        a, b = lf_d[:, c], of[:, j]
        bad = np.isnan(a) != np.isnan(b)
        if bad.any():
            raise AssertionError("%s: float column %d nan pattern differs at locus %d" % (tag, c, first(bad)))
        ok = ~np.isnan(a)
        aa, bb = a[ok], b[ok]
        if c in (LF_HWEP_LEN, LF_HWEP_STR):
            # p-values: RELATIVE to themselves all the way down (1e-200 against 1.1e-200 is a 10 % error, not an
            # absolute 1e-201), except where float64 runs out: below P_TINY both sides only have to be that small
            # (the exact test's pmf(k) underflows there; the device returns 0, an incomplete-beta tail may return a
            # denormal-range value -- tests/test_gpu_binomtest.py documents that boundary against scipy)
            tiny = np.abs(bb) < P_TINY
            err = np.abs(aa - bb) / np.maximum(np.abs(bb), P_TINY)
            err[tiny] = np.where(np.abs(aa[tiny]) < 10 * P_TINY, 0.0, np.inf)
        else:
            err = np.abs(aa - bb) / np.maximum(1.0, np.abs(bb))
            # values below 1 relative to themselves too, down to the rounding noise of a sum of ~1 terms
            small = np.abs(bb) < 1.0
            err[small] = np.minimum(err[small] * 1e3, np.abs(aa[small] - bb[small]) / np.maximum(np.abs(bb[small]), 1e-300))
        if err.size and err.max() > RTOL:
            i = np.flatnonzero(ok)[int(np.argmax(err))]
            raise AssertionError("%s: float column %d off by %.3g (rel) at locus %d: %r vs %r" %
                                 (tag, c, err.max(), lo + i, a[i], b[i]))
        if err.size:
            worst = max(worst, float(err.max()))
    cr = li_d[:, LI_N_CALLED] / float(n_samples) if n_samples else np.full(len(li_d), np.nan)
    bad = ~(np.abs(lf_d[:, LF_CALLRATE] - cr) <= 1e-15)
    if bad.any():
        raise AssertionError("%s: call rate differs at locus %d" % (tag, first(bad)))
    return worst


def check_group_stats(fetch_inputs, n_loci, n_samples, tables, group_masks, dev, block=4096, n_threads=None):
    """statSTR --samples at full size: EVERY locus of every sample group against oracle_c on the group's columns
    (statSTR.py:520-542: a group's statistics are the statistics of the record restricted to its samples).
    dev: cnt [G, sumA], li [G, L, cols], lf [G, L, cols] as the device wrote them."""
    off, lc, sc, cv = tables
    nt = n_threads or oracle_c.tuned_threads()
    worst = 0.0
    for lo in range(0, n_loci, block):
        hi = min(n_loci, lo + block)
        gt, _ = fetch_inputs(lo, hi)
        o = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        sl = slice(int(off[lo]), int(off[hi]))
        for g, m in enumerate(group_masks):
            sub = np.ascontiguousarray(gt[:, np.asarray(m, dtype=bool)])
            cnt_o, oi, of = oracle_c.batch_stats(sub, None, o, lc[sl], sc[sl], cv[sl], n_threads=nt)
            worst = max(worst, _cmp_stats('statSTR group %d' % g, lo, dev['cnt'][g][sl], dev['li'][g][lo:hi],
                                          dev['lf'][g][lo:hi], cnt_o, oi, of, int(np.sum(m))))
            if not np.all(dev['li'][g][lo:hi, LI_N_SAMPLES] == int(np.sum(m))):
                raise AssertionError("group %d: n_samples column differs in block %d" % (g, lo))
    return dict(loci=int(n_loci), groups=len(group_masks), worst_float_rel=worst, threads=nt)


def locus_filter_bits(oi, of, n_samples, min_callrate=None, min_hwep=None, min_het=None, max_het=None,
                      use_length=False):
    """filters.py:59-61, 98-103, 140-144, 181-185 + dumpSTR.py:957-971 from the oracle's per-locus rows: bit 0
    call rate, 1 HWE, 2 het low, 3 het high, 31 no calls remaining (a nan statistic never fires)."""
    n = oi.shape[0]
    bits = np.zeros(n, dtype=np.uint32)
    with np.errstate(invalid='ignore'):
        if min_callrate is not None:
            bits |= ((oi[:, 0] / float(n_samples)) < min_callrate).astype(np.uint32) << np.uint32(0)
        if min_hwep is not None:
            bits |= (of[:, 8 if use_length else 9] < min_hwep).astype(np.uint32) << np.uint32(1)
        het = of[:, 4 if use_length else 5]
        if min_het is not None:
            bits |= (het < min_het).astype(np.uint32) << np.uint32(2)
        if max_het is not None:
            bits |= (het > max_het).astype(np.uint32) << np.uint32(3)
    bits |= (oi[:, 0] == 0).astype(np.uint32) << np.uint32(31)
    return bits


def check_step(fetch_inputs, fetch_outputs, n_loci, n_samples, tables, filters, dp_plane, locus_args, dev,
               block=4096, full_outputs_upto=10000, n_threads=None, n_pad=0):
    """Every locus of one statSTR + dumpSTR step against oracle_c.

    fetch_inputs(lo, hi)  -> (gt int16 [n, S, 2], [plane arrays, interleaved, in the filters' plane order])
    fetch_outputs(lo, hi) -> (gt_out int16 [n, S, 2], mask uint32 [n, S]) as the device wrote them
    tables                -> (allele_off, len_class, str_class, len_class_value) of the shard
    dev                   -> dict of host copies of the device results:
        cnt_a, li_a, lf_a   statSTR (may be None: the dumpSTR half only)
        cnt_b, li_b, lf_b   dumpSTR after masking
        sample_counters [(1+nf), S], totaldp [S], dpmiss [S], bits [L], loc_counters [32]
    Returns a dict of what was covered (loci, calls compared bit for bit, worst float deviation)."""
    off, lc, sc, cv = tables
    nt = n_threads or oracle_c.tuned_threads()
    S = n_samples
    nf = len(filters)
    counters = np.zeros((1 + nf, S), dtype=np.int64)
    totaldp = np.zeros(S, dtype=np.int64)
    dpmiss = np.zeros(S, dtype=np.int64)
    loc = np.zeros(32, dtype=np.int64)
    worst = 0.0
    full_calls = 0
    use_length = bool(locus_args.get('use_length', False))
    spec = {k: locus_args.get(k) for k in ('min_callrate', 'min_hwep', 'min_het', 'max_het')}
    for lo in range(0, n_loci, block):
        hi = min(n_loci, lo + block)
        gt, planes = fetch_inputs(lo, hi)
        o = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        sl = slice(int(off[lo]), int(off[hi]))
        if dev.get('cnt_a') is not None:
            cnt_o, oi, of = oracle_c.batch_stats(gt, None, o, lc[sl], sc[sl], cv[sl], n_threads=nt)
            worst = max(worst, _cmp_stats('statSTR', lo, dev['cnt_a'][sl], dev['li_a'][lo:hi], dev['lf_a'][lo:hi],
                                          cnt_o, oi, of, S - n_pad))
        want_full = lo < full_outputs_upto
        g2, mask, c1, td, dm, err = oracle_c.call_filters(gt, planes, filters, dp_plane=dp_plane, n_threads=nt,
                                                          want_mask=want_full)
        if err[0]:
            raise AssertionError("oracle: negative depth on a PASS call in block %d" % lo)
        counters += c1
        totaldp += td
        dpmiss += dm
        if want_full:
            g_dev, m_dev = fetch_outputs(lo, hi)
            if not np.array_equal(m_dev, mask):
                bad = np.argwhere(m_dev != mask)[0]
                raise AssertionError("filter mask differs at locus %d sample %d: %#x vs %#x" %
                                     (lo + bad[0], bad[1], m_dev[tuple(bad)], mask[tuple(bad)]))
            if not np.array_equal(g_dev, g2):
                bad = np.argwhere(g_dev != g2)[0]
                raise AssertionError("masked genotype differs at locus %d sample %d" % (lo + bad[0], bad[1]))
            full_calls += (hi - lo) * S
        cnt_o, oi, of = oracle_c.batch_stats(g2, None, o, lc[sl], sc[sl], cv[sl], n_threads=nt)
        worst = max(worst, _cmp_stats('dumpSTR', lo, dev['cnt_b'][sl], dev['li_b'][lo:hi], dev['lf_b'][lo:hi],
                                      cnt_o, oi, of, S - n_pad))
        bits = locus_filter_bits(oi, of, S - n_pad, use_length=use_length, **spec)
        bad = bits != dev['bits'][lo:hi]
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise AssertionError("locus filter bits differ at locus %d: %#x vs %#x" % (lo + i, dev['bits'][lo + i], bits[i]))
        for k in range(4):
            loc[LC_FILTER0 + k] += int(((bits >> np.uint32(k)) & np.uint32(1)).sum())
        loc[LC_NO_CALLS] += int((oi[:, 0] == 0).sum())
        loc[LC_PASS] += int((bits == 0).sum())
        loc[LC_TOTALCALLS] += int(oi[bits == 0, 0].sum())
        if spec['min_hwep'] is not None:
            st = oi[:, 5 if use_length else 6]
            loc[LC_HWE_ERRORS] += int(((st == 2) | (st == 3)).sum())
    if dev.get('sample_counters') is not None:      # cohort-wide sums: only comparable when `dev` holds this shard alone
        if not np.array_equal(dev['sample_counters'], counters):
            raise AssertionError("per-sample counters (sample_info) differ")
        if not np.array_equal(dev['totaldp'], totaldp) or not np.array_equal(dev['dpmiss'], dpmiss):
            raise AssertionError("per-sample depth sums differ")
        if not np.array_equal(dev['loc_counters'], loc):
            raise AssertionError("loc_info counters differ: %r vs %r" % (dev['loc_counters'][:8], loc[:8]))
    return dict(loci=int(n_loci), calls=int(n_loci) * int(S), calls_bit_for_bit=int(full_calls),
                worst_float_rel=worst, threads=nt, sums=(counters, totaldp, dpmiss, loc))


def check_assoc(fetch_gt, n_loci, n_samples, tables_off, allele_len, x, y, dev_int, dev_f64, sample_in=None,
                non_major_cutoff=20.0, block=4096, n_threads=None):
    """associaTR scan (SURVEY 8 row f3 / BASELINE configs[4]) at full size: EVERY locus against oracle_c's
    orc_assoc_locus (pinned to oracle/associatr_oracle.py by tests/test_oracle_c.py).

    fetch_gt(lo, hi) -> int16 [n, S, 2]; x [S, M] design (column 0 reserved, column 1 ones, covariates), y [S];
    dev_int [L, >= 2] (n_tested, status), dev_f64 [L, >= 4] (p, coef_std, se_std, R^2) as the device wrote them.
    Tested-sample counts and the filter decision of every locus exactly; the four statistics to 1e-9 relative
    (R^2 near 0: 1e-13 absolute).  Returns dict(loci, regressed, worst_rel)."""
    nt = n_threads or oracle_c.tuned_threads()
    off = np.asarray(tables_off)
    worst, regressed = 0.0, 0
    for lo in range(0, n_loci, block):
        hi = min(n_loci, lo + block)
        gt = fetch_gt(lo, hi)
        o = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        oi, of = oracle_c.assoc_scan(gt, o, allele_len[int(off[lo]):int(off[hi])], x, y, sample_in=sample_in,
                                     non_major_cutoff=non_major_cutoff, n_threads=nt)
        di, df = dev_int[lo:hi], dev_f64[lo:hi]
        bad = di[:, 0] != oi[:, 0]
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise AssertionError("assoc: n_tested differs at locus %d: %d vs %d" % (lo + i, di[i, 0], oi[i, 0]))
        # statuses 0-4 are the reference's decisions; 5 (oracle: degenerate) / 5-6 (device) carry no statistics
        o_ok, d_ok = oi[:, 1] == 0, di[:, 1] == 0
        degenerate = (oi[:, 1] == 5) | (di[:, 1] >= 5)
        bad = (oi[:, 1] != np.minimum(di[:, 1], 5)) & ~degenerate
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise AssertionError("assoc: filter decision differs at locus %d: %d vs %d" % (lo + i, di[i, 1], oi[i, 1]))
        both = o_ok & d_ok
        regressed += int(both.sum())
        for col in range(4):
            a, b = df[both, col], of[both, col]
            err = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
            if col == 3:
                err = np.where(np.abs(a - b) <= 1e-13, 0.0, err)
            if err.size and not (err.max() <= RTOL):
                i = int(np.flatnonzero(both)[int(np.nanargmax(err))])
                raise AssertionError("assoc: column %d off by %.3g (rel) at locus %d: %r vs %r" %
                                     (col, np.nanmax(err), lo + i, df[i, col], of[i, col]))
            if err.size:
                worst = max(worst, float(err.max()))
    return dict(loci=int(n_loci), regressed=regressed, worst_rel=worst, threads=nt)
