"""Oracle-backed stand-in for trtools_amd.compute.DeviceCompute (TESTS ONLY).

Same methods, results computed one locus at a time by oracle/trtools_oracle.py.
It lets the CPU test-suite drive the host layer (VCF decoding, harmonisation,
batch packing, text formatting, counters/log writing) end to end against the
reference's golden files on a machine without a GPU, and it is the checker the
GPU tests compare DeviceCompute against.  Never imported by the product."""
import collections
import math

import numpy as np

from oracle import trtools_oracle as orc
from trtools_amd import _lib as L
from trtools_amd.compute import StatsHost, CallHost


def _groups(hb):
    if hb.group_bits is None:
        return [None]
    return [((hb.group_bits >> g) & 1).astype(bool) for g in range(hb.n_groups)]


def stats_of(hb, gt, nalleles_thresh):
    groups = _groups(hb)
    G, Lc = len(groups), hb.n_loci
    cnt = np.zeros((G, int(hb.allele_off[-1])), dtype=np.int32)
    li = np.zeros((G, Lc, L.TRK_LI_COLS), dtype=np.int32)
    lf = np.full((G, Lc, L.TRK_LF_COLS), np.nan)
    for l in range(Lc):
        pl = int(hb.locus_ploidy[l])
        g = gt[l][:, :pl]
        o, e = int(hb.allele_off[l]), int(hb.allele_off[l + 1])
        for gi, si in enumerate(groups):
            ol = orc.locus_stats(g, hb.allele_lens[l], hb.allele_strs[l], si, True, nalleles_thresh)
            os_ = orc.locus_stats(g, hb.allele_lens[l], hb.allele_strs[l], si, False, nalleles_thresh)
            cnt[gi, o:e] = ol['index_counts']
            I, F = li[gi, l], lf[gi, l]
            I[L.LI_N_CALLED] = ol['n_called']
            I[L.LI_N_SAMPLES] = ol['n_samples']
            I[L.LI_N_ALLELES] = int(ol['index_counts'].sum())
            I[L.LI_NALLELES_LEN], I[L.LI_NALLELES_STR] = ol['nalleles'], os_['nalleles']
            for scol, fcol, o_ in ((L.LI_HWE_STATUS_LEN, L.LF_HWEP_LEN, ol), (L.LI_HWE_STATUS_STR, L.LF_HWEP_STR, os_)):
                if o_['hwep_status'] != orc.HWE_OK:
                    I[scol] = o_['hwep_status']
                elif math.isnan(o_['hwep']):
                    I[scol] = L.HWE_NAN
                else:
                    I[scol] = L.HWE_OK
                    F[fcol] = o_['hwep']
            F[L.LF_THRESH], F[L.LF_MEAN], F[L.LF_MODE], F[L.LF_VAR] = ol['thresh'], ol['mean'], ol['mode'], ol['var']
            F[L.LF_HET_LEN], F[L.LF_HET_STR] = ol['het'], os_['het']
            F[L.LF_ENTROPY_LEN], F[L.LF_ENTROPY_STR] = ol['entropy'], os_['entropy']
            F[L.LF_CALLRATE] = ol['n_called'] / ol['n_samples'] if ol['n_samples'] else np.nan
    return StatsHost(cnt, li, lf)


def _eval_filter(f, planes, l, g):
    """One trk_call_filter spec on locus l through the oracle's filt_* functions."""
    op = f['op']
    pa = planes[f['plane_a']][l]
    pa = pa.reshape(pa.shape[0], -1)
    ca = f.get('col_a', 0)
    # what trk_call_filters refuses (TRK_ERR_ARG "col_a out of range"), the stand-in refuses too: a plane narrower than
    # the columns a filter names must not pass on the CPU and fail on the device
    for key, plane_key in (('col_a', 'plane_a'), ('col_a2', 'plane_a'), ('col_b', 'plane_b')):
        if key in f and f.get(plane_key, -1) >= 0:
            width = int(np.prod(planes[f[plane_key]].shape[2:], dtype=np.int64)) if planes[f[plane_key]].ndim > 2 else 1
            if not 0 <= f[key] < width:
                raise ValueError("filter spec: %s = %d out of range for a plane of %d columns" % (key, f[key], width))
    thr = f.get('thr', 0.0)
    if pa.dtype == np.int32 and float(thr) == int(thr):
        thr = int(thr)
    S = pa.shape[0]
    called = orc.get_called_samples(g)
    if op == L.F_LT:
        return orc.filt_min_value(pa[:, ca:ca + 1], thr)
    if op == L.F_GT:
        return orc.filt_max_value(pa[:, ca:ca + 1], thr)
    if op == L.F_RATIO_GT:
        pb = planes[f['plane_b']][l].reshape(S, -1)
        cb = f.get('col_b', 0)
        return orc.filt_ratio_gt(pa[:, ca:ca + 1], pb[:, cb:cb + 1], thr)
    out = np.full(S, np.nan)
    if not np.any(called):
        return out
    if op == L.F_CALLED_LT:
        v = pa[called, ca]
        out[np.nonzero(called)[0][v < thr]] = v[v < thr]
        return out
    if op == L.F_CALLED_SUM_LT:
        v = pa[called, ca] + pa[called, f['col_a2']]
        out[np.nonzero(called)[0][v < thr]] = v[v < thr]
        return out
    pb = planes[f['plane_b']][l].reshape(S, -1) if f.get('plane_b', -1) >= 0 else None
    if op == L.F_CALLED_EQ:
        v = pa[called, ca].astype(np.int64)
        hit = v == pb[called, f.get('col_b', 0)]
        out[np.nonzero(called)[0][hit]] = v[hit]
        return out
    if op == L.F_CALLED_SUM_EQ:
        v = pa[called, ca].astype(np.int64) + pa[called, f['col_a2']]
        hit = v == pb[called, f.get('col_b', 0)]
        out[np.nonzero(called)[0][hit]] = v[hit]
        return out
    if op == L.F_CALLED_OUTSIDE_CI:
        if pb.shape[1] != 2 * pa.shape[1]:       # (trk_call_filters: "REPCI plane needs 2 columns per REPCN column")
            raise ValueError("filter spec: a REPCI plane of %d columns for a REPCN plane of %d" % (pb.shape[1], pa.shape[1]))
        cis = np.array(['%d-%d,%d-%d' % tuple(r[:4]) for r in pb])
        return orc.filt_gangstr_bad_ci(g, pa, cis)
    if op == L.F_AD_SUPPORT_LT:
        return orc.filt_popstr_require_support(g, pa, thr)
    raise ValueError("unknown op %r" % op)


class OracleCompute:
    def locus_stats(self, hb, nalleles_thresh=0.01):
        return stats_of(hb, hb.gt, nalleles_thresh)

    def dumpstr_batch(self, hb, planes, filters, dp_plane, locus_spec, nalleles_thresh=0.01):
        Lc, S = hb.n_loci, hb.n_samples
        nf = len(filters)
        info = collections.OrderedDict()
        info['numcalls'] = np.zeros(S, dtype=np.int64)
        info['totaldp'] = np.zeros(S, dtype=float)
        names = ['f%d' % k for k in range(nf)]
        for n in names:
            info[n] = np.zeros(S, dtype=np.int64)
        gout = hb.gt.copy()
        mask = np.zeros((Lc, S), dtype=np.uint32)
        dp_missing = np.zeros(S, dtype=np.int64)
        for l in range(Lc):
            pl = int(hb.locus_ploidy[l])
            g = hb.gt[l][:, :pl]
            outs = [(names[k], _eval_filter(filters[k], planes, l, g)) for k in range(nf)]
            for k, (_, o) in enumerate(outs):
                mask[l] |= (~np.isnan(o)).astype(np.uint32) << np.uint32(k)
            mask[l] |= (~orc.get_called_samples(g)).astype(np.uint32) << np.uint32(31)
            dp = None if dp_plane < 0 else planes[dp_plane][l].reshape(S, -1)[:, :1]
            mg, _ = orc.apply_call_filters(g, outs, info, dp=dp)
            gout[l][:, :pl] = mg
        totaldp = info['totaldp'].copy()
        dp_missing[np.isnan(totaldp)] = 1
        totaldp[np.isnan(totaldp)] = 0
        counters = np.stack([info['numcalls']] + [info[n] for n in names]).astype(np.int64)
        if np.all(totaldp == np.floor(totaldp)):
            totaldp = totaldp.astype(np.int64)
        ch = CallHost(gout, mask, counters, totaldp, dp_missing, np.zeros(4, dtype=np.int32))
        st = stats_of(hb, gout, nalleles_thresh)
        # locus filters (dumpSTR.py:917-973) from the same statistics
        spec = dict(locus_spec)
        ext = spec.pop('extern_bits', None)
        n_ext = spec.pop('n_extern', 0)
        ul = spec.get('use_length', False)
        bits = np.zeros(Lc, dtype=np.uint32)
        lc = np.zeros(L.TRK_LC_COLS, dtype=np.int64)
        for l in range(Lc):
            I, F = st.locus_int[0, l], st.locus_f64[0, l]
            b = 0
            if spec.get('min_callrate') is not None and F[L.LF_CALLRATE] < spec['min_callrate']:
                b |= 1 << L.LOCF_CALLRATE
            if spec.get('min_hwep') is not None:
                if I[L.LI_HWE_STATUS_LEN if ul else L.LI_HWE_STATUS_STR] in (L.HWE_VALUE_ERROR, L.HWE_INDEX_ERROR):
                    lc[L.LC_HWE_ERRORS] += 1
                if F[L.LF_HWEP_LEN if ul else L.LF_HWEP_STR] < spec['min_hwep']:
                    b |= 1 << L.LOCF_HWE
            het = F[L.LF_HET_LEN if ul else L.LF_HET_STR]
            if spec.get('min_het') is not None and het < spec['min_het']:
                b |= 1 << L.LOCF_HETLOW
            if spec.get('max_het') is not None and het > spec['max_het']:
                b |= 1 << L.LOCF_HETHIGH
            if ext is not None:
                b |= (int(ext[l]) & ((1 << n_ext) - 1)) << L.LOCF_EXTERN0
            for k in range(28):
                if (b >> k) & 1:
                    lc[L.LC_FILTER0 + k] += 1
            if I[L.LI_N_CALLED] == 0:
                b |= 1 << L.LOCF_NO_CALLS
                lc[L.LC_NO_CALLS] += 1
            if b == 0:
                lc[L.LC_PASS] += 1
                lc[L.LC_TOTALCALLS] += int(I[L.LI_N_CALLED])
            bits[l] = b
        return ch, st, bits, lc

    def assoc_batch(self, hb, vec, sample_in, non_major_cutoff, precision=2):
        """trk_assoc_scan through oracle/associatr_oracle.py (one locus at a time)."""
        from oracle import associatr_oracle as ao
        from trtools_amd.compute import AssocHost
        S, M = hb.n_samples, vec.shape[0]
        sf = np.ones(S, dtype=bool) if sample_in is None else np.asarray(sample_in, dtype=bool)
        covars = np.ones((int(sf.sum()), M + 1))
        outcome = vec[0, sf]
        for k in range(1, M):
            covars[:, 1 + k] = vec[k, sf]
        li = np.zeros((hb.n_loci, L.AI_COLS), dtype=np.int32)
        lf = np.full((hb.n_loci, L.AF_COLS), np.nan)
        cnt = np.zeros(int(hb.allele_off[-1]), dtype=np.int32)
        code = {'No called samples': L.AS_NO_CALLED, 'Only one called allele': L.AS_ONE_ALLELE}
        for l in range(hb.n_loci):
            g = hb.gt[l][:, :int(hb.locus_ploidy[l])]
            gi = ao.locus_genotypes(g, hb.allele_lens[l], sf, non_major_cutoff, precision)
            n = int(np.sum(gi['called_samples_filter']))
            li[l, L.AI_N_TESTED] = n
            li[l, L.AI_N_RALLELES] = len(gi['allele_frequency'])
            curr = sf & ~np.any(g == -1, axis=1)
            sel = g[curr]
            o, e = int(hb.allele_off[l]), int(hb.allele_off[l + 1])
            cnt[o:e] = np.bincount(sel[sel >= 0].astype(int), minlength=e - o)
            li[l, L.AI_N_HAPS] = int(cnt[o:e].sum())
            reason = gi['locus_filtered']
            if reason:
                li[l, L.AI_STATUS] = code.get(reason, L.AS_NON_MAJOR)
                continue
            if M + 1 >= n:
                li[l, L.AI_STATUS] = L.AS_N_COVARS
                continue
            summed = np.sum(gi['gts'], axis=1)
            if np.std(summed) <= 1e-9 * max(1.0, abs(np.mean(summed))):
                li[l, L.AI_STATUS] = L.AS_ZERO_VARIANCE
                continue
            r = ao.locus_regression(gi['gts'], gi['called_samples_filter'], covars, outcome, 1.0)
            lf[l, L.AF_PVAL], lf[l, L.AF_COEF], lf[l, L.AF_SE] = r['pval'], r['coef_std'], r['se_std']
            lf[l, L.AF_RSQUARED], lf[l, L.AF_GT_STD], lf[l, L.AF_TVALUE] = r['rsquared'], r['std'], r['tvalue']
            lf[l, L.AF_DF_RESID], lf[l, L.AF_GT_MEAN] = r['df_resid'], np.mean(summed)
        return AssocHost(li, lf, cnt)

    def assoc_dosage_batch(self, hb, vec, sample_in, ap1, ap2, precision=2):
        """trk_assoc_scan_dosage through oracle/associatr_oracle.py."""
        from oracle import associatr_oracle as ao
        from trtools_amd.compute import AssocHost
        from trtools_amd.synth import pack_dosage_tables
        S, M = hb.n_samples, vec.shape[0]
        sf = np.ones(S, dtype=bool) if sample_in is None else np.asarray(sample_in, dtype=bool)
        covars = np.ones((int(sf.sum()), M + 1))
        outcome = vec[0, sf]
        for k in range(1, M):
            covars[:, 1 + k] = vec[k, sf]
        tabs = pack_dosage_tables(hb.allele_lens, precision)
        li = np.zeros((hb.n_loci, L.AI_COLS), dtype=np.int32)
        lf = np.full((hb.n_loci, L.AF_COLS), np.nan)
        cs = np.zeros((int(hb.allele_off[-1]), L.ADC_COLS))
        ls = np.zeros((hb.n_loci, L.ADL_COLS))
        for l in range(hb.n_loci):
            g = hb.gt[l][:, :int(hb.locus_ploidy[l])]
            A = len(hb.allele_lens[l])
            gi = ao.locus_genotypes(g, hb.allele_lens[l], sf, 0.0, precision, ap1[l][:, :A - 1], ap2[l][:, :A - 1])
            n = int(np.sum(gi['called_samples_filter']))
            li[l, L.AI_N_TESTED] = n
            curr = sf & ~np.any(g == -1, axis=1)
            # the sums the device reports, recomputed from the oracle's per-length dosage matrices
            lens = [round(float(x), precision) for x in hb.allele_lens[l]]
            uniq = np.unique(lens)
            gts = {u: np.zeros((n, 2)) for u in uniq}
            for p_, ap in ((0, ap1[l]), (1, ap2[l])):
                gts[lens[0]][:, p_] += np.maximum(0, 1 - np.sum(ap[curr, :A - 1], axis=1))
                for i in range(A - 1):
                    gts[lens[i + 1]][:, p_] += ap[curr, i]
            lut = np.array([*[float(x) for x in hb.allele_lens[l]], -2, -1])
            best = lut[g.astype(int)][curr, :]
            rbest = np.around(best, precision)
            o = int(hb.allele_off[l])
            for k, u in enumerate(uniq):
                d = gts[u].reshape(-1)
                x = (rbest == u).reshape(-1).astype(float)
                cs[o + k] = [d.sum(), (d * d).sum(), x.sum(), (x * d).sum()]
            y = np.add.reduce([u * gts[u] for u in uniq]).reshape(-1) if n else np.zeros(0)
            x = best.reshape(-1)
            ls[l] = [x.sum(), (x * x).sum(), y.sum(), (y * y).sum(), (x * y).sum(), 2 * n,
                     x.min() if n else np.inf, x.max() if n else -np.inf]
            if M + 1 >= n:
                li[l, L.AI_STATUS] = L.AS_N_COVARS
                continue
            summed = np.sum([u * np.sum(gts[u], axis=1) for u in uniq], axis=0)
            if np.std(summed) <= 1e-12 * max(1.0, abs(np.mean(summed))):
                li[l, L.AI_STATUS] = L.AS_ZERO_VARIANCE
                continue
            r = ao.locus_regression(gts, gi['called_samples_filter'], covars, outcome, 1.0, dosages=True)
            lf[l, L.AF_PVAL], lf[l, L.AF_COEF], lf[l, L.AF_SE] = r['pval'], r['coef_std'], r['se_std']
            lf[l, L.AF_RSQUARED], lf[l, L.AF_GT_STD], lf[l, L.AF_TVALUE] = r['rsquared'], r['std'], r['tvalue']
            lf[l, L.AF_DF_RESID] = r['df_resid']
        return AssocHost(li, lf, np.zeros(int(hb.allele_off[-1]), dtype=np.int32)), cs, ls, tabs

    def dosages_batch(self, hb, dosage_type, ap1=None, ap2=None):
        """trk_dosages through oracle.get_dosages; error bits recomputed from the inputs."""
        out = np.full((hb.n_loci, hb.n_samples), np.nan, dtype=np.float32)
        err = np.zeros(hb.n_loci, dtype=np.int32)
        for l in range(hb.n_loci):
            g = hb.gt[l][:, :int(hb.locus_ploidy[l])]
            A = len(hb.allele_lens[l])
            a1 = None if ap1 is None else ap1[l][:, :A - 1]
            a2 = None if ap2 is None else ap2[l][:, :A - 1]
            if a1 is not None:
                if np.any(np.sum(a1, axis=1) > 1.1) or np.any(np.sum(a2, axis=1) > 1.1):
                    err[l] |= 1
                if np.any(a1 < 0) or np.any(a2 < 0):
                    err[l] |= 2
            if err[l]:
                continue
            d = orc.get_dosages(g, hb.allele_lens[l], dosage_type, a1, a2)
            if dosage_type.endswith('_norm') and np.all(np.isnan(d)) and not np.all(np.any(g < 0, axis=1)):
                err[l] |= 4
            out[l] = d
        return out, err

    def qc_batch(self, hb, quality=None, sample_index=None, ignore_no_call=False):
        """trk_qc_reduce through oracle.qc_record, one record at a time (the arrays DeviceCompute.qc_batch returns)."""
        L_, S = hb.n_loci, hb.n_samples
        sel = np.ones(S, dtype=bool) if sample_index is None else np.asarray(sample_index, dtype=bool)
        idx = np.flatnonzero(sel)
        out = dict(sample_calls=np.zeros(S, dtype=np.int64), locus_calls=np.zeros(L_, dtype=np.int64))
        if quality is not None:
            out.update(sample_qual_sum=np.zeros(S), sample_qual_n=np.zeros(S, dtype=np.int64),
                       locus_qual_sum=np.zeros(L_), locus_qual_n=np.zeros(L_, dtype=np.int64))
        for l in range(L_):
            g = hb.gt[l][:, :int(hb.locus_ploidy[l])]
            calls, q, _ = orc.qc_record(g, None if quality is None else np.asarray(quality[l]).reshape(-1, 1), sel,
                                        ignore_no_call)
            out['sample_calls'][idx] += calls
            out['locus_calls'][l] = calls.sum()
            if q is not None:
                v = q.reshape(-1).astype(np.float64)
                ok = ~np.isnan(v)
                out['sample_qual_sum'][idx[ok]] += v[ok]
                out['sample_qual_n'][idx[ok]] += 1
                out['locus_qual_sum'][l] = v[ok].sum()
                out['locus_qual_n'][l] = ok.sum()
        return out
