/*
 * ORACLE (C half) -- TEST INFRASTRUCTURE ONLY; never linked into libtrk.so.
 *
 * Plain-C restatement of the integer reductions of the TRTools per-locus hot
 * path, written for speed on one host core so that the GPU results can be
 * checked at sizes the numpy oracle cannot reach and so that bench.py has a
 * compiled CPU baseline next to the numpy port.  Each function cites the
 * reference lines (gymrek-lab/TRTools v6.1.0) whose arithmetic it follows.
 * It is itself pinned against oracle/trtools_oracle.py (which is pinned against
 * the real reference) by tests/test_oracle_c.py.
 *
 * Independent of csrc/: own histogram code, own binomial pmf (lgamma based) and
 * tail summation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* One locus.  gt: [S, P] allele indices (-1 missing, -2 padding); only the first
 * `pl` columns belong to the record.  lc/sc: class rank of every allele index.
 * cnt[A]  <- GetAlleleCounts(index=True)              tr_harmonizer.py:1420-1499
 * out[0]  <- sum(GetGenotypeCounts().values())        tr_harmonizer.py:1326-1418 (rows without -1)
 * out[1]  <- rows without -1 that hold a -2
 * out[2]  <- num_hom by length class, out[3] by sequence class   utils.py:327-333
 * out[4]  <- rows holding an index >= A                                               */
void orc_locus_counts(const int16_t* gt, int S, int P, int pl, int A, const uint16_t* lc, const uint16_t* sc,
                      int32_t* cnt, int32_t* out) {
    memset(cnt, 0, sizeof(int32_t) * (size_t)A);
    int32_t n_called = 0, n_low = 0, n_hl = 0, n_hs = 0, n_bad = 0;
    for (int s = 0; s < S; ++s) {
        const int16_t* row = gt + (size_t)s * P;
        int miss = 0, low = 0, bad = 0;
        for (int j = 0; j < pl; ++j) {
            int a = row[j];
            if (a == -1) miss = 1;
            else if (a == -2) low = 1;
            else if (a >= A) bad = 1;
            else if (a >= 0) cnt[a]++;
        }
        n_bad += bad;
        if (miss) continue;
        n_called++;
        if (low) { n_low++; continue; }
        if (pl < 2 || bad) continue;
        /* sorted tuple: gt[0] == gt[1]  <=>  the smallest class occurs at least twice */
        int minl = 1 << 30, mins = 1 << 30, cl = 0, cs = 0;
        for (int j = 0; j < pl; ++j) {
            int l = lc[row[j]], q = sc[row[j]];
            if (l < minl) { minl = l; cl = 1; } else if (l == minl) cl++;
            if (q < mins) { mins = q; cs = 1; } else if (q == mins) cs++;
        }
        n_hl += cl >= 2;
        n_hs += cs >= 2;
    }
    out[0] = n_called; out[1] = n_low; out[2] = n_hl; out[3] = n_hs; out[4] = n_bad;
}

/* ln Gamma(x), x >= 1: Stirling's series above 15 (next term 691 / (360360 x^11) < 1e-17), the recurrence below */
static double orc_lgamma(double x) {
    double shift = 0.0;
    while (x < 15.0) { shift += log(x); x += 1.0; }
    const double xi = 1.0 / x, x2 = xi * xi;
    const double ser = xi * (1.0 / 12.0 - x2 * (1.0 / 360.0 - x2 * (1.0 / 1260.0 - x2 * (1.0 / 1680.0 - x2 * (1.0 / 1188.0)))));
    return (x - 0.5) * log(x) - x + 0.91893853320467274178 + ser - shift;
}

static double orc_binom_pmf(int64_t k, int64_t n, double p) {
    if (k < 0 || k > n) return 0.0;
    if (p <= 0.0) return k == 0 ? 1.0 : 0.0;
    if (p >= 1.0) return k == n ? 1.0 : 0.0;
    /* own log-gamma: libm's lgamma() writes the global `signgam` on every call (one cache line bouncing between
     * all cores once the loci are split over threads) and lgamma_r() scaled even worse on the hosts tried */
    double lg = orc_lgamma((double)n + 1.0) - orc_lgamma((double)k + 1.0) - orc_lgamma((double)(n - k) + 1.0);
    return exp(lg + (double)k * log(p) + (double)(n - k) * log1p(-p));
}
static double orc_lower(int64_t k, int64_t n, double p) {
    if (k < 0) return 0.0;
    double s = 0.0;
    for (int64_t i = k; i >= 0; --i) {
        double t = orc_binom_pmf(i, n, p);
        s += t;
        if (t <= s * 1e-17 && (double)i < p * (double)n) break;
    }
    return s;
}
static double orc_upper(int64_t k, int64_t n, double p) { /* sf(k) */
    double s = 0.0;
    for (int64_t i = k + 1; i <= n; ++i) {
        double t = orc_binom_pmf(i, n, p);
        s += t;
        if (t <= s * 1e-17 && (double)i > p * (double)n) break;
    }
    return s;
}
static int64_t orc_bsearch(double sign, double d, int64_t lo, int64_t hi, int64_t n, double p) {
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        double v = sign * orc_binom_pmf(mid, n, p);
        if (v < d) lo = mid + 1; else if (v > d) hi = mid - 1; else return mid;
    }
    return (sign * orc_binom_pmf(lo, n, p) <= d) ? lo : lo - 1;
}
/* scipy.stats.binomtest(k, n, p).pvalue, two-sided (call site utils.py:334-338).  lgamma-based pmf:
 * ~1e-11 relative at n = 1e4, inside the 1e-9 bar. */
double orc_binomtest(int64_t k, int64_t n, double p) {
    double d = orc_binom_pmf(k, n, p), rerr = 1.0 + 1e-7, pn = p * (double)n, pv;
    if ((double)k == pn) return 1.0;
    if ((double)k < pn) {
        int64_t ix = orc_bsearch(-1.0, -d * rerr, (int64_t)ceil(pn), n, n, p);
        int64_t y = n - ix + (d * rerr == orc_binom_pmf(ix, n, p));
        pv = orc_lower(k, n, p) + orc_upper(n - y, n, p);
    } else {
        int64_t ix = orc_bsearch(1.0, d * rerr, 0, (int64_t)floor(pn), n, p);
        pv = orc_lower(ix, n, p) + orc_upper(k - 1, n, p);
    }
    return pv < 1.0 ? pv : 1.0;
}

/* Scalars of one allele-class mode from class counts cc[ncls] (ascending class = dict order).
 * res: het, entropy, hwep (nan when undefined), status (0 ok, 1 nan, 2 ValueError, 3 IndexError).
 * utils.py:118-212, 298-338 */
void orc_mode_stats(const int32_t* cc, int ncls, int n_called, int n_low, int n_hom, int pl, double* res) {
    double nanv = nan("");
    res[0] = res[1] = res[2] = nanv; res[3] = 1;
    int64_t total = 0;
    for (int c = 0; c < ncls; ++c) total += cc[c];
    if (total <= 0) return;
    double ft = (double)total, fsum = 0.0, sq = 0.0;
    for (int c = 0; c < ncls; ++c) { if (!cc[c]) continue; double f = cc[c] / ft; fsum += f; sq += f * f; }
    if (!(fabs(1.0 - fsum) <= 0.001)) return;
    res[0] = 1.0 - sq;
    double ent = 0.0;
    for (int c = 0; c < ncls; ++c) { if (!cc[c]) continue; double pk = (cc[c] / ft) / fsum; ent -= pk * log(pk); }
    res[1] = ent / log(2.0) + 0.0;
    if (n_called == 0) { res[3] = 2; return; }
    if (pl < 2) { res[3] = 3; return; }
    if (n_low > 0) { res[3] = 1; return; }
    res[3] = 0;
    res[2] = orc_binomtest(n_hom, n_called, sq);
}

/* mean / mode / variance / max by length (utils.py:215-296, tr_harmonizer.py:1542-1575) */
void orc_length_stats(const int32_t* ccl, const double* cv, int ncls, double* res) {
    double nanv = nan("");
    res[0] = res[1] = res[2] = res[3] = nanv; /* thresh, mean, mode, var */
    int64_t total = 0;
    for (int c = 0; c < ncls; ++c) total += ccl[c];
    if (total <= 0) return;
    double ft = (double)total, fsum = 0.0;
    int best = -1, bestn = 0;
    for (int c = 0; c < ncls; ++c) {
        if (!ccl[c]) continue;
        fsum += ccl[c] / ft;
        res[0] = cv[c];
        if (ccl[c] > bestn) { bestn = ccl[c]; best = c; }
    }
    if (!(fabs(1.0 - fsum) <= 0.001)) return;
    double m = 0.0, v = 0.0;
    for (int c = 0; c < ncls; ++c) if (ccl[c]) m += cv[c] * (ccl[c] / ft);
    for (int c = 0; c < ncls; ++c) if (ccl[c]) { double d = cv[c] - m; v += (ccl[c] / ft) * (d * d); }
    res[1] = m; res[2] = cv[best]; res[3] = v;
}

static void orc_locus_all(const int16_t* gt, int l, int S, int P, const uint8_t* locus_ploidy, const int32_t* off,
                          const uint16_t* lc, const uint16_t* sc, const double* cv, int32_t* cnt, int32_t* out_i,
                          double* out_f, int32_t* ccl, int32_t* ccs);

/* Whole batch, P == 2 layout [L,S,P]: statSTR statistics of every locus (one group).
 * out_i[l*8 ..]: n_called, n_low, hom_len, hom_str, n_bad, status_len, status_str, n_alleles
 * out_f[l*10..]: thresh, mean, mode, var, het_len, het_str, ent_len, ent_str, hwep_len, hwep_str */
void orc_batch_stats(const int16_t* gt, int L, int S, int P, const uint8_t* locus_ploidy, const int32_t* off,
                     const uint16_t* lc, const uint16_t* sc, const double* cv, int32_t* cnt, int32_t* out_i,
                     double* out_f) {
    int maxA = 0;
    for (int l = 0; l < L; ++l) if (off[l + 1] - off[l] > maxA) maxA = off[l + 1] - off[l];
    int32_t* ccl = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxA + 1));
    int32_t* ccs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxA + 1));
    for (int l = 0; l < L; ++l) orc_locus_all(gt, l, S, P, locus_ploidy, off, lc, sc, cv, cnt, out_i, out_f, ccl, ccs);
    free(ccl); free(ccs);
}

/* the same, loci split over `n_threads` host threads (OpenMP): the all-cores CPU baseline of bench.py and the
 * checker of the full-size parity runs (every locus of a 100k x 10k call set in seconds) */
void orc_batch_stats_mt(const int16_t* gt, int L, int S, int P, const uint8_t* locus_ploidy, const int32_t* off,
                        const uint16_t* lc, const uint16_t* sc, const double* cv, int32_t* cnt, int32_t* out_i,
                        double* out_f, int n_threads) {
    int maxA = 0;
    for (int l = 0; l < L; ++l) if (off[l + 1] - off[l] > maxA) maxA = off[l + 1] - off[l];
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        int32_t* ccl = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxA + 1));
        int32_t* ccs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxA + 1));
#pragma omp for schedule(dynamic, 8)
        for (int l = 0; l < L; ++l) orc_locus_all(gt, l, S, P, locus_ploidy, off, lc, sc, cv, cnt, out_i, out_f, ccl, ccs);
        free(ccl); free(ccs);
    }
}

static void orc_locus_all(const int16_t* gt, int l, int S, int P, const uint8_t* locus_ploidy, const int32_t* off,
                          const uint16_t* lc, const uint16_t* sc, const double* cv, int32_t* cnt, int32_t* out_i,
                          double* out_f, int32_t* ccl, int32_t* ccs) {
    {
        int o = off[l], A = off[l + 1] - o, pl = locus_ploidy ? locus_ploidy[l] : P;
        int32_t r[5];
        orc_locus_counts(gt + (size_t)l * S * P, S, P, pl, A, lc + o, sc + o, cnt + o, r);
        memset(ccl, 0, sizeof(int32_t) * (size_t)A);
        memset(ccs, 0, sizeof(int32_t) * (size_t)A);
        int64_t tot = 0;
        for (int a = 0; a < A; ++a) { ccl[lc[o + a]] += cnt[o + a]; ccs[sc[o + a]] += cnt[o + a]; tot += cnt[o + a]; }
        double ml[4], ms[4], ln[4];
        orc_mode_stats(ccl, A, r[0], r[1], r[2], pl, ml);
        orc_mode_stats(ccs, A, r[0], r[1], r[3], pl, ms);
        orc_length_stats(ccl, cv + o, A, ln);
        int32_t* oi = out_i + (size_t)l * 8;
        oi[0] = r[0]; oi[1] = r[1]; oi[2] = r[2]; oi[3] = r[3]; oi[4] = r[4];
        oi[5] = (int32_t)ml[3]; oi[6] = (int32_t)ms[3]; oi[7] = (int32_t)tot;
        double* of = out_f + (size_t)l * 10;
        of[0] = ln[0]; of[1] = ln[1]; of[2] = ln[2]; of[3] = ln[3];
        of[4] = ml[0]; of[5] = ms[0]; of[6] = ml[1]; of[7] = ms[1]; of[8] = ml[2]; of[9] = ms[2];
    }
}

/* dumpSTR ApplyCallFilters (dumpSTR.py:613-774) for the threshold filters min-DP / max-DP / min-Q
 * (CallFilterMinValue/MaxValue, filters.py:363-409) on a [L,S,2] batch.  thr_q is compared in float32.
 * counters: [4][S] int64 numcalls, minDP, maxDP, minQ;  totaldp [S] int64;  dpmiss [S] int64 */
void orc_call_filters_dpq(const int16_t* gt, const int32_t* dp, const float* q, int L, int S, double min_dp,
                          double max_dp, double min_q, int16_t* gt_out, uint32_t* mask, int64_t* counters,
                          int64_t* totaldp, int64_t* dpmiss) {
    float tq = (float)min_q;
    for (int l = 0; l < L; ++l)
        for (int s = 0; s < S; ++s) {
            size_t c = (size_t)l * S + s;
            int a0 = gt[c * 2], a1 = gt[c * 2 + 1];
            int called = !(a0 == -1 || a1 == -1);
            uint32_t m = 0;
            if ((double)dp[c] < min_dp) m |= 1u;
            if ((double)dp[c] > max_dp) m |= 2u;
            if (q[c] < tq) m |= 4u;
            if (called) {
                if (m & 1u) counters[1 * (size_t)S + s]++;
                if (m & 2u) counters[2 * (size_t)S + s]++;
                if (m & 4u) counters[3 * (size_t)S + s]++;
            } else m |= 0x80000000u;
            mask[c] = m;
            gt_out[c * 2] = (int16_t)a0; gt_out[c * 2 + 1] = (int16_t)a1;
            if (m == 0) {
                counters[s]++;
                if (dp[c] == INT32_MIN) dpmiss[s]++;
                else if (dp[c] > 0) totaldp[s] += dp[c];
            } else if (called) { gt_out[c * 2] = -1; gt_out[c * 2 + 1] = -1; }
        }
}

/* ---- any dumpSTR call-filter set -------------------------------------------------------------------------
 * filters.py:327-867 as (op, operands) records -- the numbering of include/trk.h's TRK_F_* so that a test can hand the
 * same spec to both sides -- evaluated one call at a time, then dumpSTR.py:613-774 (ApplyCallFilters):
 *   1 LT            CallFilterMinValue :363-367      value < thr (float32 planes compare in float32, as numpy does)
 *   2 GT            CallFilterMaxValue :405-409
 *   3 RATIO_GT      HipSTRCallFlankIndels / Stutter :444-449, :479-484   a / b in float64 > thr
 *   4 CALLED_LT     GangSTRCallExpansionProb{Hom,Het} :597-639, min supporting reads on a pre-parsed plane
 *   5 CALLED_SUM_LT GangSTRCallExpansionProbTotal :665-674   (a + a2 in the plane's dtype) < thr
 *   6 CALLED_EQ     GangSTRCallSpanOnly :686-697             a == b
 *   7 CALLED_SUM_EQ GangSTRCallSpanBoundOnly :711-722        a + a2 == b
 *   8 OUTSIDE_CI    GangSTRCallBadCI :739-757                REPCN_j outside [lo_j, hi_j]
 *   9 AD_SUPPORT_LT PopSTRCallRequireSupport :858-867        AD[s, gt[s, j]] < thr (numpy negative indexing)
 * planes are interleaved [L, S, ncol] int32 (missing INT_MIN) or float32 (missing nan).
 * mask bit k: filter k's output is not nan; bit 31: the sample is a no-call (:651).  counters [(1+nf), S]:
 * row 0 numcalls (:686-687), row 1+k sample_info[filter k] (:661).  err[0] != 0: negative DP on a PASS call (:698-706). */
typedef struct { int32_t op, plane_a, col_a, plane_b, col_b, col_a2; double thr; } orc_filter;
typedef struct { const void* data; int32_t is_f32; int32_t ncol; } orc_plane;

static inline int32_t pl_i(const orc_plane* p, size_t c, int col) { return ((const int32_t*)p->data)[c * (size_t)p->ncol + col]; }
static inline float pl_f(const orc_plane* p, size_t c, int col) { return ((const float*)p->data)[c * (size_t)p->ncol + col]; }
static inline double pl_d(const orc_plane* p, size_t c, int col) { return p->is_f32 ? (double)pl_f(p, c, col) : (double)pl_i(p, c, col); }

static int orc_eval(const orc_filter* f, const orc_plane* planes, size_t c, int called, const int16_t* g, int pl) {
    const orc_plane* a = planes + f->plane_a;
    switch (f->op) {
        case 4: if (!called) return 0; /* fall through */
        case 1: return a->is_f32 ? pl_f(a, c, f->col_a) < (float)f->thr : (double)pl_i(a, c, f->col_a) < f->thr;
        case 2: return a->is_f32 ? pl_f(a, c, f->col_a) > (float)f->thr : (double)pl_i(a, c, f->col_a) > f->thr;
        case 3: return pl_d(a, c, f->col_a) / pl_d(planes + f->plane_b, c, f->col_b) > f->thr;
        case 5:
            if (!called) return 0;
            if (a->is_f32) { volatile float s = pl_f(a, c, f->col_a) + pl_f(a, c, f->col_a2); return s < (float)f->thr; }
            return (double)((int64_t)pl_i(a, c, f->col_a) + (int64_t)pl_i(a, c, f->col_a2)) < f->thr;
        case 6: return called && pl_i(a, c, f->col_a) == pl_i(planes + f->plane_b, c, f->col_b);
        case 7: return called && (int64_t)pl_i(a, c, f->col_a) + (int64_t)pl_i(a, c, f->col_a2) ==
                                     (int64_t)pl_i(planes + f->plane_b, c, f->col_b);
        case 8: {
            if (!called) return 0;
            const orc_plane* b = planes + f->plane_b;
            for (int j = 0; j < a->ncol; ++j) {
                int32_t ml = pl_i(a, c, j);
                if (ml < pl_i(b, c, 2 * j) || pl_i(b, c, 2 * j + 1) < ml) return 1;
            }
            return 0;
        }
        case 9: {
            int hit = 0;
            for (int j = 0; j < pl; ++j) {
                int q = g[j];
                if (q < 0) q += a->ncol;
                if (q < 0 || q >= a->ncol) continue;
                hit |= (double)pl_i(a, c, q) < f->thr;
            }
            return hit;
        }
    }
    return 0;
}

void orc_call_filters(const int16_t* gt, int L, int S, int P, const uint8_t* locus_ploidy, const orc_plane* planes,
                      int n_planes, const orc_filter* filters, int nf, int dp_plane, int16_t* gt_out, uint32_t* mask,
                      int64_t* counters, int64_t* totaldp, int64_t* dpmiss, int32_t* err, int n_threads) {
    (void)n_planes;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        int64_t* cn = (int64_t*)calloc((size_t)(nf + 3) * (size_t)S, sizeof(int64_t));  /* thread-private, summed below */
        int64_t* td = cn + (size_t)(nf + 1) * S;
        int64_t* dm = td + S;
#pragma omp for schedule(dynamic, 8)
        for (int l = 0; l < L; ++l) {
            const int pl = locus_ploidy ? (locus_ploidy[l] < P ? locus_ploidy[l] : P) : P;
            for (int s = 0; s < S; ++s) {
                const size_t c = (size_t)l * S + s;
                const int16_t* g = gt + c * P;
                int called = 1;
                for (int j = 0; j < pl; ++j) called &= g[j] != -1;
                uint32_t m = 0;
                for (int k = 0; k < nf; ++k)
                    if (orc_eval(filters + k, planes, c, called, g, pl)) {
                        m |= 1u << k;
                        if (called) cn[(size_t)(1 + k) * S + s]++;
                    }
                if (!called) m |= 0x80000000u;
                if (mask) mask[c] = m;
                int filtered = 0;
                if (m == 0) {
                    cn[s]++;
                    if (dp_plane >= 0) {
                        const orc_plane* d = planes + dp_plane;
                        if (d->is_f32) {   /* Float depth (ExpansionHunter LC): not summed here */
                        } else {
                            int32_t v = pl_i(d, c, 0);
                            if (v == INT32_MIN) dm[s]++;
                            else if (v < 0) { if (err) err[0] = 1; }
                            else td[s] += v;
                        }
                    }
                } else if (called) filtered = 1;
                if (gt_out) {
                    int16_t* o = gt_out + c * P;
                    for (int j = 0; j < P; ++j) o[j] = (filtered && j < pl) ? (int16_t)-1 : g[j];
                }
            }
        }
#pragma omp critical
        {
            for (size_t i = 0; i < (size_t)(nf + 1) * S; ++i) counters[i] += cn[i];
            for (int s = 0; s < S; ++s) { totaldp[s] += td[s]; dpmiss[s] += dm[s]; }
        }
        free(cn);
    }
}

/* ======================================================================================
 * associaTR linear-regression scan, one locus at a time, restated from
 *   associaTR/load_and_filter_genotypes.py:157-259  (called samples, length allele frequencies rounded to two
 *                                                    decimals, the non-major-allele filter)
 *   associaTR/associaTR.py:246-291                  (summed length standardised over the tested samples, OLS of the
 *                                                    outcome on [genotype, 1, covariates], p / coefficient / se / R^2)
 * and from statsmodels' published OLS.fit() for a full-rank design (normal equations here, in long double; the
 * numpy restatement oracle/associatr_oracle.py does the SVD pseudo-inverse -- tests/test_oracle_c.py pins this
 * function to it on random loci and on the golden cases' designs).  Diploid tensors.
 * ====================================================================================== */
#include <stdio.h>

/* Python's round(x, 2) (correctly rounded decimal, half to even on the exact binary value) */
static double orc_round2(double x) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", x);
    return strtod(buf, NULL);
}

/* numpy's add.reduce over a contiguous float64 array of n <= 128 elements (pairwise_sum, numpy/core/src/umath/
 * loops_utils.h.src): a plain loop below 8 elements, eight accumulators above */
static double orc_np_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;           /* (numpy starts from -0.0; the sign of an empty or all-zero sum does not matter here) */
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* regularised incomplete beta I_x(a, b) by the continued fraction (modified Lentz), for the Student-t tail */
static double orc_betacf(double a, double b, double x) {
    const double tiny = 1e-300;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; ++m) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}
/* I_x(a, b) with xc = 1 - x handed in separately (either may be tiny; neither is formed by subtraction here) */
static double orc_betainc(double a, double b, double x, double xc) {
    if (x <= 0.0) return 0.0;
    if (xc <= 0.0) return 1.0;
    /* (long double log-gammas: at df ~ 1e5 the three terms are ~1e6 each and their difference is wanted to 1e-12) */
    double lbt = (double)(lgammal((long double)a + b) - lgammal((long double)a) - lgammal((long double)b)) + a * log(x) + b * log(xc);
    if (x < (a + 1.0) / (a + b + 2.0)) return exp(lbt) * orc_betacf(a, b, x) / a;
    return 1.0 - exp(lbt) * orc_betacf(b, a, xc) / b;
}
/* 2 * scipy.stats.t.sf(|t|, df) = I_{df / (df + t^2)}(df / 2, 1 / 2) */
double orc_t_two_sided(double t, double df) {
    if (t != t || df != df) return NAN;
    const double t2 = t * t;
    const double x = df / (df + t2), xc = t2 / (df + t2);
    /* |t| < 1: the p-value is of order 1 and x rounds towards 1 -- the complementary form keeps the digits */
    if (t2 < 1.0) return 1.0 - orc_betainc(0.5, 0.5 * df, xc, x);
    return orc_betainc(0.5 * df, 0.5, x, xc);
}

/* One locus.  x: design [S][M] in sample order, column 0 reserved for the genotype, column 1 the intercept, the rest
 * standardised covariates; y: outcome [S]; sample_in: uint8[S] or NULL.  out_i: {n_tested, status}; status 0 tested,
 * 1 no called samples, 2 only one called allele, 3 non-major allele count below the cutoff, 4 n covars >= n samples,
 * 5 degenerate (constant genotype / singular design: not compared).  out_f: {p, coef_std, se_std, R^2}. */
void orc_assoc_locus(const int16_t* gt, int S, int A, const double* alen, const uint8_t* sample_in, const double* x,
                     int M, const double* y, double non_major_cutoff, int32_t* out_i, double* out_f) {
    out_f[0] = out_f[1] = out_f[2] = out_f[3] = NAN;
    /* allele frequencies by rounded length, ascending */
    double* rl = (double*)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1));
    int* cls = (int*)malloc(sizeof(int) * (size_t)(A > 0 ? A : 1));
    double* ukey = (double*)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1));
    int nu = 0;
    for (int a = 0; a < A; ++a) rl[a] = orc_round2(alen[a]);
    for (int a = 0; a < A; ++a) {           /* distinct rounded lengths, ascending */
        int found = 0;
        for (int u = 0; u < nu; ++u) found |= ukey[u] == rl[a];
        if (!found) ukey[nu++] = rl[a];
    }
    for (int i = 1; i < nu; ++i) {          /* insertion sort */
        double k = ukey[i]; int j = i - 1;
        while (j >= 0 && ukey[j] > k) { ukey[j + 1] = ukey[j]; --j; }
        ukey[j + 1] = k;
    }
    for (int a = 0; a < A; ++a) for (int u = 0; u < nu; ++u) if (ukey[u] == rl[a]) cls[a] = u;
    /* counts by EXACT length first (np.unique over the called alleles' lengths), then merged by rounded key in
     * ascending order (clean_len_alleles): frequency of a rounded class = sum of c / total over its exact lengths in
     * ascending exact-length order */
    int64_t* cnt = (int64_t*)calloc((size_t)(A > 0 ? A : 1), sizeof(int64_t));
    int n = 0;
    long double sg = 0.0L, sgg = 0.0L;
    double* g = (double*)malloc(sizeof(double) * (size_t)(S > 0 ? S : 1));
    int* idx = (int*)malloc(sizeof(int) * (size_t)(S > 0 ? S : 1));
    int64_t total = 0;
    for (int s = 0; s < S; ++s) {
        int a0 = gt[2 * (size_t)s], a1 = gt[2 * (size_t)s + 1];
        if (a0 == -1 || a1 == -1) continue;
        if (sample_in && !sample_in[s]) continue;
        double v = 0.0;
        int ok = 1;
        for (int j = 0; j < 2; ++j) {
            int a = j ? a1 : a0;
            if (a == -2) v += -2.0;                     /* lut = [*lengths, -2, -1] (tr_harmonizer.py:1239) */
            else if (a >= 0 && a < A) { v += alen[a]; cnt[a]++; total++; }
            else ok = 0;
        }
        if (!ok) continue;
        g[n] = v; idx[n] = s; ++n;
        sg += v;
    }
    out_i[0] = n;
    /* distinct exact lengths among the counted alleles, ascending; their frequencies merged by rounded class */
    double* fr = (double*)calloc((size_t)(nu > 0 ? nu : 1), sizeof(double));
    int* seen = (int*)calloc((size_t)(nu > 0 ? nu : 1), sizeof(int));
    {
        /* order allele indices by exact length (stable), merge equal exact lengths first */
        int* ord = (int*)malloc(sizeof(int) * (size_t)(A > 0 ? A : 1));
        for (int a = 0; a < A; ++a) ord[a] = a;
        for (int i = 1; i < A; ++i) {
            int k = ord[i], j = i - 1;
            while (j >= 0 && alen[ord[j]] > alen[k]) { ord[j + 1] = ord[j]; --j; }
            ord[j + 1] = k;
        }
        int i = 0;
        while (i < A) {
            int j = i;
            int64_t c = 0;
            while (j < A && alen[ord[j]] == alen[ord[i]]) c += cnt[ord[j++]];
            if (c > 0) {
                int u = cls[ord[i]];
                double f = (double)c / (double)total;
                if (!seen[u]) { fr[u] = f; seen[u] = 1; } else fr[u] += f;
            }
            i = j;
        }
        free(ord);
    }
    int nfr = 0;
    double* af = (double*)malloc(sizeof(double) * (size_t)(nu > 0 ? nu : 1));
    for (int u = 0; u < nu; ++u) if (seen[u]) af[nfr++] = fr[u];
    int status = 0;
    if (nfr == 0) status = 1;
    else if (nfr == 1) status = 2;
    else {
        int am = 0;
        for (int u = 1; u < nfr; ++u) if (af[u] > af[am]) am = u;       /* np.argmax: first maximum */
        for (int u = am; u + 1 < nfr; ++u) af[u] = af[u + 1];
        if (orc_np_sum(af, nfr - 1) * n * 2 < non_major_cutoff) status = 3;
    }
    if (!status && M >= n) status = 4;
    if (!status) {
        const long double mean = sg / n;
        for (int i = 0; i < n; ++i) { long double d = g[i] - mean; sgg += d * d; }
        const long double sd = sqrtl(sgg / n);
        if (!(sd > 0.0L)) status = 5;
        else {
            /* normal equations of [g_std, x_1..x_{M-1}] in long double, Gauss-Jordan with partial pivoting */
            long double* N = (long double*)calloc((size_t)M * (M + 1), sizeof(long double));
            long double yy = 0.0L, ysum = 0.0L;
            long double* row = (long double*)malloc(sizeof(long double) * (size_t)M);
            for (int i = 0; i < n; ++i) {
                const int s = idx[i];
                row[0] = (g[i] - mean) / sd;
                for (int k = 1; k < M; ++k) row[k] = x[(size_t)s * M + k];
                for (int a = 0; a < M; ++a) {
                    for (int b = 0; b < M; ++b) N[a * (M + 1) + b] += row[a] * row[b];
                    N[a * (M + 1) + M] += row[a] * y[s];
                }
                yy += (long double)y[s] * y[s];
                ysum += y[s];
            }
            /* inverse of the normal matrix alongside: solve for beta and the (0,0) entry of the inverse */
            long double* W = (long double*)calloc((size_t)M * 2 * M, sizeof(long double));
            for (int a = 0; a < M; ++a) {
                for (int b = 0; b < M; ++b) W[a * 2 * M + b] = N[a * (M + 1) + b];
                W[a * 2 * M + M + a] = 1.0L;
            }
            int singular = 0;
            for (int c = 0; c < M && !singular; ++c) {
                int piv = c;
                for (int r = c + 1; r < M; ++r) if (fabsl(W[r * 2 * M + c]) > fabsl(W[piv * 2 * M + c])) piv = r;
                if (fabsl(W[piv * 2 * M + c]) < 1e-12L * n) { singular = 1; break; }
                if (piv != c) for (int k = 0; k < 2 * M; ++k) { long double t = W[c * 2 * M + k]; W[c * 2 * M + k] = W[piv * 2 * M + k]; W[piv * 2 * M + k] = t; }
                long double pv = W[c * 2 * M + c];
                for (int k = 0; k < 2 * M; ++k) W[c * 2 * M + k] /= pv;
                for (int r = 0; r < M; ++r) if (r != c) {
                    long double f = W[r * 2 * M + c];
                    if (f != 0.0L) for (int k = 0; k < 2 * M; ++k) W[r * 2 * M + k] -= f * W[c * 2 * M + k];
                }
            }
            if (singular) status = 5;
            else {
                long double* beta = (long double*)calloc((size_t)M, sizeof(long double));
                for (int a = 0; a < M; ++a) for (int b = 0; b < M; ++b) beta[a] += W[a * 2 * M + M + b] * N[b * (M + 1) + M];
                /* ssr from the residuals (second pass), as the reference does */
                long double ssr = 0.0L;
                for (int i = 0; i < n; ++i) {
                    const int s = idx[i];
                    long double fit = beta[0] * ((g[i] - mean) / sd);
                    for (int k = 1; k < M; ++k) fit += beta[k] * x[(size_t)s * M + k];
                    long double r = y[s] - fit;
                    ssr += r * r;
                }
                const long double df = (long double)n - M;
                const long double scale = ssr / df;
                const long double se = sqrtl(W[0 * 2 * M + M + 0] * scale);
                const long double ym = ysum / n;
                const long double sst = yy - n * ym * ym;
                out_f[0] = orc_t_two_sided((double)(beta[0] / se), (double)df);
                out_f[1] = (double)beta[0];
                out_f[2] = (double)se;
                out_f[3] = (double)(1.0L - ssr / sst);
                free(beta);
            }
            free(W); free(row); free(N);
        }
    }
    out_i[1] = status;
    free(af); free(seen); free(fr); free(idx); free(g); free(cnt); free(ukey); free(cls); free(rl);
}

void orc_assoc_scan_mt(const int16_t* gt, int L, int S, const int32_t* off, const double* alen, const uint8_t* sample_in,
                       const double* x, int M, const double* y, double non_major_cutoff, int32_t* out_i, double* out_f,
                       int n_threads) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int l = 0; l < L; ++l)
        orc_assoc_locus(gt + (size_t)l * S * 2, S, off[l + 1] - off[l], alen + off[l], sample_in, x, M, y, non_major_cutoff,
                        out_i + 2 * (size_t)l, out_f + 4 * (size_t)l);
}
