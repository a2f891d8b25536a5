// trk_binom.h -- float64 binomial pmf / tails / two-sided exact test, usable on
// host and device.  Replaces the third-party call at the end of the reference's
// HWE statistic (trtools/utils/utils.py:334-338):
//     scipy.stats.binomtest(num_hom, n=total_samples, p=exp_hom_frac).pvalue
// scipy (1.15.3, _binomtest.py, two-sided branch) does:
//     d = pmf(k); rerr = 1 + 1e-7
//     k == p*n            -> 1
//     k <  p*n : ix = bsearch(-pmf, -d*rerr, ceil(p*n), n)
//                y  = n - ix + (d*rerr == pmf(ix));  pval = cdf(k) + sf(n-y)
//     k >  p*n : ix = bsearch(pmf, d*rerr, 0, floor(p*n))
//                y  = ix + 1;                        pval = cdf(y-1) + sf(k-1)
//     min(1, pval)
// with `_binary_search_for_binom_tst` returning mid on equality, else lo or lo-1
// by a final `a(lo) <= d` test.  The search bounds are floats in scipy (np.ceil /
// np.floor of p*n); they hold integral values, so int64 reproduces them.
//
// pmf: Loader's saddle-point evaluation (C. Loader, "Fast and accurate
// computation of binomial probabilities", 2000) -- relative error ~1e-15, the
// same class of accuracy as the boost implementation behind scipy's binom.pmf.
// Tails: every tail the test needs lies on the far side of the mean from the
// mode, so the terms decay monotonically away from the starting point and are
// summed directly with the pmf ratio recurrence until they no longer contribute.
#ifndef TRK_BINOM_H
#define TRK_BINOM_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TRK_HD __host__ __device__
#else
#define TRK_HD
#endif
// On the device a test is a handful of pmf evaluations (~400 instructions each).  The evaluations every test makes
// are inlined at their few call sites; the rare ones (fallbacks, edge shapes) share ONE out-of-line copy, so that the
// kernel neither pays the call sequence (register save / restore through scratch) on its hot path nor grows past the
// instruction cache.
#if defined(__HIP_DEVICE_COMPILE__)
#define TRK_HOT __attribute__((always_inline))
#define TRK_COLD __attribute__((noinline))
#else
#define TRK_HOT
#define TRK_COLD
#endif

namespace trkmath {

// steps the boundary search walks from its guess before it gives up (a step costs a hundredth of a pmf evaluation,
// the bisection behind it a dozen evaluations)
#define TRK_BINOM_WALK 192

// Where the pmf on the far side of the mean falls to pmf(k): k mirrored at the mean, moved by the skew.  With
// z = (k - np) / sigma, ln pmf ~ -z^2/2 + g (z^3 - 3z)/6, g = (1 - 2p)/sigma, and equal values at -z1 and z1 + delta
// give delta = g z1^2 / 3, i.e. (1 - 2p) z^2 / 3 counts -- 13 at z = 10, p = 0.3; more than a hundred at z = 30,
// which call sets far from equilibrium do reach.  Only the cost of the search depends on the guess.
TRK_HD inline int64_t binom_mirror_guess(double kd, double pn, double p) {
    const double q = 1.0 - p;
    const double var = pn * q;
    const double dz = kd - pn;
    const double skew = var > 0.0 ? (1.0 - 2.0 * p) * (dz * dz) / (3.0 * var) : 0.0;
    const double g = 2.0 * pn - kd + skew + 0.5;
    return (int64_t)floor(g < -1.0 ? -1.0 : (g > 4.0e18 ? 4.0e18 : g));
}

// 1/x: on the device v_rcp_f64 refined by one Newton step (relative error
// ~1e-16, an order of magnitude cheaper than the IEEE division sequence; the
// statistic's parity bar is 1e-9); plain division on the host.
TRK_HD TRK_HOT inline double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
#else
    return 1.0 / x;
#endif
}

// Stirling series error  ln(n!) - [ (n+1/2) ln n - n + ln sqrt(2 pi) ]
TRK_HD TRK_HOT inline double stirlerr(double n) {
    const double S0 = 0.083333333333333333333;        // 1/12
    const double S1 = 0.00277777777777777777778;      // 1/360
    const double S2 = 0.00079365079365079365079365;   // 1/1260
    const double S3 = 0.000595238095238095238095238;  // 1/1680
    const double S4 = 0.0008417508417508417508417508; // 1/1188
    if (n <= 15.0) {
        // exact values for integer n (only integers reach this function)
        switch ((int)n) {
            case 0: return 0.0;
            case 1: return 0.0810614667953272610700921;
            case 2: return 0.0413406959554092970354766;
            case 3: return 0.0276779256849983383570457;
            case 4: return 0.0207906721037650933647800;
            case 5: return 0.0166446911898211931390978;
            case 6: return 0.0138761288230707484359083;
            case 7: return 0.0118967099458917695276039;
            case 8: return 0.0104112652619720962021699;
            case 9: return 0.0092554621827127328548279;
            case 10: return 0.0083305634333628707927089;
            case 11: return 0.0075736754879518405902949;
            case 12: return 0.0069428401072095299179088;
            case 13: return 0.0064089941880042071431500;
            case 14: return 0.0059513701127588474956709;
            default: return 0.0055547335519628010525039;
        }
    }
    double nn = n * n;
    if (n > 500.0) return (S0 - S1 / nn) / n;
    if (n > 80.0) return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 35.0) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}

// deviance term  x ln(x/np) + np - x , accurate when x ~ np
TRK_HD TRK_HOT inline double bd0(double x, double np) {
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        double ej = 2.0 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; ++j) {
            ej *= v;
            double s1 = s + ej * fast_rcp((double)(2 * j + 1));
            if (s1 == s) return s1;
            s = s1;
        }
        return s;
    }
    return x * log(x / np) + np - x;
}

TRK_HD TRK_HOT inline double binom_pmf(int64_t ki, int64_t ni, double p) {
    if (ki < 0 || ki > ni) return 0.0;
    double k = (double)ki, n = (double)ni, q = 1.0 - p;
    if (p <= 0.0) return ki == 0 ? 1.0 : 0.0;
    if (q <= 0.0) return ki == ni ? 1.0 : 0.0;
    if (ki == 0) {
        if (ni == 0) return 1.0;
        double lc = (p < 0.1) ? -bd0(n, n * q) - n * p : n * log(q);
        return exp(lc);
    }
    if (ki == ni) {
        double lc = (q < 0.1) ? -bd0(n, n * p) - n * q : n * log(p);
        return exp(lc);
    }
    double lc = stirlerr(n) - stirlerr(k) - stirlerr(n - k) - bd0(k, n * p) - bd0(n - k, n * q);
    // ln(2 pi k (n-k)/n)
    double lf = 1.8378770664093455611265426 + log(k) + log1p(-k / n);
    return exp(lc - 0.5 * lf);
}

// the out-of-line copy for the rare call sites
TRK_HD TRK_COLD inline double binom_pmf_cold(int64_t k, int64_t n, double p) { return binom_pmf(k, n, p); }

// sum_{i=0..k} pmf(i); intended for k at or below the mean (terms shrink downwards)
TRK_HD inline double binom_lower_tail(int64_t k, int64_t n, double p) {
    if (k < 0) return 0.0;
    if (k >= n) return 1.0;
    double q = 1.0 - p;
    if (p <= 0.0) return 1.0;
    if (q <= 0.0) return 0.0;  // k < n
    double t = binom_pmf_cold(k, n, p);
    double sum = t;
    const double r = q / p;
    double di = (double)k, dd = (double)(n - k + 1);
    for (int64_t i = k; i > 0; --i) {
        double ratio = di * fast_rcp(dd) * r;
        t *= ratio;
        sum += t;
        if (ratio < 1.0 && t <= sum * 1e-18) break;
        di -= 1.0;
        dd += 1.0;
    }
    return sum;
}

// sum_{i=k+1..n} pmf(i) = sf(k); intended for k+1 at or above the mean
TRK_HD inline double binom_upper_tail(int64_t k, int64_t n, double p) {
    if (k >= n) return 0.0;
    if (k < 0) return 1.0;
    double q = 1.0 - p;
    if (p <= 0.0) return 0.0;
    if (q <= 0.0) return 1.0;  // k < n
    double t = binom_pmf_cold(k + 1, n, p);
    double sum = t;
    const double r = p / q;
    double dn = (double)(n - k - 1), dd = (double)(k + 2);
    for (int64_t i = k + 1; i < n; ++i) {
        double ratio = dn * fast_rcp(dd) * r;
        t *= ratio;
        sum += t;
        if (ratio < 1.0 && t <= sum * 1e-18) break;
        dn -= 1.0;
        dd += 1.0;
    }
    return sum;
}

// scipy _binary_search_for_binom_tst on a(x) = sign * pmf(x)
// (a LEAF function on the device: ONE inlined pmf evaluation serves the loop and the final test -- a call from here
// would cost a stack frame for the return address in every kernel that can reach the bisection)
TRK_HD TRK_COLD inline int64_t binom_bsearch(double sign, double d, int64_t lo, int64_t hi, int64_t n,
                                    double p) {
    for (;;) {
        const bool last = !(lo < hi);
        const int64_t mid = last ? lo : lo + (hi - lo) / 2;
        const double midval = sign * binom_pmf(mid, n, p);
        if (last) return midval <= d ? lo : lo - 1;
        if (midval < d) {
            lo = mid + 1;
        } else if (midval > d) {
            hi = mid - 1;
        } else {
            return mid;
        }
    }
}

// The index scipy's bisection returns -- the largest i in [lo, hi] with sign * pmf(i) <= d, lo - 1 if there is none
// (sign * pmf is increasing over the range: it lies on one side of the mode) -- found from a GUESS instead: one pmf
// evaluation at the guess, a walk along the pmf ratio recurrence to the crossing, and two direct evaluations that
// confirm it (pmf(ix) is returned: the caller compares it with d).  The bisection costs ~log2(range) + 1 evaluations
// of ~400 instructions each; the guess (k mirrored at the mean) is a few steps off.  Falls back to the bisection
// when the walk does not arrive.
// *pmf_next = pmf(ix + 1) when ix < hi: both values start a tail sum of the test, which therefore needs no further
// evaluation.
TRK_HD TRK_HOT inline int64_t binom_boundary(double sign, double d, int64_t lo, int64_t hi, int64_t n, double p, int64_t guess,
                                     double* pmf_ix, double* pmf_next) {
    if (lo > hi) {                       // (the bisection's final test alone)
        const int64_t ix = sign * binom_pmf_cold(lo, n, p) <= d ? lo : lo - 1;
        *pmf_ix = binom_pmf_cold(ix, n, p);
        *pmf_next = binom_pmf_cold(ix + 1, n, p);
        return ix;
    }
    const double q = 1.0 - p;
    int64_t i = guess < lo ? lo : (guess > hi ? hi : guess);
    double t = binom_pmf(i, n, p);
    const double up = p / q, down = q / p;   // pmf(i+1) = pmf(i) (n-i)/(i+1) p/q ; pmf(i-1) = pmf(i) i/(n-i+1) q/p
    bool ok = p > 0.0 && q > 0.0;
    if (ok) {
        int steps = 0;
        if (sign * t <= d) {             // inside: move right while the next one is inside too
            while (i < hi && steps < TRK_BINOM_WALK) {
                const double tn = t * ((double)(n - i) * fast_rcp((double)(i + 1)) * up);
                if (!(sign * tn <= d)) break;
                t = tn;
                ++i;
                ++steps;
            }
        } else {                         // outside: move left until inside (or past lo)
            while (i >= lo && steps < TRK_BINOM_WALK) {
                if (i == lo) { i = lo - 1; break; }
                t = t * ((double)i * fast_rcp((double)(n - i + 1)) * down);
                --i;
                ++steps;
                if (sign * t <= d) break;
            }
        }
        ok = steps < TRK_BINOM_WALK;
    }
    if (ok) {
        // confirm with the direct evaluation (the walk's values carry the recurrence's rounding): i inside, i + 1 not
        for (int fix = 0; fix < 4 && ok; ++fix) {
            const double ti = i >= lo ? binom_pmf(i, n, p) : 0.0;
            if (i >= lo && !(sign * ti <= d)) { --i; continue; }
            double tn_last = 0.0;
            if (i < hi) {
                tn_last = binom_pmf(i + 1, n, p);
                if (sign * tn_last <= d) { ++i; continue; }
            }
            *pmf_ix = i >= lo ? ti : binom_pmf_cold(i, n, p);
            *pmf_next = tn_last;   // (0 when ix == hi: no caller starts a tail there)
            return i;
        }
    }
    const int64_t ix = binom_bsearch(sign, d, lo, hi, n, p);
    *pmf_ix = binom_pmf_cold(ix, n, p);
    *pmf_next = binom_pmf_cold(ix + 1, n, p);
    return ix;
}

// sum_{i<=kl} pmf(i) + sum_{i>ku} pmf(i): both far tails advanced in ONE loop (two
// independent recurrences per iteration: twice the ILP, half the trip count, and --
// on the GPU -- one code path for every lane whatever side of the mean k lies on).
// t_l0 = pmf(kl), t_u0 = pmf(ku + 1) when the caller already holds them (nan: evaluated here).
TRK_HD inline double binom_two_tails(int64_t kl, int64_t ku, int64_t n, double p, double t_l0 = __builtin_nan(""),
                                     double t_u0 = __builtin_nan("")) {
    const double q = 1.0 - p;
    double sum_l = 0.0, sum_u = 0.0, t_l = 0.0, t_u = 0.0;
    bool run_l = false, run_u = false;
    // lower tail: edge cases of binom_lower_tail
    if (kl >= n) sum_l = 1.0;
    else if (kl >= 0) {
        if (p <= 0.0) sum_l = 1.0;
        else if (q > 0.0) { t_l = t_l0 == t_l0 ? t_l0 : binom_pmf_cold(kl, n, p); sum_l = t_l; run_l = kl > 0; }
    }
    // upper tail: edge cases of binom_upper_tail
    if (ku < 0) sum_u = 1.0;
    else if (ku < n) {
        if (q <= 0.0) sum_u = 1.0;
        else if (p > 0.0) { t_u = t_u0 == t_u0 ? t_u0 : binom_pmf_cold(ku + 1, n, p); sum_u = t_u; run_u = ku + 1 < n; }
    }
    const double r_l = (p > 0.0) ? q / p : 0.0;
    const double r_u = (q > 0.0) ? p / q : 0.0;
    double di = (double)kl, ddl = (double)(n - kl + 1);
    double dn = (double)(n - ku - 1), ddu = (double)(ku + 2);
    while (run_l | run_u) {
        if (run_l) {
            const double ratio = di * fast_rcp(ddl) * r_l;
            t_l *= ratio;
            sum_l += t_l;
            di -= 1.0;
            ddl += 1.0;
            run_l = !((ratio < 1.0 && t_l <= sum_l * 1e-18) || di <= 0.0);
        }
        if (run_u) {
            const double ratio = dn * fast_rcp(ddu) * r_u;
            t_u *= ratio;
            sum_u += t_u;
            dn -= 1.0;
            ddu += 1.0;
            run_u = !((ratio < 1.0 && t_u <= sum_u * 1e-18) || dn <= 0.0);
        }
    }
    return sum_l + sum_u;
}

// scipy.stats.binomtest(k, n, p, alternative='two-sided').pvalue  (n >= 1, 0 <= k <= n)
// One code path for k below and above the mean (the branch only selects the search
// range, the sign of the searched function and which two tails are summed).
TRK_HD inline double binomtest_two_sided(int64_t k, int64_t n, double p) {
    const double d = binom_pmf(k, n, p);
    const double rerr = 1.0 + 1e-7;
    const double pn = p * (double)n;
    const double kd = (double)k;
    if (kd == pn) return 1.0;
    // pmf(k) underflows (|z| beyond ~38): every term of both tails is at most pmf(k), the sums below come out 0.0 --
    // after a search over indices whose values are all 0, which no guess shortens
    if (d == 0.0) return 0.0;
    const bool below = kd < pn;
    const double sign = below ? -1.0 : 1.0;
    const int64_t lo = below ? (int64_t)ceil(pn) : 0;
    const int64_t hi = below ? n : (int64_t)floor(pn);
    // (the mirror image of k at the mean is where a symmetric pmf would cross; skew moves the crossing a few steps)
    double pmf_ix, pmf_next;
    const int64_t ix = binom_boundary(sign, sign * d * rerr, lo, hi, n, p, binom_mirror_guess(kd, pn, p), &pmf_ix,
                                      &pmf_next);
    // the four terms the two tails start from are pmf(k), pmf(ix) and pmf(ix + 1): all evaluated above
    int64_t kl, ku;
    double t_l0, t_u0;
    if (below) {
        const bool eq = d * rerr == pmf_ix;
        const int64_t y = n - ix + (eq ? 1 : 0);
        kl = k;          // cdf(k)
        ku = n - y;      // sf(n - y): starts at pmf(ix + 1 - eq)
        t_l0 = d;
        t_u0 = eq ? pmf_ix : pmf_next;
    } else {
        kl = ix;         // cdf(y - 1), y = ix + 1
        ku = k - 1;      // sf(k - 1)
        t_l0 = pmf_ix;
        t_u0 = d;
    }
    const double pval = binom_two_tails(kl, ku, n, p, t_l0, t_u0);
    return pval < 1.0 ? pval : 1.0;
}

#if defined(__HIPCC__)
__device__ TRK_COLD inline double binomtest_two_sided_cold(int64_t k, int64_t n, double p) { return binomtest_two_sided(k, n, p); }

// The same test by TWO neighbouring lanes (h = 0 / 1, both holding k, n, p): the pmf evaluations -- the cost of a
// test, ~400 dependent instructions each -- go two at a time (pmf(k) beside pmf(guess), pmf(i) beside pmf(i + 1)),
// and each lane sums one tail.  Every decision and every sum is the serial function's, term for term.
__device__ TRK_HOT inline double binomtest_two_sided_pair(int64_t k, int64_t n, double p, int h, bool* ok);
// the pair routine with the serial one behind it (out of line), for callers that are not short of registers
__device__ inline double binomtest_two_sided_pair_or_serial(int64_t k, int64_t n, double p, int h) {
    bool ok;
    const double pv = binomtest_two_sided_pair(k, n, p, h, &ok);
    return ok ? pv : binomtest_two_sided_cold(k, n, p);
}

// *ok = false (both lanes): a shape the pair does not handle -- p at 0 or 1, or the walk from the guess did not arrive
// -- and the caller runs the serial routine instead (k_hwe_test: through its overflow list, so that the hot kernel
// holds no call and no second copy of the test).
__device__ TRK_HOT inline double binomtest_two_sided_pair(int64_t k, int64_t n, double p, int h, bool* ok) {
    *ok = true;
    const double rerr = 1.0 + 1e-7;
    const double pn = p * (double)n;
    const double kd = (double)k;
    const double q = 1.0 - p;
    const bool below = kd < pn;
    const double sign = below ? -1.0 : 1.0;
    const int64_t lo = below ? (int64_t)ceil(pn) : 0;
    const int64_t hi = below ? n : (int64_t)floor(pn);
    if (kd == pn) return 1.0;
    if (lo > hi || !(p > 0.0) || !(q > 0.0)) { *ok = false; return 0.0; }   // (the pair agrees)
    const int64_t guess = binom_mirror_guess(kd, pn, p);
    int64_t i = guess < lo ? lo : (guess > hi ? hi : guess);
    const double v = binom_pmf(h ? i : k, n, p);
    const double vo = __shfl_xor(v, 1);
    const double d = h ? vo : v;       // pmf(k)
    double t = h ? v : vo;             // pmf(i)
    if (d == 0.0) return 0.0;          // (as the serial routine)
    const double up = p / q, down = q / p;
    const double thr = sign * d * rerr;
    int steps = 0;
    if (sign * t <= thr) {
        while (i < hi && steps < TRK_BINOM_WALK) {
            const double tn = t * ((double)(n - i) * fast_rcp((double)(i + 1)) * up);
            if (!(sign * tn <= thr)) break;
            t = tn;
            ++i;
            ++steps;
        }
    } else {
        while (i >= lo && steps < TRK_BINOM_WALK) {
            if (i == lo) { i = lo - 1; break; }
            t = t * ((double)i * fast_rcp((double)(n - i + 1)) * down);
            --i;
            ++steps;
            if (sign * t <= thr) break;
        }
    }
    bool found = false;
    double pmf_ix = 0.0, pmf_next = 0.0;
    for (int fix = 0; fix < 4 && steps < TRK_BINOM_WALK; ++fix) {
        // (i == lo - 1, nothing inside the range -- k is the mode: pmf(lo - 1) is still wanted, as pmf_ix)
        const bool need = h ? i < hi : i >= lo - 1;
        const double e = need ? binom_pmf(i + h, n, p) : 0.0;
        const double eo = __shfl_xor(e, 1);
        const double ti = h ? eo : e, tn = h ? e : eo;
        if (i >= lo && !(sign * ti <= thr)) { --i; continue; }
        if (i < hi && sign * tn <= thr) { ++i; continue; }
        pmf_ix = ti;
        pmf_next = tn;
        found = true;
        break;
    }
    if (!found) { *ok = false; return 0.0; }   // the walk did not arrive: the serial path (bisection)
    // lane 0: sum_{j <= kl} pmf(j) downwards from t_l0; lane 1: sum_{j > ku} pmf(j) upwards from t_u0
    int64_t kl, ku;
    double t_l0, t_u0;
    if (below) {
        const bool eq = d * rerr == pmf_ix;
        kl = k;
        ku = i - (eq ? 1 : 0);
        t_l0 = d;
        t_u0 = eq ? pmf_ix : pmf_next;
    } else {
        kl = i;
        ku = k - 1;
        t_l0 = pmf_ix;
        t_u0 = d;
    }
    double sum = 0.0, tt = 0.0, num, den;
    bool run = false;
    if (h == 0) {
        if (kl >= n) sum = 1.0;
        else if (kl >= 0) { tt = t_l0; sum = tt; run = kl > 0; }
        num = (double)kl;
        den = (double)(n - kl + 1);
    } else {
        if (ku < 0) sum = 1.0;
        else if (ku < n) { tt = t_u0; sum = tt; run = ku + 1 < n; }
        num = (double)(n - ku - 1);
        den = (double)(ku + 2);
    }
    const double r = h ? up : down;
    while (run) {
        const double ratio = num * fast_rcp(den) * r;
        tt *= ratio;
        sum += tt;
        num -= 1.0;
        den += 1.0;
        run = !((ratio < 1.0 && tt <= sum * 1e-18) || num <= 0.0);
    }
    const double so = __shfl_xor(sum, 1);
    const double pval = h ? so + sum : sum + so;   // sum_l + sum_u
    return pval < 1.0 ? pval : 1.0;
}
#endif

}  // namespace trkmath
#endif
