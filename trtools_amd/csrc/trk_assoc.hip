// trk_assoc.hip -- associaTR linear-regression scan for gfx950.
//
// Reference loop (one Python iteration + one statsmodels OLS fit per locus):
//   trtools/associaTR/load_and_filter_genotypes.py:157-259, trtools/associaTR/associaTR.py:246-291.
// Here: ONE pass over the genotype tensor produces, per locus, the cross-products the
// regression needs (float64) and the allele histogram of the tested samples; a second,
// thread-per-locus kernel applies the locus filters, solves the normal equations by
// Cholesky and evaluates the Student-t tail.
//
//   k_assoc_gram      Gram matrix of [outcome, covariates, 1] over the regression sample set
//                     (once per call; per-locus Grams are this minus the rows of samples whose
//                     call is missing at the locus).
//   k_assoc_scan<MV>  HBM-bound streaming kernel.  Workgroup = 16 waves sharing one chunk of
//                     the sample vectors in LDS (float64, masked); each WAVE streams the
//                     genotype row segment of one locus (16 B nontemporal loads per lane), looks
//                     the two allele lengths up in a per-wave LDS table, and accumulates
//                         n, sum g, sum g^2, sum g*vec_k        (g = summed length, pivoted)
//                     in registers, the allele histogram in K bank-private LDS copies (as
//                     k_locus_count_v2), and -- on the rare lanes whose call is missing -- the
//                     Gram correction, one matrix entry per lane.  4 B of HBM traffic per call.
//   k_assoc_scan_any  any ploidy / alignment / allele count (correctness path).
//   k_assoc_finalize  thread per locus: filters (numpy's pairwise summation order restated so
//                     that the cutoff comparison is bit-exact), Cholesky with the genotype
//                     column last (beta = z_p / L_pp, var = scale / L_pp^2, ssr = y'y - |z|^2),
//                     two-sided t tail (trk_student.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/trk.h"
#include "../../include/trk_test.h"
#include "trk_internal.h"
#include "trk_student.h"

namespace {

constexpr int WAVE = 64;
constexpr int AS_WAVES = 16;                 // waves per workgroup of the scan kernel
constexpr int AS_THREADS = WAVE * AS_WAVES;  // 1024
constexpr int AS_MAXV = TRK_ASSOC_MAX_VEC;   // 31
constexpr int AS_MAXNC = (AS_MAXV + 1) * (AS_MAXV + 2) / 2;  // 528 Gram entries of [vec..., 1]
constexpr int AS_E = (AS_MAXNC + WAVE - 1) / WAVE;           // Gram entries per lane, generic kernel (9)
constexpr int FIN_T = 64;                    // finaliser threads per block (fewer when the normal matrix is large)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// columns of the per-(chunk, locus) partial record (float64):
//   0 n | 1 sum g | 2 sum g^2 | 3..3+M-1 sum g*vec_k | 3+M.. Gram correction (NC) | last: n_bad
struct AssocArgs {
    trk_batch b;
    const double* vec;         // [M, S]
    const uint8_t* sample_in;  // [S] or null
    const double* allele_len;  // [sumA]
    double* partial;           // [nchunks, L, NS]
    int32_t* allele_count;     // [sumA]
    int M, NS, NC;
    int chunk;                 // samples per chunk (multiple of 256)
    int nchunks;
    int loci_per_wg;           // 0: persistent workgroups, waves take loci from work_counter
    int* work_counter;         // [nchunks] next locus of each sample chunk (zeroed by the launcher)
    int wave_bytes;            // LDS bytes of one wave's private area (LUT + histogram)
    int kshift;
    int n_cu;
    uint8_t pa[AS_MAXNC], pb[AS_MAXNC];  // Gram entry e = row pa[e] x row pb[e]; row M = ones
    // MFMA kernel: the missing calls of the regression set, one bit per sample, for k_assoc_gram_miss
    unsigned long long* missbits;   // [L][nsteps][4]: word (step, j), bit `lane` <-> sample 256 step + 4 lane + j
    double* vect;                   // [S][16 RT] sample-major copy of the vector block (+ the row of ones, zero rows)
    int nsteps;
};

// -------------------------------------------------------------------------------------------
// Gram matrix of the sample vectors (+ ones) over the regression set: one thread per entry
// -------------------------------------------------------------------------------------------
template <bool WIDE>   // WIDE: more entries than the pa / pb tables hold (M > 31): the pair is derived from e
__global__ __launch_bounds__(256) void k_assoc_gram(AssocArgs a, double* __restrict__ full) {
    const int e = blockIdx.x;
    const int S = a.b.n_samples;
    int ra, rb;
    if (WIDE) {
        int r = 0, q = e;
        while (q >= a.M + 1 - r) {
            q -= a.M + 1 - r;
            ++r;
        }
        ra = r;
        rb = r + q;
    } else {
        ra = a.pa[e];
        rb = a.pb[e];
    }
    double acc = 0.0;
    for (int s = threadIdx.x; s < S; s += 256) {
        if (a.sample_in && !a.sample_in[s]) continue;
        const double x = ra == a.M ? 1.0 : a.vec[(size_t)ra * S + s];
        const double y = rb == a.M ? 1.0 : a.vec[(size_t)rb * S + s];
        acc += x * y;
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) full[e] = red[0];
}

// -------------------------------------------------------------------------------------------
// streaming scan, diploid fast path
// -------------------------------------------------------------------------------------------
template <int MV>
struct Acc {
    static constexpr int E = ((MV + 1) * (MV + 2) / 2 + WAVE - 1) / WAVE;  // Gram entries per lane
    double sg = 0.0, sgg = 0.0;
    double sgv[MV];
    double corr[E];
};

struct ScanCtx {
    const double* lut;   // LDS, by BIN: 0 '-2', 1 '-1' (NaN), 2..A+1 alleles, A+2 out of range
    uint32_t* hist;      // LDS, (A+3) bins x K copies
    const double* vec;   // LDS [M][Sr]
    int Sr, M, kshift, kslot;   // Sr: row stride of vec (chunk, +1 when lanes read different rows)
    uint32_t amax2;
};

// One chunk = 4 consecutive samples of one lane.  `mk`: their regression-set bytes (MASK only).
// Returns the 4-bit set of samples that are in the regression set but not called here.
template <int MV, bool MASK, bool TAIL>
__device__ __forceinline__ uint32_t scan_chunk(const ScanCtx& x, const u32x4 v, uint32_t mk, int s0, bool live,
                                               Acc<MV>& acc) {
    // few vectors: fetch the four samples of every vector up front (two 16-byte LDS reads each);
    // many vectors: read inside the loop, the accumulators need the registers
    constexpr bool PRE = MV <= 2;
    double y[PRE ? MV : 1][4];
    if (PRE) {
#pragma unroll
        for (int k = 0; k < MV; ++k) {
                if (!TAIL || live) {
                    const double2 a0 = *reinterpret_cast<const double2*>(&x.vec[(size_t)k * x.Sr + s0]);
                    const double2 a1 = *reinterpret_cast<const double2*>(&x.vec[(size_t)k * x.Sr + s0 + 2]);
                    y[k][0] = a0.x; y[k][1] = a0.y; y[k][2] = a1.x; y[k][3] = a1.y;
                } else {
                    y[k][0] = y[k][1] = y[k][2] = y[k][3] = 0.0;
                }
            }
    }
    uint32_t rare = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = v[j];
        u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
        u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, x.amax2));
        const uint32_t t = __builtin_bit_cast(uint32_t, t2);
        const uint32_t lo = t & 0xffffu, hi = t >> 16;
        double g = x.lut[lo] + x.lut[hi];   // NaN when either haplotype is '-1'
        const bool called = g == g;
        bool ok = called;
        if (MASK) {
            const bool in = (mk >> (8 * j)) & 1u;
            ok = called & in;
            rare |= (uint32_t)(in & !called) << j;
        } else {
            rare |= (uint32_t)(!called) << j;
        }
        if (TAIL) {
            ok &= live;
        }
        g = ok ? g : 0.0;
        acc.sg += g;
        acc.sgg = __builtin_fma(g, g, acc.sgg);
#pragma unroll
        for (int k = 0; k < MV; ++k) {
            // few vectors: M == MV.  Many: the (uniform) guard also keeps the compiler from
            // hoisting MV x 4 x U LDS reads to the top of the loop, which spills
            if (!PRE && k >= x.M) continue;
            const double yv = PRE ? y[k][j] : ((!TAIL || live) ? x.vec[(size_t)k * x.Sr + s0 + j] : 0.0);
            acc.sgv[k] = __builtin_fma(g, yv, acc.sgv[k]);
        }
        // calls that are not tested are counted in bin 1 (2 per call): n = calls - bin1 / 2
        if (!TAIL || live) {
            atomicAdd(&x.hist[((ok ? lo : 1u) << x.kshift) + x.kslot], 1u);
            atomicAdd(&x.hist[((ok ? hi : 1u) << x.kshift) + x.kslot], 1u);
        }
    }
    if (TAIL && !live) rare = 0;
    return rare;
}

// Samples of the regression set whose call is missing at this locus leave its Gram matrix
// (a few % of the calls).  Doing that work where the call is met serialises the wave on LDS
// latency (measured: 5.4 of 6.6 ms), so the main loop only QUEUES them -- one record per
// 4-sample chunk that holds any: (chunk << 4 | 4-bit set), appended with ballot/mbcnt, order
// deterministic -- and the queue is drained in bulk, when nearly full and at the end of the row.
constexpr int AS_QCAP = 512;  // records per wave (uint16: 12-bit chunk index, 4-bit set)

__device__ __forceinline__ void queue_push(uint16_t* q, int& qlen, uint32_t rare, int c) {
    const uint64_t mm = __ballot(rare != 0);
    if (mm) {
        const int pos = qlen + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
        if (rare) q[pos] = (uint16_t)(((uint32_t)c << 4) | rare);
        qlen += __popcll(mm);
    }
}

// few vectors: lane = queued sample; every lane keeps the whole (small) Gram triangle of its
// samples in registers -- rows 0..MV-1 the vectors (zero beyond M), row MV the ones
template <int MV>
__device__ __forceinline__ void drain_by_sample(const ScanCtx& x, const uint16_t* q, int qlen, int lane, double* cs) {
    wave_fence();
    for (int base = 0; base < qlen; base += WAVE) {
        const uint32_t rec = base + lane < qlen ? q[base + lane] : 0u;
        uint32_t bits = rec & 15u;
        const int c = (int)(rec >> 4);
        while (bits) {
            const int j = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int s = c * 4 + j;
            double z[MV + 1];
#pragma unroll
            for (int k = 0; k < MV; ++k) z[k] = x.vec[(size_t)k * x.Sr + s];
            z[MV] = 1.0;
            int e = 0;
#pragma unroll
            for (int r = 0; r <= MV; ++r)
#pragma unroll
                for (int cc = r; cc <= MV; ++cc) cs[e++] += z[r] * z[cc];
        }
    }
    wave_fence();
}

// many vectors: lane = Gram entry (row pa x row pb, row M = ones); the queued samples are
// walked four at a time so that eight LDS reads are in flight
template <int MV>
__device__ __forceinline__ void drain_by_entry(const ScanCtx& x, const uint16_t* q, int qlen, int lane,
                                               const int* pa, const int* pb, double* corr) {
    constexpr int E = Acc<MV>::E;
    wave_fence();
    auto one = [&](int s) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (pa[e] < 0) continue;
            const double xa = pa[e] == x.M ? 1.0 : x.vec[(size_t)pa[e] * x.Sr + s];
            const double xb = pb[e] == x.M ? 1.0 : x.vec[(size_t)pb[e] * x.Sr + s];
            corr[e] += xa * xb;
        }
    };
    for (int base = 0; base < qlen; base += WAVE) {
        const int cnt = min(WAVE, qlen - base);
        const uint32_t rec = lane < cnt ? q[base + lane] : 0u;
        int i = 0;
        for (; i + 4 <= cnt; i += 4) {  // lowest sample of four records at a time
            int s4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rec, i + k);
                s4[k] = (int)(r >> 4) * 4 + __ffs((int)(r & 15u)) - 1;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (pa[e] < 0) continue;
                double xa[4], xb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xa[k] = pa[e] == x.M ? 1.0 : x.vec[(size_t)pa[e] * x.Sr + s4[k]];
                    xb[k] = pb[e] == x.M ? 1.0 : x.vec[(size_t)pb[e] * x.Sr + s4[k]];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) corr[e] += xa[k] * xb[k];
            }
        }
        for (; i < cnt; ++i) {
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rec, i);
            one((int)(r >> 4) * 4 + __ffs((int)(r & 15u)) - 1);
        }
        // records that hold more than one sample (under 1 % of the chunks): the rest, one at a time
        const uint32_t b0 = rec & 15u;
        uint32_t rest = b0 & (b0 - 1u);
        uint64_t mm = __ballot(rest != 0);
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rec, src);
            uint32_t bits = (r & 15u) & ((r & 15u) - 1u);
            while (bits) {
                const int j = __ffs((int)bits) - 1;
                bits &= bits - 1;
                one((int)(r >> 4) * 4 + j);
            }
        }
    }
    wave_fence();
}

template <int MV, bool MASK>
__global__ __launch_bounds__(AS_THREADS) void k_assoc_scan(const AssocArgs a) {
    extern __shared__ double lds_d[];
    constexpr int E = Acc<MV>::E;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wid = tid >> 6;
    const int S = a.b.n_samples, M = a.M, Sc = a.chunk;
    const int s_begin = blockIdx.y * Sc;
    const int ns = min(S - s_begin, Sc);  // multiple of 4
    // row stride: the by-entry drain reads one sample of MANY rows at once; a stride that is a
    // multiple of 32 banks would put them all on one bank pair
    const int Sr = Sc + (MV > 4 ? 1 : 0);
    double* vec = lds_d;                                                   // [MV][Sr], rows >= M zero
    uint32_t* maskw = reinterpret_cast<uint32_t*>(vec + (size_t)MV * Sr);  // [Sc/4] one byte per sample
    unsigned char* wave_area = reinterpret_cast<unsigned char*>(maskw + Sc / 4) + (size_t)wid * a.wave_bytes;
    uint16_t* queue = reinterpret_cast<uint16_t*>(wave_area);  // [AS_QCAP] missing-call records
    wave_area += AS_QCAP * sizeof(uint16_t);
    constexpr int U = MV == 1 ? 4 : 2;  // 16-byte loads in flight per lane
    constexpr bool BY_SAMPLE = MV <= 4;
    constexpr int NCS = BY_SAMPLE ? (MV + 1) * (MV + 2) / 2 : 1;

    // stage this chunk of the sample vectors (zero for samples outside the regression set)
    for (int i = tid; i < ns; i += AS_THREADS) {
        const bool in = !MASK || a.sample_in[s_begin + i];
        if (MASK) reinterpret_cast<unsigned char*>(maskw)[i] = in ? 1 : 0;
#pragma unroll
        for (int k = 0; k < MV; ++k) vec[(size_t)k * Sr + i] = (in && k < M) ? a.vec[(size_t)k * S + s_begin + i] : 0.0;
    }
    __syncthreads();

    int pa[E], pb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane + e * WAVE;
        pa[e] = idx < a.NC ? a.pa[idx] : -1;
        pb[e] = idx < a.NC ? a.pb[idx] : -1;
    }
    const int K = 1 << a.kshift;
    const int nch = ns >> 2;
    const int nfull = nch - nch % (U * WAVE);  // chunks covered by full U-deep iterations
    // waves are independent from here on.  Persistent workgroups (one per CU, the LDS admits no
    // more; the trait vectors are staged once per CU).  Each wave first walks its static share of
    // the first 7/8 of the loci, then takes single loci of the rest from a global counter so that
    // no wave idles at the end (all loci through the counter would saturate same-address atomics:
    // ~10 ns each, 1 ms for 100k loci).
    const int n_waves = gridDim.x * AS_WAVES;
    const int l_dyn0 = a.loci_per_wg ? 0 : (int)(((int64_t)a.b.n_loci * 7 / 8) / n_waves) * n_waves;
    int l_static = a.loci_per_wg ? blockIdx.x * a.loci_per_wg + wid : blockIdx.x * AS_WAVES + wid;
    const int l_end = a.loci_per_wg ? min(a.b.n_loci, (int)(blockIdx.x + 1) * a.loci_per_wg) : a.b.n_loci;
    for (;;) {
        int l;
        if (a.loci_per_wg) {
            l = l_static;
            l_static += AS_WAVES;
        } else if (l_static < l_dyn0) {
            l = l_static;
            l_static += n_waves;
        } else {
            int t = 0;
            if (lane == 0) t = atomicAdd(&a.work_counter[blockIdx.y], 1);
            l = l_dyn0 + __builtin_amdgcn_readfirstlane(t);
        }
        if (l >= l_end) break;
        const int off = a.b.allele_off[l];
        const int A = a.b.allele_off[l + 1] - off;
        double* lut = reinterpret_cast<double*>(wave_area);           // [A+3]
        uint32_t* hist = reinterpret_cast<uint32_t*>(lut + (A + 3));  // [(A+3) << kshift]
        const double pivot = a.allele_len[off];
        for (int i = lane; i < A; i += WAVE) lut[i + 2] = a.allele_len[off + i] - pivot;
        if (lane == 0) {
            lut[0] = -2.0 - pivot;  // GetLengthGenotypes maps the padding index -2 to the length -2
            lut[1] = __builtin_nan("");
            lut[A + 2] = 0.0;
        }
        for (int i = lane; i < ((A + 3) << a.kshift); i += WAVE) hist[i] = 0;
        wave_fence();

        ScanCtx x{lut, hist, vec, Sr, M, a.kshift, lane & (K - 1), (uint32_t)(A + 2) * 0x00010001u};
        Acc<MV> acc;
#pragma unroll
        for (int k = 0; k < MV; ++k) acc.sgv[k] = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) acc.corr[e] = 0.0;
        double cs[NCS];
#pragma unroll
        for (int e = 0; e < NCS; ++e) cs[e] = 0.0;
        int qlen = 0;
        auto drain = [&]() {
            if (BY_SAMPLE) {
                drain_by_sample<MV>(x, queue, qlen, lane, cs);
            } else {
                drain_by_entry<MV>(x, queue, qlen, lane, pa, pb, acc.corr);
            }
            qlen = 0;
        };
        const u32x4* row = reinterpret_cast<const u32x4*>(a.b.gt + ((int64_t)l * S + s_begin) * 2);

        int c0 = 0;
        for (; c0 < nfull; c0 += U * WAVE) {
            u32x4 v[U];
            uint32_t mk[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = __builtin_nontemporal_load(&row[c0 + u * WAVE + lane]);
                mk[u] = MASK ? maskw[c0 + u * WAVE + lane] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                rr[u] = scan_chunk<MV, MASK, false>(x, v[u], mk[u], (c0 + u * WAVE + lane) * 4, true, acc);
#pragma unroll
            for (int u = 0; u < U; ++u) queue_push(queue, qlen, rr[u], c0 + u * WAVE + lane);
            if (qlen > AS_QCAP - U * WAVE) drain();
        }
        for (; c0 < nch; c0 += WAVE) {
            const int c = c0 + lane;
            const bool live = c < nch;
            u32x4 v = {0u, 0u, 0u, 0u};
            uint32_t mk = 0;
            if (live) {
                v = __builtin_nontemporal_load(&row[c]);
                if (MASK) mk = maskw[c];
            }
            const uint32_t r = scan_chunk<MV, MASK, true>(x, v, mk, c * 4, live, acc);
            queue_push(queue, qlen, r, c);
            if (qlen > AS_QCAP - U * WAVE) drain();
        }
        drain();
        wave_fence();
        // ---- reduce and write this (chunk, locus) record --------------------------------------
        double* rec = a.partial + ((size_t)blockIdx.y * a.b.n_loci + l) * a.NS;
        const double sg = wave_sum_f64(acc.sg), sgg = wave_sum_f64(acc.sgg);
        if (lane == 0) {
            rec[1] = sg;
            rec[2] = sgg;
        }
#pragma unroll
        for (int k = 0; k < MV; ++k)
            if (k < M) {
                const double t = wave_sum_f64(acc.sgv[k]);
                if (lane == 0) rec[3 + k] = t;
            }
        if (BY_SAMPLE) {
            int e = 0;
#pragma unroll
            for (int r = 0; r <= MV; ++r)
#pragma unroll
                for (int cc = r; cc <= MV; ++cc) {
                    const double t = wave_sum_f64(cs[e++]);
                    // static rows -> this call's rows: vectors beyond M do not exist, ones = row M
                    const int rr = r == MV ? M : r, rc = cc == MV ? M : cc;
                    if (lane == 0 && (r == MV || r < M) && (cc == MV || cc < M))
                        rec[3 + M + rr * (M + 1) - rr * (rr - 1) / 2 + (rc - rr)] = t;
                }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (pa[e] >= 0) rec[3 + M + lane + e * WAVE] = acc.corr[e];
        }
        for (int bin = lane; bin < A + 3; bin += WAVE) {
            uint32_t sum = 0;
            for (int k = 0; k < K; ++k) sum += hist[(bin << a.kshift) + ((k + lane) & (K - 1))];
            if (bin >= 2 && bin < A + 2) {
                if (a.nchunks == 1)
                    a.allele_count[off + bin - 2] = (int32_t)sum;
                else if (sum)
                    atomicAdd(&a.allele_count[off + bin - 2], (int32_t)sum);
            }
            if (bin == 1) rec[0] = (double)(ns - (int)(sum >> 1));  // tested samples of this chunk
            if (bin == A + 2) rec[a.NS - 1] = (double)sum;
        }
        wave_fence();
    }
}

// -------------------------------------------------------------------------------------------
// streaming scan, ONE or TWO trait vectors (associaTR's default: the outcome alone, or with one covariate).
// Same layout and results as k_assoc_scan; rewritten around what the counters say of it at 100k x 10k
// (SQ_ACTIVE_INST_VALU = 86 % of the wave-cycles of a SIMD, 22 VALU instructions per 64 calls of which 7 are per-locus
// work): the kernel is bound by vector instruction issue, not by LDS or HBM.  So:
//   * the four calls of a 16-byte chunk are a batch: their eight LUT reads and the trait values are issued back to
//     back ahead of the chunk's histogram atomics (reads cannot pass atomics, so the per-call version met the LDS
//     latency once per call);
//   * LDS byte addresses straight from the packed 16-bit bins with v_mad_u32_u16 (op_sel picks the high half): one
//     instruction per address instead of and/shift + shift-add; the not-tested select is made once on the packed
//     pair (-> bins 1, 1); without a sample mask the summed length of a missing call is NaN with a zero low word,
//     so clearing its HIGH word alone makes it +0.0;
//   * sum g is not accumulated per call: it follows from the histogram (sum over bins of count x length);
//   * histogram copies per LOCUS: as many as its bins leave room for in the wave's 3 KiB (32 = one per bank of a
//     32-lane group, conflict free), folded by bins x parts lanes with 16-byte reads;
//   * the next U chunks of the row are requested while the current ones are processed, the first ones before the
//     locus's tables are built;
//   * missing calls are queued once per U chunks (one 32-bit record: first chunk, 4U-bit set) instead of per chunk;
//   * the per-locus wave reductions (sum g^2, sum g v, the Gram corrections) go through LDS, eight values at a
//     time (column sums by 8 lanes each + three shuffle steps), instead of six shuffle steps per value.
// 100k x 10k, 1 trait: 1.05 -> ... ms (profiles/r02_notes.md section 7).
// -------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const double* lds_cf64;
typedef __attribute__((address_space(3))) uint32_t* lds_u32;
__device__ __forceinline__ uint32_t mad16_lo(uint32_t t, uint32_t k, uint32_t b) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(t), "v"(k), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t mad16_hi(uint32_t t, uint32_t k, uint32_t b) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(t), "v"(k), "v"(b));
    return r;
}
struct QuadCtx {
    const double* vec;  // LDS [MV][Sr]
    int Sr;
    uint32_t lut_b;     // LDS byte address of the length LUT (by bin)
    uint32_t hist_b;    // LDS byte address of this lane's histogram column
    uint32_t kbytes;    // bytes between two bins of the histogram (4 << kshift)
    uint32_t eight;
    uint32_t amax2;
};
template <int MV>
struct FewAcc {
    double sgg = 0.0;
    double sgv[MV];
};
// The four calls of one 16-byte chunk; `mk`: their regression-set bytes (MASK only).  Returns the 4-bit set of
// samples that are in the regression set but not called here.
template <int MV, bool MASK>
__device__ __forceinline__ uint32_t scan_quad(const QuadCtx& q, const u32x4 v, uint32_t mk, int s0, FewAcc<MV>& acc) {
    uint32_t t[4];
    double gl[4], gh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = v[j];   // (a bit_cast of the vector element itself reads element 0)
        u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
        u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, q.amax2));
        t[j] = __builtin_bit_cast(uint32_t, t2);
        gl[j] = *(lds_cf64)(uintptr_t)mad16_lo(t[j], q.eight, q.lut_b);
        gh[j] = *(lds_cf64)(uintptr_t)mad16_hi(t[j], q.eight, q.lut_b);
    }
    double y[MV][4];
#pragma unroll
    for (int k = 0; k < MV; ++k) {
        const double2 a0 = *reinterpret_cast<const double2*>(&q.vec[(size_t)k * q.Sr + s0]);
        const double2 a1 = *reinterpret_cast<const double2*>(&q.vec[(size_t)k * q.Sr + s0 + 2]);
        y[k][0] = a0.x; y[k][1] = a0.y; y[k][2] = a1.x; y[k][3] = a1.y;
    }
    uint32_t rare = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double g = gl[j] + gh[j];   // NaN when either haplotype is '-1'
        const bool called = g == g;
        bool ok = called;
        // the flag: rare = 2 rare + (this call), the wave-wide mask of the test as the carry of ONE add (a select and
        // an OR before; the kernel is bound by vector issue, profiles/r04_notes.md section 12) -- call j ends at bit 3 - j
        if (MASK) {
            const bool in = (mk >> (8 * j)) & 1u;
            ok = called & in;
            const uint64_t mm = __ballot(in & !called);
            asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(rare) : "s"(mm) : "vcc");
            g = ok ? g : 0.0;
        } else {
            const uint64_t mm = __ballot(!called);
            asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(rare) : "s"(mm) : "vcc");
            // the NaN is the LUT's (payload 0, and NaN + x keeps it): without its high word it is +0.0
            const uint64_t gb = __builtin_bit_cast(uint64_t, g);
            const uint32_t hi = called ? (uint32_t)(gb >> 32) : 0u;
            g = __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint32_t)gb);
        }
        // calls that are not tested are counted in bin 1 (2 per call): n = calls - bin1 / 2
        const uint32_t tb = ok ? t[j] : 0x00010001u;
        __hip_atomic_fetch_add((lds_u32)(uintptr_t)mad16_lo(tb, q.kbytes, q.hist_b), 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add((lds_u32)(uintptr_t)mad16_hi(tb, q.kbytes, q.hist_b), 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        acc.sgg = __builtin_fma(g, g, acc.sgg);
#pragma unroll
        for (int k = 0; k < MV; ++k) acc.sgv[k] = __builtin_fma(g, y[k][j], acc.sgv[k]);
    }
    return rare;
}

// U 16-byte chunks of a genotype row per lane, chunk indices past `last` clamped (the consumer skips them)
template <int U>
__device__ __forceinline__ void scan_fetch(const u32x4* __restrict__ row, int lane, int last, u32x4 (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = lane + u * WAVE;
        v[u] = __builtin_nontemporal_load(&row[c < last ? c : (last > 0 ? last : 0)]);
    }
}

constexpr int AF_QCAP = 256;  // 32-bit records per wave: first chunk of the lane's iteration << 16 | 4U-bit set
constexpr int AF_AREA = 3072; // histogram + LUT of the wave's current locus

// lane = queued record; every lane keeps the (small) Gram triangle of its samples in registers -- rows 0..MV-1 the
// vectors (zero beyond M), row MV the ones
template <int MV>
__device__ __forceinline__ void drain_few(const QuadCtx& q, const uint32_t* queue, int qlen, int lane, double* cs) {
    wave_fence();
    for (int b0 = 0; b0 < qlen; b0 += WAVE) {
        const uint32_t rec = b0 + lane < qlen ? queue[b0 + lane] : 0u;
        uint32_t bits = rec & 0xffffu;
        const int c0 = (int)(rec >> 16);
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int s = ((c0 + (b >> 2) * WAVE) << 2) + (3 - (b & 3));   // (scan_quad: call j of a chunk at bit 3 - j)
            double z[MV + 1];
#pragma unroll
            for (int k = 0; k < MV; ++k) z[k] = q.vec[(size_t)k * q.Sr + s];
            z[MV] = 1.0;
            int e = 0;
#pragma unroll
            for (int r = 0; r <= MV; ++r)
#pragma unroll
                for (int cc = r; cc <= MV; ++cc) cs[e++] += z[r] * z[cc];
        }
    }
    wave_fence();
}

template <int MV, bool MASK>
__global__ __launch_bounds__(AS_THREADS) void k_assoc_scan_few(const AssocArgs a) {
    static_assert(MV == 1 || MV == 2, "trait vectors");
    extern __shared__ double lds_d[];
    constexpr int U = MV == 1 ? 4 : 2;             // 16-byte chunks per lane and iteration
    constexpr int NCS = (MV + 1) * (MV + 2) / 2;   // Gram triangle of [vec..., 1]
    constexpr int NV = 2 + MV + NCS;               // per-locus wave sums: g^2, g, g v_k, corrections
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wid = tid >> 6;
    const int S = a.b.n_samples, M = a.M, Sc = a.chunk;
    const int s_begin = blockIdx.y * Sc;
    const int ns = min(S - s_begin, Sc);  // multiple of 4
    const int Sr = Sc;
    double* vec = lds_d;                                                   // [MV][Sr], rows >= M zero
    uint32_t* maskw = reinterpret_cast<uint32_t*>(vec + (size_t)MV * Sr);  // [Sc/4] one byte per sample
    unsigned char* wave_base = reinterpret_cast<unsigned char*>(maskw + Sc / 4) + (size_t)wid * a.wave_bytes;
    uint32_t* queue = reinterpret_cast<uint32_t*>(wave_base);              // [AF_QCAP] missing-call records
    unsigned char* area_p = wave_base + AF_QCAP * sizeof(uint32_t);        // [AF_AREA]
    double* red = reinterpret_cast<double*>(wave_base);                    // [8][WAVE], end of a locus

    // stage this chunk of the sample vectors (zero for samples outside the regression set)
    for (int i = tid; i < ns; i += AS_THREADS) {
        const bool in = !MASK || a.sample_in[s_begin + i];
        if (MASK) reinterpret_cast<unsigned char*>(maskw)[i] = in ? 1 : 0;
#pragma unroll
        for (int k = 0; k < MV; ++k) vec[(size_t)k * Sr + i] = (in && k < M) ? a.vec[(size_t)k * S + s_begin + i] : 0.0;
    }
    __syncthreads();

    const int nch = ns >> 2;
    // waves are independent from here on; loci are handed out as in k_assoc_scan
    const int n_waves = gridDim.x * AS_WAVES;
    const int l_dyn0 = a.loci_per_wg ? 0 : (int)(((int64_t)a.b.n_loci * 7 / 8) / n_waves) * n_waves;
    int l_static = a.loci_per_wg ? blockIdx.x * a.loci_per_wg + wid : blockIdx.x * AS_WAVES + wid;
    const int l_end = a.loci_per_wg ? min(a.b.n_loci, (int)(blockIdx.x + 1) * a.loci_per_wg) : a.b.n_loci;
    for (;;) {
        int l;
        if (a.loci_per_wg) {
            l = l_static;
            l_static += AS_WAVES;
        } else if (l_static < l_dyn0) {
            l = l_static;
            l_static += n_waves;
        } else {
            int t = 0;
            if (lane == 0) t = atomicAdd(&a.work_counter[blockIdx.y], 1);
            l = l_dyn0 + __builtin_amdgcn_readfirstlane(t);
        }
        if (l >= l_end) break;
        // the row's first chunks travel while the wave builds the locus's tables
        const u32x4* row = reinterpret_cast<const u32x4*>(a.b.gt + ((int64_t)l * S + s_begin) * 2);
        u32x4 cur[U];
        scan_fetch<U>(row, lane, nch - 1, cur);
        const int off = a.b.allele_off[l];
        const int A = a.b.allele_off[l + 1] - off;
        const int nb = A + 3;   // bins: 0 '-2', 1 '-1' / not tested, 2..A+1 alleles, A+2 out of range
        int kshift = 5;
        while (kshift > 0 && nb * (8 + (4 << kshift)) + 8 > AF_AREA) --kshift;
        const int K = 1 << kshift;
        uint32_t* hist = reinterpret_cast<uint32_t*>(area_p);                                   // [nb << kshift]
        double* lut = reinterpret_cast<double*>(area_p + (((nb << kshift) * 4 + 7) & ~7));      // [nb]
        const double pivot = a.allele_len[off];
        for (int i = lane; i < A; i += WAVE) lut[i + 2] = a.allele_len[off + i] - pivot;
        if (lane == 0) {
            lut[0] = -2.0 - pivot;  // GetLengthGenotypes maps the padding index -2 to the length -2
            lut[1] = __builtin_nan("");
            lut[A + 2] = 0.0;
        }
        if (kshift >= 2) {
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            for (int i = lane; i < (nb << (kshift - 2)); i += WAVE) reinterpret_cast<u32x4*>(hist)[i] = z4;
        } else {
            for (int i = lane; i < (nb << kshift); i += WAVE) hist[i] = 0;
        }
        wave_fence();

        const QuadCtx qc{vec, Sr, (uint32_t)(uintptr_t)(lds_cf64)lut,
                         (uint32_t)(uintptr_t)(lds_u32)hist + 4u * (uint32_t)(lane & (K - 1)), 4u << kshift, 8u,
                         (uint32_t)(A + 2) * 0x00010001u};
        FewAcc<MV> acc;
#pragma unroll
        for (int k = 0; k < MV; ++k) acc.sgv[k] = 0.0;
        double cs[NCS];
#pragma unroll
        for (int e = 0; e < NCS; ++e) cs[e] = 0.0;
        int qlen = 0;
        for (int base = 0; base < nch; base += U * WAVE) {
            u32x4 nxt[U];
            const bool more = base + U * WAVE < nch;   // uniform over the wave
            if (more) scan_fetch<U>(row + base + U * WAVE, lane, nch - 1 - (base + U * WAVE), nxt);
            uint32_t set = 0;
            if (base + U * WAVE <= nch) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = base + u * WAVE + lane;
                    set |= scan_quad<MV, MASK>(qc, cur[u], MASK ? maskw[c] : 0u, c * 4, acc) << (4 * u);
                }
            } else {   // the row's last, partial iteration: lanes past the end sit out
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = base + u * WAVE + lane;
                    if (c < nch) set |= scan_quad<MV, MASK>(qc, cur[u], MASK ? maskw[c] : 0u, c * 4, acc) << (4 * u);
                }
            }
            {   // queue the lanes that met missing calls (ballot + mbcnt: deterministic order)
                const uint64_t mm = __ballot(set != 0);
                if (mm) {
                    const int pos = qlen + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                    if (set) queue[pos] = ((uint32_t)(base + lane) << 16) | set;
                    qlen += __popcll(mm);
                }
            }
            if (qlen > AF_QCAP - WAVE) {
                drain_few<MV>(qc, queue, qlen, lane, cs);
                qlen = 0;
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            }
        }
        drain_few<MV>(qc, queue, qlen, lane, cs);
        // ---- fold the histogram: lanes = bins x parts, every part adds K / parts copies read 16 bytes at a time ----
        double* rec = a.partial + ((size_t)blockIdx.y * a.b.n_loci + l) * a.NS;
        double sg_h = 0.0;   // sum g of the tested calls = sum over their haplotypes' bins of count x length
        auto bin_done = [&](int bin, uint32_t sum) {
            if (bin != 1) sg_h += (double)sum * lut[bin];
            if (bin >= 2 && bin < A + 2) {
                if (a.nchunks == 1)
                    a.allele_count[off + bin - 2] = (int32_t)sum;
                else if (sum)
                    atomicAdd(&a.allele_count[off + bin - 2], (int32_t)sum);
            }
            if (bin == 1) rec[0] = (double)(ns - (int)(sum >> 1));  // tested samples of this chunk
            if (bin == A + 2) rec[a.NS - 1] = (double)sum;
        };
        if (kshift >= 2) {
            int pshift = kshift - 2;
            while (pshift > 0 && (nb << pshift) > WAVE) --pshift;
            const int part = lane & ((1 << pshift) - 1);
            const int per4 = K >> (pshift + 2);   // 16-byte reads per part
            for (int bin0 = 0; bin0 < nb; bin0 += WAVE >> pshift) {
                const int bin = bin0 + (lane >> pshift);
                uint32_t sum = 0;
                if (bin < nb) {
                    const u32x4* hp = reinterpret_cast<const u32x4*>(hist + (bin << kshift)) + part * per4;
                    for (int i = 0; i < per4; ++i) {
                        const u32x4 h = hp[i];
                        sum += (h.x + h.y) + (h.z + h.w);
                    }
                }
                for (int o = (1 << pshift) >> 1; o > 0; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, WAVE);
                if (bin < nb && part == 0) bin_done(bin, sum);
            }
        } else {
            for (int bin = lane; bin < nb; bin += WAVE) {
                uint32_t sum = 0;
                for (int k = 0; k < K; ++k) sum += hist[(bin << kshift) + k];
                bin_done(bin, sum);
            }
        }
        wave_fence();
        // ---- the per-lane sums, eight values at a time through the wave's LDS (queue and tables are done with) ----
        double vals[NV];
        vals[0] = acc.sgg;
        vals[1] = sg_h;
#pragma unroll
        for (int k = 0; k < MV; ++k) vals[2 + k] = acc.sgv[k];
#pragma unroll
        for (int e = 0; e < NCS; ++e) vals[2 + MV + e] = cs[e];
#pragma unroll
        for (int v0 = 0; v0 < NV; v0 += 8) {
#pragma unroll
            for (int v = v0; v < NV && v < v0 + 8; ++v) red[(v - v0) * WAVE + lane] = vals[v];
            wave_fence();
            const int vi = v0 + (lane >> 3), p8 = lane & 7;
            double sum = 0.0;
            if (vi < NV) {
                const double2* rp = reinterpret_cast<const double2*>(red + (vi - v0) * WAVE + p8 * 8);
                const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
                sum = ((r0.x + r0.y) + (r1.x + r1.y)) + ((r2.x + r2.y) + (r3.x + r3.y));
            }
            sum += __shfl_xor(sum, 1, WAVE);
            sum += __shfl_xor(sum, 2, WAVE);
            sum += __shfl_xor(sum, 4, WAVE);
            if (vi < NV && p8 == 0) {
                int tgt = -1;   // column of the partial record (static rows -> this call's rows: vectors beyond M
                if (vi == 0) {  // do not exist, the ones are row M)
                    tgt = 2;
                } else if (vi == 1) {
                    tgt = 1;
                } else if (vi < 2 + MV) {
                    if (vi - 2 < M) tgt = 3 + (vi - 2);
                } else {
                    int e = vi - 2 - MV, r = 0;
                    while (e >= MV + 1 - r) {
                        e -= MV + 1 - r;
                        ++r;
                    }
                    const int cc = r + e;
                    if ((r == MV || r < M) && (cc == MV || cc < M)) {
                        const int rr = r == MV ? M : r, rc = cc == MV ? M : cc;
                        tgt = 3 + M + rr * (M + 1) - rr * (rr - 1) / 2 + (rc - rr);
                    }
                }
                if (tgt >= 0) rec[tgt] = sum;
            }
            wave_fence();
        }
    }
}

// -------------------------------------------------------------------------------------------
// streaming scan, many covariates: the cross-products  sum_s g[l][s] * v_k[s]  of SIXTEEN loci
// and up to 16*RT vector rows are a GEMM tile, done with v_mfma_f64_16x16x4_f64.
//   workgroup = 16 waves = 16 loci; the row is walked in steps of 256 samples:
//     * all threads stage the step's block of the vectors (+ a row of ones, + zero rows up to
//       16*RT) into LDS -- loaded a step ahead into registers, as is each wave's genotype chunk;
//     * every wave decodes ITS locus as in the one-trait kernel (NaN LUT, sum g^2, histogram,
//       tested-sample count from the '-1' bin) and writes the 256 summed lengths to its row of
//       the G tile in LDS;                                               -- barrier --
//     * MFMA: wave w multiplies columns [16w, 16w+16) of the G tile (A: 16 loci x 4 samples) with
//       the same columns of the vector block (B: 4 samples x 16 rows), accumulating C[16 x 16*RT];
//     * missing calls: each wave gathers the vector columns of ITS locus's missing samples (still
//       resident) four at a time and accumulates their Gram matrix, A = B = Z (16 rows x 4
//       samples) per tile pair;                                           -- barrier --
//   Inside a step the 256 samples sit in LDS in the order (cell j of the 16-byte chunk, lane):
//   column j*64 + lane holds sample 4*lane + j, for the G tile and the vector block alike, so
//   that a wave's 64 lanes write consecutive doubles (sample order would be a 16-way conflict).
//   end of row: the 16 partial C tiles are summed through LDS in wave order (deterministic).
// MFMA lane layout (probed on gfx950, tools/mfma_probe): A lane i = A[i%16][i/16],
// B lane i = B[i/16][i%16], D lane i reg r = D[4r + i/16][i%16].
// -------------------------------------------------------------------------------------------
constexpr int MF_SB = 256;    // samples per step
constexpr int MF_SBR = 257;   // LDS row stride (odd: 16 rows of one column fall on 16 bank pairs);
                              // column MF_SB is a zero column (padding of the missing-sample batches)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int RT, bool MASK>
__global__ __launch_bounds__(AS_THREADS) void k_assoc_scan_mfma(const AssocArgs a) {
    extern __shared__ double lds_d[];
    constexpr int ROWS = 16 * RT;
    __shared__ int tile_sh;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wid = tid >> 6;
    const int S = a.b.n_samples, M = a.M, L = a.b.n_loci;
    double* Gt2 = lds_d;                // [2][16][MF_SBR]  summed lengths of the 16 loci (0 where not tested), two steps
    unsigned char* wave_area = reinterpret_cast<unsigned char*>(Gt2 + 2 * 16 * MF_SBR) + (size_t)wid * a.wave_bytes;
    const int K = 1 << a.kshift;
    const int nsteps = a.nsteps;
    // B operand of this lane: row (lane & 15) of a tile, the sample behind column 16 wid + 4 ks + lane / 16 of the
    // step (column j * 64 + c holds sample 4 c + j): straight from the sample-major vector block, 128 contiguous
    // bytes per sample and tile.  a.vect has nsteps * 256 rows (zero beyond S).
    const int bj = wid >> 2;
    const double* vrow = a.vect + (lane & 15);
    auto load_b = [&](int step, double (*out)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ((16 * wid + 4 * ks) & 63) + (lane >> 4);
            const size_t s = (size_t)step * MF_SB + 4 * c + bj;
#pragma unroll
            for (int t = 0; t < RT; ++t) out[t][ks] = vrow[s * ROWS + 16 * t];
        }
    };

    for (;;) {
        if (tid == 0) tile_sh = atomicAdd(&a.work_counter[0], 1);
        __syncthreads();
        const int tile = tile_sh;
        __syncthreads();
        if (tile * 16 >= L) break;
        const int l = tile * 16 + wid;
        const bool has = l < L;
        const int off = has ? a.b.allele_off[l] : 0;
        const int A = has ? a.b.allele_off[l + 1] - off : 1;
        double* lut = reinterpret_cast<double*>(wave_area);
        uint32_t* hist = reinterpret_cast<uint32_t*>(lut + (A + 3));
        if (has) {
            const double pivot = a.allele_len[off];
            for (int i = lane; i < A; i += WAVE) lut[i + 2] = a.allele_len[off + i] - pivot;
            if (lane == 0) {
                lut[0] = -2.0 - pivot;
                lut[1] = __builtin_nan("");
                lut[A + 2] = 0.0;
            }
            for (int i = lane; i < ((A + 3) << a.kshift); i += WAVE) hist[i] = 0;
        }
        wave_fence();
        const uint32_t amax2 = (uint32_t)(A + 2) * 0x00010001u;
        const int kslot = lane & (K - 1);
        const u32x4* row = reinterpret_cast<const u32x4*>(a.b.gt + (int64_t)(has ? l : 0) * S * 2);

        d4 C[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) C[t] = (d4){0.0, 0.0, 0.0, 0.0};
        double sgg = 0.0;
        double a_def[4] = {0.0, 0.0, 0.0, 0.0}, b_def[RT][4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) b_def[t][j] = 0.0;

        const u32x4 dead = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        u32x4 vnext = dead;
        uint32_t mnext = 0x01010101u;
        if (has && 4 * lane < S) {
            vnext = __builtin_nontemporal_load(&row[lane]);
            if (MASK) mnext = *reinterpret_cast<const uint32_t*>(a.sample_in + 4 * lane);
        }

        for (int step = 0; step < nsteps; ++step) {
            const int s0 = step * MF_SB;
            double* Gt = Gt2 + (step & 1) * 16 * MF_SBR;
            const u32x4 v = vnext;
            const uint32_t mk = mnext;
            const bool live = has && s0 + 4 * lane < S;
            uint32_t rare = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the PREVIOUS step's tile product, from registers: the matrix pipe works while this
                // wave (and its three SIMD mates) decode the next calls
#pragma unroll
                for (int t = 0; t < RT; ++t) C[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_def[j], b_def[t][j], C[t], 0, 0, 0);
                const uint32_t w = v[j];
                u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
                u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, amax2));
                const uint32_t t = __builtin_bit_cast(uint32_t, t2);
                const uint32_t lo = t & 0xffffu, hi = t >> 16;
                double g = lut[lo] + lut[hi];
                const bool called = g == g;
                const bool in = !MASK || ((mk >> (8 * j)) & 0xffu) != 0;
                const bool ok = called & in & live;
                rare |= (uint32_t)(in & !called & live) << j;
                g = ok ? g : 0.0;
                Gt[wid * MF_SBR + j * 64 + lane] = g;      // (this step's buffer: its last readers are behind the previous barrier)
                sgg = __builtin_fma(g, g, sgg);
                if (live) {
                    atomicAdd(&hist[((ok ? lo : 1u) << a.kshift) + kslot], 1u);
                    atomicAdd(&hist[((ok ? hi : 1u) << a.kshift) + kslot], 1u);
                }
            }
            // ---- next step's genotype chunk and this step's B operands (they arrive during the decode of the
            //      next step, where they are multiplied) ----------------------------------------------------
            vnext = dead;
            mnext = 0x01010101u;
            if (step + 1 < nsteps) {
                const int sn = s0 + MF_SB + 4 * lane;
                if (has && sn < S) {
                    vnext = __builtin_nontemporal_load(&row[(sn >> 2)]);
                    if (MASK) mnext = *reinterpret_cast<const uint32_t*>(a.sample_in + sn);
                }
            }
            load_b(step, b_def);
            // ---- this wave's missing samples of the step: four ballots, for k_assoc_gram_miss ---------
            {
                unsigned long long mine = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long mm = __ballot((rare >> j) & 1u);
                    if (lane == j) mine = mm;
                }
                if (has && lane < 4) a.missbits[((size_t)l * nsteps + step) * 4 + lane] = mine;
            }
            __syncthreads();     // the G tile of this step is complete (the other buffer is free: its readers are past
                                 // the previous barrier)
            // ---- A operands of this wave's share of the tile product, columns [16 wid, 16 wid + 16) ----
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a_def[ks] = Gt[(lane & 15) * MF_SBR + 16 * wid + 4 * ks + (lane >> 4)];
        }

        // ---- end of the row --------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)  // the last step's tile product
#pragma unroll
            for (int t = 0; t < RT; ++t) C[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_def[ks], b_def[t][ks], C[t], 0, 0, 0);
        double* rec = a.partial + (size_t)(has ? l : 0) * a.NS;
        __syncthreads();            // every wave has read its last A operands: the G tile buffers become the reduction area
        double* red = Gt2;  // [16 waves][256]
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wid * 256 + lane * 4 + r] = C[t][r];
            __syncthreads();
            if (lane < 16) {
                // locus wid = 4*reg + lane/16  ->  reg = wid / 4, lane group wid % 4; vector row 16 t + lane
                double sum = 0.0;
                for (int w2 = 0; w2 < AS_WAVES; ++w2) sum += red[w2 * 256 + (16 * (wid & 3) + lane) * 4 + (wid >> 2)];
                const int vrow_i = 16 * t + lane;
                if (has) {
                    if (vrow_i < M) rec[3 + vrow_i] = sum;
                    else if (vrow_i == M) rec[1] = sum;   // the row of ones: sum g
                }
            }
            __syncthreads();
        }
        const double sgg_w = wave_sum_f64(sgg);
        if (has) {
            if (lane == 0) rec[2] = sgg_w;
            for (int bin = lane; bin < A + 3; bin += WAVE) {
                uint32_t sum = 0;
                for (int k = 0; k < K; ++k) sum += hist[(bin << a.kshift) + ((k + lane) & (K - 1))];
                if (bin >= 2 && bin < A + 2) a.allele_count[off + bin - 2] = (int32_t)sum;
                if (bin == 1) rec[0] = (double)(S - (int)(sum >> 1));
                if (bin == A + 2) rec[a.NS - 1] = (double)sum;
            }
        }
        wave_fence();
    }
}

// -------------------------------------------------------------------------------------------
// Gram correction of the MFMA scan: sum over the MISSING calls of a locus (regression-set samples whose call is not
// made) of z z^T, z = the sample's column of [vectors..., 1] -- what the finaliser subtracts from the Gram matrix of
// the whole regression set.  The scan leaves one bit per sample (AssocArgs.missbits); here one wave per locus walks
// the bits, gathers the samples' columns from the sample-major copy of the vector block (AssocArgs.vect: a column is
// 128 contiguous bytes per 16-row tile) four samples at a time, A = B = Z (16 rows x 4 samples) per tile pair, and
// writes the NC entries of the record.  (Inside the scan's step loop this cost 1.0 ms of 3.1 at M = 15 and 2.3 ms
// of 5.3 at M = 31, 100k x 10k: the accumulators of the tile pairs pushed the scan into scratch.)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assoc_vect(const AssocArgs a, int rows) {
    const int S = a.b.n_samples, M = a.M;
    const int64_t n = (int64_t)a.nsteps * MF_SB * rows;      // whole steps: zero columns beyond S
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i / rows), r = (int)(i - (int64_t)s * rows);
        a.vect[i] = s >= S ? 0.0 : (r < M ? a.vec[(size_t)r * S + s] : (r == M ? 1.0 : 0.0));
    }
}

constexpr int GM_WAVES = 4;     // loci per workgroup
constexpr int GM_LIST = 1032;   // pending sample indices of one wave: a quarter round (16 words) always fits behind the < 4 left over
#ifndef GM_GROUP_ALL
#define GM_GROUP_ALL (RT >= 3 ? 4 : 8)
#endif
template <int RT> struct GmCfg {
    static constexpr int GROUP = GM_GROUP_ALL;    // batches of four samples per software-pipeline stage
    static constexpr bool TWO = RT == 1;            // a second accumulator set (one tile pair: consecutive products
};                                                  // would wait for each other)
template <int RT>
__global__ __launch_bounds__(WAVE* GM_WAVES) void k_assoc_gram_miss(const AssocArgs a) {
    extern __shared__ unsigned long long gm_lds[];
    constexpr int ROWS = 16 * RT;
    constexpr int NP = RT * (RT + 1) / 2;
    constexpr int GM_GROUP = GmCfg<RT>::GROUP;
    constexpr bool TWO = GmCfg<RT>::TWO;
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    const int l = blockIdx.x * GM_WAVES + wid;
    const int M = a.M, nw = a.nsteps * 4;
    if (l >= a.b.n_loci) return;
    uint32_t* list = reinterpret_cast<uint32_t*>(gm_lds) + wid * GM_LIST;
    const unsigned long long* mb = a.missbits + (size_t)l * nw;
    d4 Gc[NP], Gd[NP];     // two accumulator sets, alternating: consecutive products do not wait for each other
#pragma unroll
    for (int p = 0; p < NP; ++p) Gc[p] = Gd[p] = (d4){0.0, 0.0, 0.0, 0.0};
    const double* zbase = a.vect + (lane & 15);
    int cnt = 0;
    // the columns of GM_GROUP batches (four samples each, lane group lane / 16 takes one) starting at entry e0;
    // entries at or beyond `end` are zero columns
    auto fetch = [&](double (*z)[RT], int e0, int end) {
#pragma unroll
        for (int u = 0; u < GM_GROUP; ++u) {
            const int e = e0 + 4 * u + (lane >> 4);
            const bool ok = e < end;
            const uint32_t sx = ok ? list[e] : 0u;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                z[u][t] = ok ? zbase[(size_t)sx * ROWS + 16 * t] : 0.0;
            }
        }
    };
    auto multiply = [&](double (*z)[RT]) {
#pragma unroll
        for (int u = 0; u < GM_GROUP; ++u) {
            int p = 0;
#pragma unroll
            for (int ti = 0; ti < RT; ++ti)
#pragma unroll
                for (int tj = ti; tj < RT; ++tj) {
                    if (TWO && (u & 1)) Gd[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[u][ti], z[u][tj], Gd[p], 0, 0, 0);
                    else Gc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[u][ti], z[u][tj], Gc[p], 0, 0, 0);
                    ++p;
                }
        }
    };
    // consumes the list (all of it when `flush`, else whole batches of four; the rest moves to the front): the
    // columns of the next GM_GROUP batches are requested while the matrix pipe works on the current ones
    auto consume = [&](bool flush) {
        const int end = flush ? cnt : (cnt & ~3);
        constexpr int STRIDE = 4 * GM_GROUP;
        double za[GM_GROUP][RT], zb[GM_GROUP][RT];
        if (end > 0) fetch(za, 0, end);
        for (int e0 = 0; e0 < end; e0 += 2 * STRIDE) {
            if (e0 + STRIDE < end) fetch(zb, e0 + STRIDE, end);
            multiply(za);
            if (e0 + STRIDE < end) {
                if (e0 + 2 * STRIDE < end) fetch(za, e0 + 2 * STRIDE, end);
                multiply(zb);
            }
        }
        wave_fence();
        if (!flush) {
            const int rest = cnt - end;
            uint32_t keep = 0;
            if (lane < rest) keep = list[end + lane];
            wave_fence();
            if (lane < rest) list[lane] = keep;
            cnt = rest;
            wave_fence();
        }
    };
    // the set bits of `mm` (this lane's word w = 4 step + j: bit b <-> sample 256 step + 4 b + j) appended to the list
    auto append = [&](unsigned long long mm, int w) {
        const int c = __popcll(mm);
        int incl = c;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += t;
        }
        const int total = __shfl(incl, WAVE - 1, WAVE);
        if (total == 0) return;
        if (cnt + total > GM_LIST) {
            wave_fence();
            consume(false);
        }
        int pos = cnt + incl - c;
        const uint32_t base = (uint32_t)(w >> 2) * MF_SB + (uint32_t)(w & 3);
        while (mm) {
            const int b = __ffsll((unsigned long long)mm) - 1;
            mm &= mm - 1;
            list[pos++] = base + 4u * (uint32_t)b;
        }
        cnt += total;
    };
    for (int w0 = 0; w0 < nw; w0 += WAVE) {
        const int w = w0 + lane;
        const unsigned long long mm = w < nw ? mb[w] : 0ull;
        if (wave_sum_i32(__popcll(mm)) <= GM_LIST - 4) {
            append(mm, w);
        } else {                                  // a round of mostly missing calls: a quarter of the lanes at a time
            for (int q = 0; q < 4; ++q) append((lane >> 4) == q ? mm : 0ull, w);
        }
    }
    wave_fence();
    consume(true);
#pragma unroll
    for (int p = 0; p < NP; ++p) Gc[p] += Gd[p];
    double* rec = a.partial + (size_t)l * a.NS;
    int p = 0;
#pragma unroll
    for (int ti = 0; ti < RT; ++ti)
#pragma unroll
        for (int tj = ti; tj < RT; ++tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ra = 16 * ti + 4 * r + (lane >> 4), rb = 16 * tj + (lane & 15);
                if (ra <= rb && rb <= M) rec[3 + M + ra * (M + 1) - ra * (ra - 1) / 2 + (rb - ra)] = Gc[p][r];
            }
            ++p;
        }
}

// -------------------------------------------------------------------------------------------
// streaming scan, any ploidy / alignment / allele count: wave per locus, operands from global
// memory, histogram by global atomics (allele_count zeroed by the launcher)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assoc_scan_any(const AssocArgs a) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= a.b.n_loci) return;
    const int S = a.b.n_samples, P = a.b.ploidy, M = a.M;
    const int pl = a.b.locus_ploidy ? a.b.locus_ploidy[l] : P;
    const int off = a.b.allele_off[l];
    const int A = a.b.allele_off[l + 1] - off;
    const double pivot = a.allele_len[off];
    int n = 0, n_bad = 0;
    double sg = 0.0, sgg = 0.0, sgv[AS_MAXV], corr[AS_E];
    for (int k = 0; k < AS_MAXV; ++k) sgv[k] = 0.0;
    for (int e = 0; e < AS_E; ++e) corr[e] = 0.0;
    for (int s0 = 0; s0 < S; s0 += WAVE) {
        const int s = s0 + lane;
        bool in = false, miss = false;
        double g = 0.0;
        if (s < S) {
            in = !a.sample_in || a.sample_in[s];
            const int16_t* cell = a.b.gt + ((int64_t)l * S + s) * P;
            for (int p = 0; p < pl; ++p) miss |= cell[p] == -1;
            if (in && !miss) {
                for (int p = 0; p < pl; ++p) {
                    const int al = cell[p];
                    if (al == -2) {
                        g += -2.0 - pivot;
                    } else if (al >= 0 && al < A) {
                        g += a.allele_len[off + al] - pivot;
                        atomicAdd(&a.allele_count[off + al], 1);
                    } else {
                        ++n_bad;
                    }
                }
                ++n;
                sg += g;
                sgg = __builtin_fma(g, g, sgg);
                for (int k = 0; k < M; ++k) sgv[k] = __builtin_fma(g, a.vec[(size_t)k * S + s], sgv[k]);
            }
        }
        uint64_t mm = __ballot(in & miss);
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int sm = s0 + src;
            for (int e = 0; e < AS_E; ++e) {
                const int idx = lane + e * WAVE;
                if (idx >= a.NC) continue;
                const double x = a.pa[idx] == M ? 1.0 : a.vec[(size_t)a.pa[idx] * S + sm];
                const double y = a.pb[idx] == M ? 1.0 : a.vec[(size_t)a.pb[idx] * S + sm];
                corr[e] += x * y;
            }
        }
    }
    double* rec = a.partial + (size_t)l * a.NS;
    n = wave_sum_i32(n);
    n_bad = wave_sum_i32(n_bad);
    sg = wave_sum_f64(sg);
    sgg = wave_sum_f64(sgg);
    if (lane == 0) {
        rec[0] = (double)n;
        rec[1] = sg;
        rec[2] = sgg;
        rec[a.NS - 1] = (double)n_bad;
    }
    for (int k = 0; k < M; ++k) {
        const double t = wave_sum_f64(sgv[k]);
        if (lane == 0) rec[3 + k] = t;
    }
    for (int e = 0; e < AS_E; ++e) {
        const int idx = lane + e * WAVE;
        if (idx < a.NC) rec[3 + M + idx] = corr[e];
    }
}

// -------------------------------------------------------------------------------------------
// --beagle-dosages: the regressor is the expected summed length from the AP1/AP2 allele
// probabilities (see trk_assoc_dosage in trk.h).  Wave per locus; two passes over the locus's
// samples: sample-major for the regression sums, class-major for the per-class sums.
// -------------------------------------------------------------------------------------------
struct DosArgs {
    trk_assoc_dosage d;
    double* class_sums;
    double* locus_sums;
};

// numpy's float32 pairwise sum of x[0..n) (np.sum(ap[curr, :], axis=1), contiguous rows)
__device__ float np_sum_f32(const float* x, int n) {
    int lo_[12], n_[12], stage[12];
    float left[12];
    int sp = 0;
    lo_[0] = 0;
    n_[0] = n;
    stage[0] = 0;
    float ret = 0.f;
    while (sp >= 0) {
        const int m = n_[sp], lo = lo_[sp];
        if (m <= 128) {
            if (m < 8) {
                float res = 0.f;
                for (int i = 0; i < m; ++i) res += x[lo + i];
                ret = res;
            } else {
                float r[8];
                for (int j = 0; j < 8; ++j) r[j] = x[lo + j];
                int i = 8;
                for (; i < m - (m % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r[j] += x[lo + i + j];
                float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < m; ++i) res += x[lo + i];
                ret = res;
            }
            --sp;
            continue;
        }
        int n2 = m / 2;
        n2 -= n2 % 8;
        if (stage[sp] == 0) {
            stage[sp] = 1;
            lo_[sp + 1] = lo;
            n_[sp + 1] = n2;
            stage[sp + 1] = 0;
            ++sp;
        } else if (stage[sp] == 1) {
            left[sp] = ret;
            stage[sp] = 2;
            lo_[sp + 1] = lo + n2;
            n_[sp + 1] = m - n2;
            stage[sp + 1] = 0;
            ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return 0.f + ret;
}

// Single pass for loci with few alleles (A <= DQ_A): the locus's tables sit in LDS, the row sums
// are numpy's plain left-to-right float32 sums (fewer than 8 terms), and the per-class sums are
// kept in lane-private LDS columns (sized by the batch's largest allele set), so that nothing is
// read twice.  4 waves per workgroup, one locus per wave; the next 64 samples' operands are
// loaded while the current ones are processed.  Same arithmetic as k_assoc_dosage.
constexpr int DQ_A = 8;    // alleles (ref + 7 alternates: numpy sums fewer than 8 floats sequentially)
__global__ __launch_bounds__(256) void k_assoc_dosage_small(const AssocArgs a, const DosArgs q, int cmax) {
    extern __shared__ double dq_lds[];               // [4 waves][cmax * 4][64 lanes] class accumulators
    __shared__ int tab_perm[4][DQ_A], tab_cls[4][DQ_A], tab_best[4][DQ_A];
    __shared__ double tab_val[4][DQ_A], tab_len[4][DQ_A];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    const int l = blockIdx.x * 4 + wid;
    if (l >= a.b.n_loci) return;
    double (*cls_acc)[WAVE] = reinterpret_cast<double (*)[WAVE]>(dq_lds + (size_t)wid * cmax * 4 * WAVE);
    const int S = a.b.n_samples, M = a.M, Kc = q.d.n_alt_cols;
    const int off = a.b.allele_off[l];
    const int A = a.b.allele_off[l + 1] - off;
    if (lane < A) {
        tab_perm[wid][lane] = q.d.perm[off + lane];
        tab_cls[wid][lane] = q.d.dclass[off + lane];
        tab_best[wid][lane] = q.d.best_class[off + lane];
        tab_val[wid][lane] = q.d.dclass_value[off + lane];
        tab_len[wid][lane] = a.allele_len[off + lane];
    }
    for (int i = 0; i < cmax * 4; ++i) cls_acc[i][lane] = 0.0;
    wave_fence();
    int ncls = 0;
    for (int i = 0; i < A; ++i) ncls = max(ncls, tab_cls[wid][i] + 1);
    const double pivot = 2.0 * tab_val[wid][tab_cls[wid][0]];

    int n = 0;
    double sg = 0.0, sgg = 0.0, sgv[AS_MAXV], corr[AS_E];
    double rx = 0.0, rxx = 0.0, ry = 0.0, ryy = 0.0, rxy = 0.0, xmin = INFINITY, xmax = -INFINITY;
    for (int k = 0; k < AS_MAXV; ++k) sgv[k] = 0.0;
    for (int e = 0; e < AS_E; ++e) corr[e] = 0.0;
    // operands of the next 64 samples, in flight while the current ones are processed
    uint32_t w_n = 0xffffffffu;
    float ap_n[2][DQ_A - 1];
    bool in_n = false;
    auto fetch = [&](int s) {
        w_n = 0xffffffffu;
        in_n = false;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < DQ_A - 1; ++i) ap_n[p][i] = 0.f;
        if (s < S) {
            in_n = !a.sample_in || a.sample_in[s];
            w_n = reinterpret_cast<const uint32_t*>(a.b.gt)[(int64_t)l * S + s];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float* ap = (p ? q.d.ap2 : q.d.ap1) + ((int64_t)l * S + s) * Kc;
#pragma unroll
                for (int i = 0; i < DQ_A - 1; ++i)
                    if (i < A - 1) ap_n[p][i] = ap[i];
            }
        }
    };
    fetch(lane);
    for (int s0 = 0; s0 < S; s0 += WAVE) {
        const int s = s0 + lane;
        const uint32_t w = w_n;
        float apv[2][DQ_A - 1];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < DQ_A - 1; ++i) apv[p][i] = ap_n[p][i];
        bool in = in_n, miss = false;
        fetch(s + WAVE);
        if (s < S) {
            const int a0 = (int)(int16_t)(w & 0xffffu), a1 = (int)(int16_t)(w >> 16);
            miss = (a0 == -1) | (a1 == -1);
            if (in && !miss) {
                float x[2][DQ_A];   // x[p][allele]: AP value of every allele (the reference allele derived)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float sum = 0.f;
#pragma unroll
                    for (int i = 1; i < DQ_A; ++i) {
                        const float v = apv[p][i - 1];
                        x[p][i] = v;
                        if (i < A) sum += v;
                    }
                    x[p][0] = fmaxf(0.f, 1.f - sum);
                }
                double g = 0.0, y[2] = {0.0, 0.0}, d[2] = {0.0, 0.0};
                int u = tab_cls[wid][tab_perm[wid][0]];
                const int b0 = (a0 >= 0 && a0 < A) ? tab_best[wid][a0] : 0xffff;
                const int b1 = (a1 >= 0 && a1 < A) ? tab_best[wid][a1] : 0xffff;
                for (int i = 0; i <= A; ++i) {
                    const int al = i < A ? tab_perm[wid][i] : -1;
                    const int u2 = i < A ? tab_cls[wid][al] : -1;
                    if (u2 != u) {  // class u complete
                        const double lv = tab_val[wid][u];
                        g += lv * (d[0] + d[1]);
                        y[0] += lv * d[0];
                        y[1] += lv * d[1];
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const double xi = ((p ? b1 : b0) == u) ? 1.0 : 0.0;
                            cls_acc[u * 4 + 0][lane] += d[p];
                            cls_acc[u * 4 + 1][lane] += d[p] * d[p];
                            cls_acc[u * 4 + 2][lane] += xi;
                            cls_acc[u * 4 + 3][lane] += xi * d[p];
                        }
                        u = u2;
                        d[0] = d[1] = 0.0;
                    }
                    if (i < A) {
                        float v0 = x[0][0], v1 = x[1][0];
#pragma unroll
                        for (int t = 1; t < DQ_A; ++t) {   // static register indexing
                            v0 = al == t ? x[0][t] : v0;
                            v1 = al == t ? x[1][t] : v1;
                        }
                        d[0] += (double)v0;
                        d[1] += (double)v1;
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int al = p ? a1 : a0;
                    const double xv = al == -2 ? -2.0 : (al >= 0 && al < A ? tab_len[wid][al] : 0.0);
                    xmin = fmin(xmin, xv);
                    xmax = fmax(xmax, xv);
                    rx += xv;
                    rxx += xv * xv;
                    ry += y[p];
                    ryy += y[p] * y[p];
                    rxy += xv * y[p];
                }
                g -= pivot;
                ++n;
                sg += g;
                sgg = __builtin_fma(g, g, sgg);
                for (int k = 0; k < M; ++k) sgv[k] = __builtin_fma(g, a.vec[(size_t)k * S + s], sgv[k]);
            }
        }
        uint64_t mm = __ballot(in & miss);
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int sm = s0 + src;
            for (int e = 0; e < AS_E; ++e) {
                const int idx = lane + e * WAVE;
                if (idx >= a.NC) continue;
                const double xa = a.pa[idx] == M ? 1.0 : a.vec[(size_t)a.pa[idx] * S + sm];
                const double xb = a.pb[idx] == M ? 1.0 : a.vec[(size_t)a.pb[idx] * S + sm];
                corr[e] += xa * xb;
            }
        }
    }
    double* rec = a.partial + (size_t)l * a.NS;
    n = wave_sum_i32(n);
    sg = wave_sum_f64(sg);
    sgg = wave_sum_f64(sgg);
    rx = wave_sum_f64(rx);
    rxx = wave_sum_f64(rxx);
    ry = wave_sum_f64(ry);
    ryy = wave_sum_f64(ryy);
    rxy = wave_sum_f64(rxy);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fmin(xmin, __shfl_xor(xmin, o, WAVE));
        xmax = fmax(xmax, __shfl_xor(xmax, o, WAVE));
    }
    if (lane == 0) {
        rec[0] = (double)n;
        rec[1] = sg;
        rec[2] = sgg;
        rec[a.NS - 1] = 0.0;
        double* ls = q.locus_sums + (size_t)l * TRK_ADL_COLS;
        ls[0] = rx; ls[1] = rxx; ls[2] = ry; ls[3] = ryy; ls[4] = rxy; ls[5] = 2.0 * n; ls[6] = xmin; ls[7] = xmax;
    }
    for (int k = 0; k < M; ++k) {
        const double t = wave_sum_f64(sgv[k]);
        if (lane == 0) rec[3 + k] = t;
    }
    for (int e = 0; e < AS_E; ++e) {
        const int idx = lane + e * WAVE;
        if (idx < a.NC) rec[3 + M + idx] = corr[e];
    }
    for (int i = 0; i < ncls * 4; ++i) {
        const double t = wave_sum_f64(cls_acc[i][lane]);
        if (lane == 0) q.class_sums[(size_t)(off + (i >> 2)) * TRK_ADC_COLS + (i & 3)] = t;
    }
}

__global__ __launch_bounds__(256) void k_assoc_dosage(const AssocArgs a, const DosArgs q) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= a.b.n_loci) return;
    const int S = a.b.n_samples, P = a.b.ploidy, M = a.M, Kc = q.d.n_alt_cols;
    const int pl = a.b.locus_ploidy ? a.b.locus_ploidy[l] : P;
    const int off = a.b.allele_off[l];
    const int A = a.b.allele_off[l + 1] - off;
    const int32_t* perm = q.d.perm + off;
    const uint16_t* dcl = q.d.dclass + off;
    const double* dval = q.d.dclass_value + off;
    const uint16_t* bcl = q.d.best_class + off;
    const double pivot = 2.0 * dval[dcl[0]];  // any constant: the variance is shift invariant
    int ncls = 0;
    for (int i = 0; i < A; ++i) ncls = max(ncls, (int)dcl[i] + 1);

    // ---- pass 1: sample-major -----------------------------------------------------------------
    int n = 0;
    double sg = 0.0, sgg = 0.0, sgv[AS_MAXV], corr[AS_E];
    double rx = 0.0, rxx = 0.0, ry = 0.0, ryy = 0.0, rxy = 0.0, xmin = INFINITY, xmax = -INFINITY;
    for (int k = 0; k < AS_MAXV; ++k) sgv[k] = 0.0;
    for (int e = 0; e < AS_E; ++e) corr[e] = 0.0;
    for (int s0 = 0; s0 < S; s0 += WAVE) {
        const int s = s0 + lane;
        bool in = false, miss = false;
        if (s < S) {
            in = !a.sample_in || a.sample_in[s];
            const int16_t* cell = a.b.gt + ((int64_t)l * S + s) * P;
            for (int p = 0; p < pl; ++p) miss |= cell[p] == -1;
            if (in && !miss) {
                const float* ap[2] = {q.d.ap1 + ((int64_t)l * S + s) * Kc, q.d.ap2 + ((int64_t)l * S + s) * Kc};
                float ref[2];
                for (int p = 0; p < 2; ++p) ref[p] = fmaxf(0.f, 1.f - np_sum_f32(ap[p], A - 1));
                double g = 0.0, y[2] = {0.0, 0.0};
                int u = dcl[perm[0]];
                double d[2] = {0.0, 0.0};
                for (int i = 0; i <= A; ++i) {
                    const int al = i < A ? perm[i] : -1;
                    const int u2 = i < A ? (int)dcl[al] : -1;
                    if (u2 != u) {  // class u complete
                        g += dval[u] * (d[0] + d[1]);
                        y[0] += dval[u] * d[0];
                        y[1] += dval[u] * d[1];
                        u = u2;
                        d[0] = d[1] = 0.0;
                    }
                    if (i < A)
                        for (int p = 0; p < 2; ++p) d[p] += (double)(al == 0 ? ref[p] : ap[p][al - 1]);
                }
                // best-guess length per haplotype (GetLengthGenotypes: -2 stays -2)
                for (int p = 0; p < 2 && p < pl; ++p) {
                    const int al = cell[p];
                    const double x = al == -2 ? -2.0 : (al >= 0 && al < A ? a.allele_len[off + al] : 0.0);
                    xmin = fmin(xmin, x);
                    xmax = fmax(xmax, x);
                    rx += x;
                    rxx += x * x;
                    ry += y[p];
                    ryy += y[p] * y[p];
                    rxy += x * y[p];
                }
                g -= pivot;
                ++n;
                sg += g;
                sgg = __builtin_fma(g, g, sgg);
                for (int k = 0; k < M; ++k) sgv[k] = __builtin_fma(g, a.vec[(size_t)k * S + s], sgv[k]);
            }
        }
        uint64_t mm = __ballot(in & miss);
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int sm = s0 + src;
            for (int e = 0; e < AS_E; ++e) {
                const int idx = lane + e * WAVE;
                if (idx >= a.NC) continue;
                const double xa = a.pa[idx] == M ? 1.0 : a.vec[(size_t)a.pa[idx] * S + sm];
                const double xb = a.pb[idx] == M ? 1.0 : a.vec[(size_t)a.pb[idx] * S + sm];
                corr[e] += xa * xb;
            }
        }
    }
    double* rec = a.partial + (size_t)l * a.NS;
    n = wave_sum_i32(n);
    sg = wave_sum_f64(sg);
    sgg = wave_sum_f64(sgg);
    rx = wave_sum_f64(rx);
    rxx = wave_sum_f64(rxx);
    ry = wave_sum_f64(ry);
    ryy = wave_sum_f64(ryy);
    rxy = wave_sum_f64(rxy);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fmin(xmin, __shfl_xor(xmin, o, WAVE));
        xmax = fmax(xmax, __shfl_xor(xmax, o, WAVE));
    }
    if (lane == 0) {
        rec[0] = (double)n;
        rec[1] = sg;
        rec[2] = sgg;
        rec[a.NS - 1] = 0.0;
        double* ls = q.locus_sums + (size_t)l * TRK_ADL_COLS;
        ls[0] = rx; ls[1] = rxx; ls[2] = ry; ls[3] = ryy; ls[4] = rxy; ls[5] = 2.0 * n; ls[6] = xmin; ls[7] = xmax;
    }
    for (int k = 0; k < M; ++k) {
        const double t = wave_sum_f64(sgv[k]);
        if (lane == 0) rec[3 + k] = t;
    }
    for (int e = 0; e < AS_E; ++e) {
        const int idx = lane + e * WAVE;
        if (idx < a.NC) rec[3 + M + idx] = corr[e];
    }
    // ---- pass 2: class-major --------------------------------------------------------------------
    int first = 0;
    for (int u = 0; u < ncls; ++u) {
        int last = first;
        while (last < A && dcl[perm[last]] == u) ++last;
        double sd = 0.0, sdd = 0.0, sx = 0.0, sxd = 0.0;
        for (int s = lane; s < S; s += WAVE) {
            if (a.sample_in && !a.sample_in[s]) continue;
            const int16_t* cell = a.b.gt + ((int64_t)l * S + s) * P;
            bool miss = false;
            for (int p = 0; p < pl; ++p) miss |= cell[p] == -1;
            if (miss) continue;
            const float* ap[2] = {q.d.ap1 + ((int64_t)l * S + s) * Kc, q.d.ap2 + ((int64_t)l * S + s) * Kc};
            for (int p = 0; p < 2; ++p) {
                double d = 0.0;
                for (int i = first; i < last; ++i) {
                    const int al = perm[i];
                    d += (double)(al == 0 ? fmaxf(0.f, 1.f - np_sum_f32(ap[p], A - 1)) : ap[p][al - 1]);
                }
                const int al = p < pl ? cell[p] : -2;
                const double x = (al >= 0 && al < A && bcl[al] == u) ? 1.0 : 0.0;
                sd += d;
                sdd += d * d;
                sx += x;
                sxd += x * d;
            }
        }
        sd = wave_sum_f64(sd);
        sdd = wave_sum_f64(sdd);
        sx = wave_sum_f64(sx);
        sxd = wave_sum_f64(sxd);
        if (lane == 0) {
            double* cs = q.class_sums + (size_t)(off + u) * TRK_ADC_COLS;
            cs[0] = sd; cs[1] = sdd; cs[2] = sx; cs[3] = sxd;
        }
        first = last;
    }
}

// -------------------------------------------------------------------------------------------
// per-sample dosages (TRRecord.GetDosages, tr_harmonizer.py:1098-1208): one thread per call
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dosages(trk_batch b, const double* __restrict__ allele_len, int type,
                                                 const float* __restrict__ ap1, const float* __restrict__ ap2, int Kc,
                                                 float* __restrict__ out, int32_t* __restrict__ locus_err) {
    const int l = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int S = b.n_samples, P = b.ploidy;
    if (s >= S) return;
    const int pl = b.locus_ploidy ? b.locus_ploidy[l] : P;
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    double lmin = allele_len[off], lmax = allele_len[off], amax = -INFINITY;
    for (int i = 1; i < A; ++i) {
        const double v = allele_len[off + i];
        lmin = fmin(lmin, v);
        lmax = fmax(lmax, v);
        amax = fmax(amax, v);
    }
    const bool norm = type == TRK_DOS_BESTGUESS_NORM || type == TRK_DOS_BEAGLEAP_NORM;
    int err = 0;
    float unnorm;
    if (type == TRK_DOS_BESTGUESS || type == TRK_DOS_BESTGUESS_NORM) {
        const int16_t* cell = b.gt + ((int64_t)l * S + s) * P;
        double d = 0.0;
        for (int p = 0; p < pl; ++p) {
            const int al = cell[p];
            double v;
            if (al < 0) v = norm ? (double)NAN : 0.0;
            else v = al < A ? allele_len[off + al] : (double)NAN;
            d = p == 0 ? v : d + v;
        }
        unnorm = (float)d;
    } else {
        const float* ap[2] = {ap1 + ((int64_t)l * S + s) * Kc, ap2 + ((int64_t)l * S + s) * Kc};
        double h[2] = {0.0, 0.0};
        float refd[2];
        for (int p = 0; p < 2; ++p) {
            const float sum = np_sum_f32(ap[p], A - 1);
            if (sum > 1.1f) err |= 1;
            double dot = 0.0;
            for (int i = 0; i < A - 1; ++i) {
                if (ap[p][i] < 0.f) err |= 2;
                dot += (double)ap[p][i] * allele_len[off + 1 + i];
            }
            if (A > 1) h[p] = dot == dot ? fmin(fmax(dot, 0.0), amax) : dot;   // np.clip keeps nan
            const float om = 1.f - sum;
            const float ref = om == om ? fminf(fmaxf(om, 0.f), 1.f) : om;
            refd[p] = ref * (float)allele_len[off];
        }
        if (A > 1)
            unnorm = (float)(((h[0] + h[1]) + (double)refd[0]) + (double)refd[1]);
        else
            unnorm = refd[0] + refd[1];
    }
    float res = unnorm;
    if (norm) {
        if (lmin == lmax) {
            res = 0.f;
        } else {
            res = (unnorm - (float)(2.0 * lmin)) / (float)(lmax - lmin);
            if (res >= 2.1f || res <= -0.1f) err |= 4;
            if (res == res) res = fminf(fmaxf(res, 0.f), 2.f);   // nan stays nan (np.clip)
        }
    }
    out[(int64_t)l * S + s] = res;
    if (err) atomicOr(&locus_err[l], err);
}

// -------------------------------------------------------------------------------------------
// finaliser
// -------------------------------------------------------------------------------------------
// The frequencies of the ROUNDED length alleles in ascending order, one at a time
// (load_and_filter_genotypes.py:37-45 on top of GetAlleleFreqs, tr_harmonizer.py:1501-1540):
// classes of equal length were merged as integer counts; classes whose rounded lengths
// coincide are merged as float frequencies, added in ascending order.
struct AfStream {
    const int32_t* cc;          // class counts of this locus (entry c at cc[c * cs])
    const uint16_t* rcls;       // rounded class of each length class
    int ncls, c;
    double total;
    int cs = 1;
    __device__ void reset() { c = 0; }
    __device__ bool next(double& f) {
        while (c < ncls && cc[c * cs] == 0) ++c;
        if (c >= ncls) return false;
        const int r = rcls[c];
        f = (double)cc[c * cs] / total;
        ++c;
        while (c < ncls) {
            if (cc[c * cs] == 0) {
                ++c;
                continue;
            }
            if (rcls[c] != r) break;
            f += (double)cc[c * cs] / total;
            ++c;
        }
        return true;
    }
};

// numpy's pairwise summation (np.add.reduce on a contiguous float64 vector: blocks of 8
// accumulators up to 128 elements, recursive halving above) over the stream with element
// `skip` removed; elements are consumed strictly in order, so the recursion only needs the sizes
struct PwSum {
    AfStream* st;
    int idx, skip;
    __device__ double take() {
        double f = 0.0;
        for (;;) {
            st->next(f);
            if (idx++ != skip) return f;
        }
    }
    __device__ double leaf(int n) {
        if (n < 8) {
            double res = 0.0;
            for (int i = 0; i < n; ++i) res += take();
            return res;
        }
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = take();
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += take();
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += take();
        return res;
    }
    // numpy's recursion as a loop with NO arrays (an indexed stack would be scratch memory in every launch, touched or
    // not): the path to the current node is two bits per level (1: in the left child, 2: in the right one) from which
    // the node's size is replayed, the left halves' sums wait in twelve named registers.  Classes are 16-bit
    // (rlen_class), so n <= 65536 = 128 * 2^9: ten levels at most.
    static constexpr int PW_LEVELS = 12;
    __device__ double sum(int n) {
        if (n <= 128) return leaf(n);
        double left[PW_LEVELS];
#pragma unroll
        for (int k = 0; k < PW_LEVELS; ++k) left[k] = 0.0;
        uint32_t path = 0;   // stage of level d: (path >> 2 d) & 3
        int sp = 0;
        double ret = 0.0;
        while (sp >= 0) {
            int m = n;
            for (int d = 0; d < sp; ++d) {
                int h = m / 2;
                h -= h % 8;
                m = ((path >> (2 * d)) & 3u) == 1u ? h : m - h;
            }
            if (m <= 128 || sp >= PW_LEVELS - 1) {   // (the second condition cannot hold for n <= 65536)
                ret = leaf(m);
                --sp;
                continue;
            }
            const uint32_t stage = (path >> (2 * sp)) & 3u;
            if (stage == 0u) {
                path |= 1u << (2 * sp);
                ++sp;
                path &= ~(3u << (2 * sp));
            } else if (stage == 1u) {
#pragma unroll
                for (int k = 0; k < PW_LEVELS; ++k) left[k] = k == sp ? ret : left[k];
                path += 1u << (2 * sp);
                ++sp;
                path &= ~(3u << (2 * sp));
            } else {
                double lv = 0.0;
#pragma unroll
                for (int k = 0; k < PW_LEVELS; ++k) lv = k == sp ? left[k] : lv;
                ret = lv + ret;
                --sp;
            }
        }
        return ret;
    }
};

struct FinArgs {
    trk_batch b;
    const double* partial;
    const double* full;          // [NC]
    const int32_t* allele_count;
    int32_t* cc;                 // [sumA] scratch, zeroed
    const uint16_t* rlen_class;
    const double* allele_len;
    int32_t* locus_int;
    double* locus_f64;
    double cutoff;
    int M, NS, NC, nchunks;
    int cc_lds, cc_lds_off;      // cc_lds: class counts in thread-private LDS columns ([class][thread] int32), which start
                                 // cc_lds_off doubles x blockDim.x into the block's dynamic LDS; else the global scratch `cc`
    int dosage;                  // 1: no allele-frequency filters here (the caller applies them)
    int wave_regress;            // 1: the regression is left to k_assoc_regress_wave (wide designs)
};

// index of Gram entry (r, c), r <= c, rows 0..M (row M = ones), row-major upper triangle
__device__ __forceinline__ int gidx(int r, int c, int M) { return r * (M + 1) - r * (r - 1) / 2 + (c - r); }

__global__ __launch_bounds__(FIN_T) void k_assoc_finalize(const FinArgs a) {
    extern __shared__ double fin_lds[];  // [entries][blockDim.x], one column per thread
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.b.n_loci) return;
    const int M = a.M, L = a.b.n_loci;
    const int P = M + 1;  // design columns: ones, covariates 1..M-1, genotype (last)
#define LW(e) fin_lds[(size_t)(e) * blockDim.x + threadIdx.x]
    int32_t* li = a.locus_int + (size_t)l * TRK_AI_COLS;
    double* lf = a.locus_f64 + (size_t)l * TRK_AF_COLS;
    for (int i = 0; i < TRK_AI_COLS; ++i) li[i] = 0;
    for (int i = 0; i < TRK_AF_COLS; ++i) lf[i] = NAN;

    // ---- partial records of the sample chunks, in chunk order --------------------------------
    double n_d = 0.0, sg = 0.0, sgg = 0.0, n_bad = 0.0;
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const double* rec = a.partial + ((size_t)ch * L + l) * a.NS;
        n_d += rec[0];
        sg += rec[1];
        sgg += rec[2];
        n_bad += rec[a.NS - 1];
    }
    const int n = (int)n_d;
    li[TRK_AI_N_TESTED] = n;
    li[TRK_AI_N_BAD] = (int)n_bad;

    // ---- allele frequencies and the locus filters --------------------------------------------
    const int off = a.b.allele_off[l];
    const int A = a.b.allele_off[l + 1] - off;
    // (LDS columns: the read-add-write of a class count is an LDS round trip, not a global one -- this kernel is a
    // chain of latencies, one thread per locus at 1.5 waves per SIMD)
    const bool cc_lds = a.cc_lds && A <= a.b.max_alleles;   // (a locus beyond the stated maximum keeps the global scratch)
    int32_t* cc = cc_lds ? reinterpret_cast<int32_t*>(fin_lds + (size_t)a.cc_lds_off * blockDim.x) + threadIdx.x
                         : a.cc + off;
    const int cs = cc_lds ? (int)blockDim.x : 1;
    if (cc_lds)
        for (int i = 0; i < A; ++i) cc[i * cs] = 0;
    int64_t total = 0;
    for (int i = 0; i < A; ++i) {
        const int cnt = a.allele_count[off + i];
        if (cnt) {
            cc[a.b.len_class[off + i] * cs] += cnt;
            total += cnt;
        }
    }
    li[TRK_AI_N_HAPS] = (int)total;
    AfStream st{cc, a.rlen_class + off, A, 0, (double)total, cs};
    int R = 0, argmax = 0;
    double fmax = -1.0, f;
    st.reset();
    while (st.next(f)) {
        if (f > fmax) {
            fmax = f;
            argmax = R;
        }
        ++R;
    }
    li[TRK_AI_N_RALLELES] = R;
    int status = TRK_AS_OK;
    if (a.dosage) {
        if (M + 1 >= n) status = TRK_AS_N_COVARS;
    } else if (R == 0) {
        status = TRK_AS_NO_CALLED;
    } else if (R == 1) {
        status = TRK_AS_ONE_ALLELE;
    } else {
        st.reset();
        PwSum pw{&st, 0, argmax};
        const double s_af = 0.0 + pw.sum(R - 1);
        const double nonmajor = s_af * (double)n * 2.0;
        lf[TRK_AF_NONMAJOR] = nonmajor;
        if (nonmajor < a.cutoff)
            status = TRK_AS_NON_MAJOR;
        else if (M + 1 >= n)
            status = TRK_AS_N_COVARS;
    }
    if (status != TRK_AS_OK) {
        li[TRK_AI_STATUS] = status;
        return;
    }

    // ---- standardised genotype ---------------------------------------------------------------
    const double mean = sg / n_d;
    const double var = (sgg - sg * mean) / n_d;
    const int plv = a.b.locus_ploidy ? a.b.locus_ploidy[l] : a.b.ploidy;
    lf[TRK_AF_GT_MEAN] = mean + (double)plv * a.allele_len[off];  // undo the pivot (len(ref) per haplotype)
    if (!(var > 1e-12 * (sgg / n_d))) {  // constant up to rounding of the running sums
        li[TRK_AI_STATUS] = TRK_AS_ZERO_VARIANCE;
        return;
    }
    const double sd = sqrt(var);
    lf[TRK_AF_GT_STD] = sd;
    if (a.wave_regress) {
        lf[TRK_AF_COLS - 1] = mean;          // pivoted mean, for k_assoc_regress_wave
        li[TRK_AI_STATUS] = TRK_AS_OK;
        li[TRK_AI_RANK] = -1;                // "regression pending"
        return;
    }

    // ---- Gram matrix of the called samples: full - correction --------------------------------
    // LDS column: packed lower triangle of the P x P normal matrix (row-major), then rhs [P]
    const int ntri = P * (P + 1) / 2;
    auto G = [&](int r, int c) {  // Gram entry of vec rows r <= c (row M = ones)
        const int e = gidx(r, c, M);
        double v = a.full[e];
        for (int ch = 0; ch < a.nchunks; ++ch) v -= a.partial[((size_t)ch * L + l) * a.NS + 3 + M + e];
        return v;
    };
    auto row_of = [&](int j) { return j == 0 ? M : j; };  // design column j < M -> vec row
    for (int i = 0; i < M; ++i)
        for (int j = 0; j <= i; ++j) {
            int r = row_of(i), c = row_of(j);
            if (r > c) {
                int t = r;
                r = c;
                c = t;
            }
            LW(i * (i + 1) / 2 + j) = G(r, c);
        }
    const double sy = G(0, M), yy = G(0, 0);
    for (int j = 0; j < M; ++j) {
        // genotype row: sum g~ * column_j = (sum g*col_j - mean * sum col_j) / sd
        double sgc, sc;
        if (j == 0) {
            sgc = sg;
            sc = n_d;
        } else {
            sgc = 0.0;
            for (int ch = 0; ch < a.nchunks; ++ch) sgc += a.partial[((size_t)ch * L + l) * a.NS + 3 + j];
            sc = G(j, M);
        }
        LW(M * (M + 1) / 2 + j) = (sgc - mean * sc) / sd;
        LW(ntri + j) = j == 0 ? sy : G(0, j);  // rhs: column_j ' y
    }
    LW(M * (M + 1) / 2 + M) = n_d;  // g~ ' g~
    {
        double sgy = 0.0;
        for (int ch = 0; ch < a.nchunks; ++ch) sgy += a.partial[((size_t)ch * L + l) * a.NS + 3];
        LW(ntri + M) = (sgy - mean * sy) / sd;
    }

    // ---- Cholesky with forward substitution; dependent columns are dropped (pinv semantics) ---
    int rank = 0;
    double zz = 0.0, lpp = 0.0, zp = 0.0;
    bool last_dependent = false;
    for (int j = 0; j < P; ++j) {
        const double ajj = LW(j * (j + 1) / 2 + j);
        double d = ajj;
        for (int k = 0; k < j; ++k) {
            const double v = LW(j * (j + 1) / 2 + k);
            d -= v * v;
        }
        if (!(d > 1e-11 * ajj)) {  // column in the span of the previous ones
            for (int i = j; i < P; ++i) LW(i * (i + 1) / 2 + j) = 0.0;
            LW(ntri + j) = 0.0;
            if (j == P - 1) last_dependent = true;
            continue;
        }
        const double ljj = sqrt(d);
        LW(j * (j + 1) / 2 + j) = ljj;
        for (int i = j + 1; i < P; ++i) {
            double v = LW(i * (i + 1) / 2 + j);
            for (int k = 0; k < j; ++k) v -= LW(i * (i + 1) / 2 + k) * LW(j * (j + 1) / 2 + k);
            LW(i * (i + 1) / 2 + j) = v / ljj;
        }
        double z = LW(ntri + j);
        for (int k = 0; k < j; ++k) z -= LW(j * (j + 1) / 2 + k) * LW(ntri + k);
        z /= ljj;
        LW(ntri + j) = z;
        zz += z * z;
        ++rank;
        if (j == P - 1) {
            lpp = ljj;
            zp = z;
        }
    }
    li[TRK_AI_RANK] = rank;
    if (last_dependent) {
        li[TRK_AI_STATUS] = TRK_AS_COLLINEAR;
        return;
    }
    const double df = n_d - (double)rank;
    const double ssr = yy - zz;
    const double scale = ssr / df;
    const double coef = zp / lpp;
    const double se = sqrt(scale) / lpp;
    const double tval = coef / se;
    const double tss = yy - sy * sy / n_d;
    lf[TRK_AF_COEF] = coef;
    lf[TRK_AF_SE] = se;
    lf[TRK_AF_TVALUE] = tval;
    lf[TRK_AF_DF_RESID] = df;
    lf[TRK_AF_RSQUARED] = 1.0 - ssr / tss;
    lf[TRK_AF_PVAL] = trkmath::student_t_two_sided(tval, df);
    li[TRK_AI_STATUS] = TRK_AS_OK;
#undef LW
}

// -------------------------------------------------------------------------------------------
// regression of one locus per WAVE (wide designs: a 32 x 32 Cholesky per thread costs 6.5 ms per
// 100k loci).  Lane i owns row i of the normal matrix [ones, covariates 1..M-1, genotype] in an
// LDS tile; lane P carries the right-hand side as one more row (L_Pj = z_j, forward
// substitution for free).  Left-looking: column j needs row j (broadcast reads) and each lane's
// own row.  Same arithmetic and the same dropped-column rule as the thread-per-locus finaliser.
// -------------------------------------------------------------------------------------------
// R rows per lane (rows lane, lane + 64, ...): R = 1 up to 62 trait columns, R = 2 up to 126 (round 4; the triangle of a
// 128-row tile is 66 KB: two waves per workgroup then, blockDim.x / 64 says how many).
constexpr int RW_WAVES = 4;
template <int R>
__global__ __launch_bounds__(WAVE* RW_WAVES) void k_assoc_regress_wave(const FinArgs a) {
    extern __shared__ double rw_lds[];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    const int l = blockIdx.x * (int)(blockDim.x >> 6) + wid;
    if (l >= a.b.n_loci) return;
    int32_t* li = a.locus_int + (size_t)l * TRK_AI_COLS;
    double* lf = a.locus_f64 + (size_t)l * TRK_AF_COLS;
    if (li[TRK_AI_STATUS] != TRK_AS_OK || li[TRK_AI_RANK] != -1) return;  // filtered, or already regressed
    const int M = a.M, L = a.b.n_loci, P = M + 1;
    // rows packed as a lower triangle (row i: columns 0..i at i (i + 1) / 2; row P = the right-hand side): half the LDS
    // of a square tile -- twice the waves per CU at 63 rows -- and no common row stride (a stride of 64 doubles put
    // all 64 lanes on one bank)
    double* A = rw_lds + (size_t)wid * ((P + 1) * (P + 2) / 2);
#define AT(i, k) A[(i) * ((i) + 1) / 2 + (k)]
    auto psum = [&](int col) {
        double v = 0.0;
        for (int ch = 0; ch < a.nchunks; ++ch) v += a.partial[((size_t)ch * L + l) * a.NS + col];
        return v;
    };
    auto G = [&](int r, int c) {
        if (r > c) { const int t = r; r = c; c = t; }
        const int e = gidx(r, c, M);
        return a.full[e] - psum(3 + M + e);
    };
    const double n_d = (double)li[TRK_AI_N_TESTED];
    const double mean = lf[TRK_AF_COLS - 1], sd = lf[TRK_AF_GT_STD];
    const double sg = psum(1);
    const double sy = G(0, M), yy = G(0, 0);
    // ---- build.  Tile row i < M: row [ones, covariates 1..M-1][i] of Z'Z; row M the genotype row; row P the
    //      right-hand side.  The Gram entries (vector rows r <= c, row M = ones -> tile index 0, row 0 = outcome ->
    //      the right-hand side) are read in record order, 64 consecutive entries per step ------------------------
    {
        int r = 0, pos = lane;                 // entry e = lane, lane + 64, ...: row r, column r + pos
        while (r <= M && pos >= M + 1 - r) {
            pos -= M + 1 - r;
            ++r;
        }
        for (int e = lane; e < a.NC; e += WAVE) {
            const int c = r + pos;
            const double val = a.full[e] - psum(3 + M + e);
            const int ic = c == M ? 0 : c;
            if (r == 0) {
                if (c != 0) AT(P, ic) = val;   // (0, 0) is y'y
            } else {
                const int ir = r == M ? 0 : r;
                const int hi = ir > ic ? ir : ic, lo = ir > ic ? ic : ir;
                AT(hi, lo) = val;
            }
            pos += WAVE;
            while (r <= M && pos >= M + 1 - r) {
                pos -= M + 1 - r;
                ++r;
            }
        }
    }
    wave_fence();
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int i = lane + q * WAVE;
        if (i < M) {
            const double sgc = i == 0 ? sg : psum(3 + i);
            const double sc = i == 0 ? n_d : AT(i, 0);      // G(i, M): the ones column of row i
            AT(M, i) = (sgc - mean * sc) / sd;
        } else if (i == M) {
            AT(M, M) = n_d;
        } else if (i == P) {
            AT(P, M) = (psum(3) - mean * sy) / sd;
        }
    }
    wave_fence();
    // ---- left-looking Cholesky; rows j..P (row P = rhs) are updated for column j ---------------
    int rank = 0;
    bool last_dependent = false;
    double zz = 0.0;
    for (int j = 0; j < P; ++j) {
        double v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = lane + q * WAVE;
            v[q] = 0.0;
            if (i >= j && i <= P) {
                v[q] = AT(i, j);
                for (int k = 0; k < j; ++k) v[q] -= AT(i, k) * AT(j, k);
            }
        }
        const double ajj = AT(j, j);  // still the original diagonal entry
        double d = 0.0;               // row j's value: lane j % 64, slot j / 64
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const double t = __shfl(v[q], j & (WAVE - 1), WAVE);
            if ((j >> 6) == q) d = t;
        }
        wave_fence();
        if (!(d > 1e-11 * ajj)) {  // column in the span of the previous ones: dropped (pinv semantics)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int i = lane + q * WAVE;
                if (i >= j && i <= P) AT(i, j) = 0.0;
            }
            if (j == P - 1) last_dependent = true;
        } else {
            const double ljj = sqrt(d);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int i = lane + q * WAVE;
                if (i == j) AT(i, j) = ljj;
                else if (i > j && i <= P) AT(i, j) = v[q] / ljj;
            }
            ++rank;
        }
        wave_fence();
    }
    if (lane == 0) {
        li[TRK_AI_RANK] = rank;
        if (last_dependent) {
            li[TRK_AI_STATUS] = TRK_AS_COLLINEAR;
        } else {
            for (int j = 0; j < P; ++j) {
                const double z = AT(P, j);
                zz += z * z;
            }
            const double lpp = AT(M, M), zp = AT(P, M);
            const double df = n_d - (double)rank;
            const double ssr = yy - zz;
            const double scale = ssr / df;
            const double coef = zp / lpp;
            const double se = sqrt(scale) / lpp;
            const double tval = coef / se;
            const double tss = yy - sy * sy / n_d;
            lf[TRK_AF_COEF] = coef;
            lf[TRK_AF_SE] = se;
            lf[TRK_AF_TVALUE] = tval;
            lf[TRK_AF_DF_RESID] = df;
            lf[TRK_AF_RSQUARED] = 1.0 - ssr / tss;
            lf[TRK_AF_PVAL] = trkmath::student_t_two_sided(tval, df);
        }
        lf[TRK_AF_COLS - 1] = NAN;
    }
#undef AT
}


// -------------------------------------------------------------------------------------------
// wide designs (32..126 vector rows; 63 and more always, 32..62 outside the one-pass MFMA kernel's conditions): the
// rows are cut into groups of <= 15 and every PAIR of groups
// is scanned as a design of <= 30 rows by the kernels above; this kernel moves one pair's records
// (chunks summed) and its full Gram matrix to their places in the records of the whole design.
// Block l < L: locus l; block L: the full Gram matrix.
// -------------------------------------------------------------------------------------------
struct WideArgs {
    const double* sub_partial;  // [nchunks, L, ns]
    const double* sub_full;     // [nc]
    double* partial;            // [L, NS] records of the whole design (one chunk)
    double* full;               // [NC]
    int L, m, ns, nc, nchunks;  // the pair's design
    int M, NS;                  // the whole design
    int first;                  // 1: this pair also carries n, sum g, sum g^2, n_bad
    uint8_t row[AS_MAXV + 1];   // row of the whole design behind row r of the pair; row[m] = M (ones)
    uint8_t pa[AS_MAXNC], pb[AS_MAXNC];
};
__global__ __launch_bounds__(256) void k_assoc_wide_gather(const WideArgs a) {
    const int l = blockIdx.x;
    if (l == a.L) {
        for (int e = threadIdx.x; e < a.nc; e += blockDim.x)
            a.full[gidx(a.row[a.pa[e]], a.row[a.pb[e]], a.M)] = a.sub_full[e];
        return;
    }
    double* dst = a.partial + (size_t)l * a.NS;
    for (int e = threadIdx.x; e < a.ns; e += blockDim.x) {
        int d;
        if (e < 3) {
            if (!a.first) continue;
            d = e;
        } else if (e < 3 + a.m) {
            d = 3 + a.row[e - 3];
        } else if (e < 3 + a.m + a.nc) {
            const int g = e - 3 - a.m;
            d = 3 + a.M + gidx(a.row[a.pa[g]], a.row[a.pb[g]], a.M);
        } else {
            if (!a.first) continue;
            d = a.NS - 1;
        }
        double v = 0.0;
        for (int ch = 0; ch < a.nchunks; ++ch) v += a.sub_partial[((size_t)ch * a.L + l) * a.ns + e];
        dst[d] = v;
    }
}

}  // namespace

namespace trk {

struct AssocPlan {
    bool fast;
    int mfma_rt;   // 0: not the MFMA kernel; 1/2: 16-row tiles of the vector block
    int chunk, nchunks, wave_bytes, kshift, loci_per_wg, mv;
    size_t lds_bytes;
};

static AssocPlan assoc_plan(const trk_batch& b, int M) {
    AssocPlan p{};
    p.nchunks = 1;
    const int S = b.n_samples, Amax = b.max_alleles;
    p.mv = M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : M <= 8 ? 8 : 16;
    if (trk_opt("TRK_AS_GENERIC")) return p;
    if (b.ploidy != 2 || b.locus_ploidy || S <= 0 || (S % 4) != 0 || Amax <= 0 || Amax + 3 >= 65535 ||
        (reinterpret_cast<uintptr_t>(b.gt) & 15))
        return p;
    // one wave's private area: missing-call queue (1 KiB) + LUT (A+3 doubles) + histogram
    // (A+3 bins x K copies), the latter two within 3 KiB
    int kshift = 4;
    while (kshift >= 0 && (Amax + 3) * (8 + (4 << kshift)) > 3072) --kshift;
    if (kshift < 0) return p;
    p.kshift = kshift;
    // many covariates: the MFMA kernel (16 loci per workgroup step); TRK_AS_MFMA_MIN moves the threshold
    {
        int mfma_min = 5;
        if (const char* e = trk_opt("TRK_AS_MFMA_MIN")) mfma_min = atoi(e);
        if (M >= mfma_min && M + 1 <= 64 && !(M > AS_MAXV && trk_opt("TRK_AS_WIDE_PAIRS"))) {
            p.mfma_rt = (M + 1 + 15) / 16;      // 16-row tiles of [vectors..., 1]: up to four (62 trait columns + ones)
            p.wave_bytes = (((Amax + 3) * (8 + (4 << kshift)) + 15) & ~15);
            p.lds_bytes = (size_t)(2 * 16) * MF_SBR * 8 + (size_t)AS_WAVES * p.wave_bytes;
            p.nchunks = 1;
            p.chunk = 0;
            p.loci_per_wg = 0;
            p.fast = true;
            return p;
        }
    }
    if (M > 16) return p;  // the LDS-resident kernels are instantiated up to 16 vectors
    p.wave_bytes = AS_QCAP * 2 + (((Amax + 3) * (8 + (4 << kshift)) + 15) & ~15);
    if (p.mv <= 2) {   // k_assoc_scan_few: queue + 3 KiB of tables per wave, copies per locus; one copy must fit
        if ((Amax + 3) * 12 + 8 > AF_AREA) return p;
        p.wave_bytes = AF_QCAP * 4 + AF_AREA;
    }
    const size_t lds_total = 160 * 1024;
    const size_t rem = lds_total - (size_t)AS_WAVES * p.wave_bytes - 64 - 8 * (size_t)p.mv;
    int chunk = (int)(rem / (8 * (size_t)p.mv + 1));
    chunk &= ~255;
    if (chunk > 16384) chunk = 16384;  // 12-bit chunk index in the queue records
    if (chunk < 256) return p;
    const int s_pad = (S + 255) & ~255;
    if (chunk >= s_pad) {
        p.chunk = s_pad;
        p.nchunks = 1;
    } else {
        p.nchunks = (S + chunk - 1) / chunk;
        p.chunk = (((S + p.nchunks - 1) / p.nchunks) + 255) & ~255;  // balanced
    }
    p.lds_bytes = (size_t)p.mv * (p.chunk + 1) * 8 + p.chunk + (size_t)AS_WAVES * p.wave_bytes;
    // one workgroup per CU at a time (LDS): size the locus blocks so that the grid is a whole
    // number of rounds over the 256 CUs, ~48 loci (3 per wave) each
    {
        const int per_round = 256 / (p.nchunks < 256 ? p.nchunks : 256) > 0 ? 256 / p.nchunks : 1;
        int rounds = (b.n_loci + per_round * 48 - 1) / (per_round * 48);
        if (rounds < 1) rounds = 1;
        p.loci_per_wg = (b.n_loci + per_round * rounds - 1) / (per_round * rounds);
        if (p.loci_per_wg < 1) p.loci_per_wg = 1;
    }
    p.loci_per_wg = 0;  // persistent workgroups by default; TRK_AS_LB=n restores static blocks of n loci
    if (const char* e = trk_opt("TRK_AS_LB")) p.loci_per_wg = atoi(e) > 0 ? atoi(e) : 0;
    p.fast = true;
    return p;
}

constexpr int AW_GROUP = 15;   // rows per group of a wide design: a pair of groups + the ones row fits two MFMA tiles
static size_t assoc_ws_one(const trk_batch& b, int M);
static size_t wide_vec_bytes(const trk_batch& b) { return ((size_t)2 * AW_GROUP * b.n_samples * 8 + 255) & ~(size_t)255; }
static size_t wide_cnt_bytes(const trk_batch& b) { return ((size_t)b.n_alleles_total * 4 + 255) & ~(size_t)255; }

size_t assoc_workspace_bytes(const trk_batch& b, int M) {
    size_t bytes = (assoc_ws_one(b, M) + 255) & ~(size_t)255;
    if (M > TRK_ASSOC_MAX_VEC)   // + one pair's workspace, its vector block, a scratch histogram
        bytes += ((assoc_ws_one(b, 2 * AW_GROUP) + 255) & ~(size_t)255) + wide_vec_bytes(b) + wide_cnt_bytes(b);
    return bytes;
}

static size_t assoc_ws_one(const trk_batch& b, int M) {
    const AssocPlan p = assoc_plan(b, M);
    const int NC = (M + 1) * (M + 2) / 2, NS = 3 + M + NC + 1;
    size_t bytes = 1024 + (size_t)NC * 8;                             // work counters, full Gram
    bytes += (size_t)p.nchunks * b.n_loci * NS * 8;                   // partial records
    bytes += ((size_t)b.n_alleles_total * 4 + 7) & ~(size_t)7;        // class counts
    if (p.fast && p.mfma_rt) {                                        // missing-call bits, sample-major vector block
        const size_t nsteps = ((size_t)b.n_samples + MF_SB - 1) / MF_SB;
        bytes = (bytes + 255) & ~(size_t)255;
        bytes += (size_t)b.n_loci * nsteps * 4 * 8 + 256;
        bytes += nsteps * MF_SB * 16 * p.mfma_rt * 8;
    }
    return bytes + 64;
}

template <int MV, bool MASK>
static hipError_t launch_scan_tm(const AssocArgs& a, const AssocPlan& p, hipStream_t stream) {
    void (*kern)(const AssocArgs) = &k_assoc_scan<MV, MASK>;
    if constexpr (MV <= 2) kern = &k_assoc_scan_few<MV, MASK>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
    int gx;
    if (p.loci_per_wg) {
        gx = (a.b.n_loci + p.loci_per_wg - 1) / p.loci_per_wg;
    } else {
        gx = (a.n_cu + p.nchunks - 1) / p.nchunks;  // ~one workgroup per CU over all chunks
        const int max_useful = (a.b.n_loci + AS_WAVES - 1) / AS_WAVES;
        if (gx > max_useful) gx = max_useful;
        if (gx < 1) gx = 1;
    }
    dim3 grid(gx, p.nchunks), block(AS_THREADS);
    hipLaunchKernelGGL(kern, grid, block, p.lds_bytes, stream, a);
    return hipGetLastError();
}
template <int MV>
static hipError_t launch_scan_t(const AssocArgs& a, const AssocPlan& p, hipStream_t stream) {
    return a.sample_in ? launch_scan_tm<MV, true>(a, p, stream) : launch_scan_tm<MV, false>(a, p, stream);
}

static bool use_wave_regress(int M) {
    int min_m = 16;  // below, the thread-per-locus solve is faster (round 3: M = 15: 0.64 vs 0.69 ms; M = 16: 1.08 vs 0.71)
    if (const char* e = trk_opt("TRK_AS_WAVE_REGRESS_MIN")) min_m = atoi(e);
    return M >= min_m && M + 2 <= 2 * WAVE;
}

static hipError_t launch_regress_wave(const FinArgs& f, hipStream_t stream) {
    const int P = f.M + 1;
    if (P + 1 > 2 * WAVE) return hipErrorInvalidValue;
    const size_t tri = (size_t)((P + 1) * (P + 2) / 2) * 8;     // one locus's tile: lower triangle + the right-hand side
    int waves = RW_WAVES;
    while (waves > 1 && waves * tri > 144 * 1024) waves >>= 1;
    const size_t lds = (size_t)waves * tri;
    void (*kern)(FinArgs) = P + 1 <= WAVE ? k_assoc_regress_wave<1> : k_assoc_regress_wave<2>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(kern, dim3((f.b.n_loci + waves - 1) / waves), dim3(WAVE * waves), lds, stream, f);
    return hipGetLastError();
}

static void assoc_build(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out, void* workspace,
                        AssocPlan& p, AssocArgs& a, FinArgs& f, double*& full) {
    const int M = prm.n_vec;
    p = assoc_plan(b, M);
    a = AssocArgs{};
    a.b = b;
    a.vec = prm.vec;
    a.sample_in = prm.sample_in;
    a.allele_len = prm.allele_len;
    a.M = M;
    a.NC = (M + 1) * (M + 2) / 2;
    a.NS = 3 + M + a.NC + 1;
    a.chunk = p.chunk;
    a.nchunks = p.nchunks;
    a.loci_per_wg = p.loci_per_wg;
    a.wave_bytes = p.wave_bytes;
    a.kshift = p.kshift;
    int e = 0;
    for (int r = 0; r <= M && M <= AS_MAXV; ++r)
        for (int c = r; c <= M; ++c) {
            a.pa[e] = (uint8_t)r;
            a.pb[e] = (uint8_t)c;
            ++e;
        }
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    a.work_counter = reinterpret_cast<int*>(ws);
    ws += 1024;
    full = reinterpret_cast<double*>(ws);
    a.partial = reinterpret_cast<double*>(ws + (size_t)a.NC * 8);
    a.allele_count = out.allele_count;
    f = FinArgs{};
    f.b = b;
    f.partial = a.partial;
    f.full = full;
    f.allele_count = out.allele_count;
    f.cc = reinterpret_cast<int32_t*>(ws + (size_t)a.NC * 8 + (size_t)p.nchunks * b.n_loci * a.NS * 8);
    if (p.fast && p.mfma_rt) {
        a.nsteps = (b.n_samples + MF_SB - 1) / MF_SB;
        size_t o = (size_t)(reinterpret_cast<unsigned char*>(f.cc) - static_cast<unsigned char*>(workspace)) +
                   (((size_t)b.n_alleles_total * 4 + 7) & ~(size_t)7);
        o = (o + 255) & ~(size_t)255;
        a.missbits = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(workspace) + o);
        o += (size_t)b.n_loci * a.nsteps * 4 * 8 + 256;
        o &= ~(size_t)255;
        a.vect = reinterpret_cast<double*>(static_cast<unsigned char*>(workspace) + o);
    }
    f.rlen_class = prm.rlen_class;
    f.allele_len = prm.allele_len;
    f.locus_int = out.locus_int;
    f.locus_f64 = out.locus_f64;
    f.cutoff = prm.non_major_cutoff;
    f.M = M;
    f.NS = a.NS;
    f.NC = a.NC;
    f.nchunks = p.nchunks;
    f.wave_regress = (use_wave_regress(M) || M > AS_MAXV) ? 1 : 0;   // wide designs: always a wave per locus
}

// step 1 (cheap): zero the scratch, Gram matrix of the regression set
hipError_t launch_assoc_prepare(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                                void* workspace, hipStream_t stream) {
    AssocPlan p;
    AssocArgs a;
    FinArgs f;
    double* full;
    assoc_build(b, prm, out, workspace, p, a, f, full);
    hipError_t err;
    if ((err = hipMemsetAsync(a.work_counter, 0, 1024, stream)) != hipSuccess) return err;
    if (b.n_alleles_total > 0) {
        if ((err = hipMemsetAsync(f.cc, 0, (size_t)b.n_alleles_total * 4, stream)) != hipSuccess) return err;
        if (prm.n_vec > AS_MAXV && !(p.fast && p.mfma_rt)) return hipSuccess;   // pairs of row groups prepare themselves
        if (!p.fast || p.nchunks > 1)
            if ((err = hipMemsetAsync(out.allele_count, 0, (size_t)b.n_alleles_total * 4, stream)) != hipSuccess)
                return err;
    }
    if (prm.n_vec > AS_MAXV) {
        if (!(p.fast && p.mfma_rt)) return hipSuccess;
        hipLaunchKernelGGL(k_assoc_gram<true>, dim3(a.NC), dim3(256), 0, stream, a, full);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_assoc_gram<false>, dim3(a.NC), dim3(256), 0, stream, a, full);
    return hipGetLastError();
}

// wide designs: every pair of row groups through prepare + scan as a design of its own, gathered
// into the records of the whole design
static hipError_t launch_assoc_scan_wide(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                                         void* workspace, int n_cu, hipStream_t stream) {
    const int M = prm.n_vec, S = b.n_samples;
    AssocPlan p;
    AssocArgs a;
    FinArgs f;
    double* full;
    assoc_build(b, prm, out, workspace, p, a, f, full);
    unsigned char* sub_ws = static_cast<unsigned char*>(workspace) + ((assoc_ws_one(b, M) + 255) & ~(size_t)255);
    double* sub_vec = reinterpret_cast<double*>(sub_ws + ((assoc_ws_one(b, 2 * AW_GROUP) + 255) & ~(size_t)255));
    int32_t* sub_cnt = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(sub_vec) + wide_vec_bytes(b));
    const int ng = (M + AW_GROUP - 1) / AW_GROUP;
    hipError_t err;
    bool first = true;
    for (int ga = 0; ga < ng; ++ga)
        for (int gb = ga + 1; gb < ng; ++gb) {
            const int a0 = ga * AW_GROUP, na = AW_GROUP;
            const int b0 = gb * AW_GROUP, nb = (M - b0 < AW_GROUP) ? M - b0 : AW_GROUP;
            const int m = na + nb;
            if ((err = hipMemcpyAsync(sub_vec, prm.vec + (size_t)a0 * S, (size_t)na * S * 8, hipMemcpyDeviceToDevice,
                                      stream)) != hipSuccess)
                return err;
            if ((err = hipMemcpyAsync(sub_vec + (size_t)na * S, prm.vec + (size_t)b0 * S, (size_t)nb * S * 8,
                                      hipMemcpyDeviceToDevice, stream)) != hipSuccess)
                return err;
            trk_assoc_params sp = prm;
            sp.n_vec = m;
            sp.vec = sub_vec;
            trk_assoc_out so = out;
            if (!first) so.allele_count = sub_cnt;   // the histogram is the same in every pass: kept from the first
            if ((err = launch_assoc_prepare(b, sp, so, sub_ws, stream)) != hipSuccess) return err;
            if ((err = launch_assoc_scan(b, sp, so, sub_ws, n_cu, stream)) != hipSuccess) return err;
            AssocPlan q;
            AssocArgs sa;
            FinArgs sf;
            double* sfull;
            assoc_build(b, sp, so, sub_ws, q, sa, sf, sfull);
            WideArgs w{};
            w.sub_partial = sa.partial;
            w.sub_full = sfull;
            w.partial = a.partial;
            w.full = full;
            w.L = b.n_loci;
            w.m = m;
            w.ns = sa.NS;
            w.nc = sa.NC;
            w.nchunks = q.nchunks;
            w.M = M;
            w.NS = a.NS;
            w.first = first ? 1 : 0;
            for (int r = 0; r < na; ++r) w.row[r] = (uint8_t)(a0 + r);
            for (int r = 0; r < nb; ++r) w.row[na + r] = (uint8_t)(b0 + r);
            w.row[m] = (uint8_t)M;
            for (int e = 0; e < sa.NC; ++e) {
                w.pa[e] = sa.pa[e];
                w.pb[e] = sa.pb[e];
            }
            hipLaunchKernelGGL(k_assoc_wide_gather, dim3(b.n_loci + 1), dim3(256), 0, stream, w);
            if ((err = hipGetLastError()) != hipSuccess) return err;
            first = false;
        }
    return hipSuccess;
}

// step 2: the streaming pass over the genotype tensor
hipError_t launch_assoc_scan(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                             void* workspace, int n_cu, hipStream_t stream) {
    AssocPlan p;
    AssocArgs a;
    FinArgs f;
    double* full;
    assoc_build(b, prm, out, workspace, p, a, f, full);
    // wide designs (32-62 trait columns): three or four 16-row tiles in ONE pass of the MFMA kernel; batches outside
    // the streaming conditions (and TRK_AS_WIDE_PAIRS=1) go pair of row groups by pair of row groups
    if (prm.n_vec > AS_MAXV && !(p.fast && p.mfma_rt))
        return b.n_loci ? launch_assoc_scan_wide(b, prm, out, workspace, n_cu, stream) : hipSuccess;
    a.n_cu = n_cu > 0 ? n_cu : 256;
    if (b.n_loci == 0) return hipSuccess;
    if (p.fast && p.mfma_rt) {
        int gx = a.n_cu;
        const int tiles = (b.n_loci + 15) / 16;
        if (gx > tiles) gx = tiles;
        hipError_t e;
#define TRK_MFMA_LAUNCH(RT_, MASK_)                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assoc_scan_mfma<RT_, MASK_>),                       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);                       \
    if (e != hipSuccess) return e;                                                                               \
    hipLaunchKernelGGL((k_assoc_scan_mfma<RT_, MASK_>), dim3(gx), dim3(AS_THREADS), p.lds_bytes, stream, a);
        hipLaunchKernelGGL(k_assoc_vect, dim3(a.n_cu * 4), dim3(256), 0, stream, a, 16 * p.mfma_rt);
        if (p.mfma_rt == 1) {
            if (a.sample_in) { TRK_MFMA_LAUNCH(1, true) } else { TRK_MFMA_LAUNCH(1, false) }
        } else if (p.mfma_rt == 2) {
            if (a.sample_in) { TRK_MFMA_LAUNCH(2, true) } else { TRK_MFMA_LAUNCH(2, false) }
        } else if (p.mfma_rt == 3) {
            if (a.sample_in) { TRK_MFMA_LAUNCH(3, true) } else { TRK_MFMA_LAUNCH(3, false) }
        } else {
            if (a.sample_in) { TRK_MFMA_LAUNCH(4, true) } else { TRK_MFMA_LAUNCH(4, false) }
        }
#undef TRK_MFMA_LAUNCH
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // the Gram correction of every locus's missing calls, from the bits the scan left
        const size_t gm_lds = (size_t)GM_WAVES * GM_LIST * 4;
        const unsigned gm_grid = (unsigned)((b.n_loci + GM_WAVES - 1) / GM_WAVES);
        if (p.mfma_rt == 1) hipLaunchKernelGGL(k_assoc_gram_miss<1>, dim3(gm_grid), dim3(WAVE * GM_WAVES), gm_lds, stream, a);
        else if (p.mfma_rt == 2) hipLaunchKernelGGL(k_assoc_gram_miss<2>, dim3(gm_grid), dim3(WAVE * GM_WAVES), gm_lds, stream, a);
        else if (p.mfma_rt == 3) hipLaunchKernelGGL(k_assoc_gram_miss<3>, dim3(gm_grid), dim3(WAVE * GM_WAVES), gm_lds, stream, a);
        else hipLaunchKernelGGL(k_assoc_gram_miss<4>, dim3(gm_grid), dim3(WAVE * GM_WAVES), gm_lds, stream, a);
        return hipGetLastError();
    }
    if (p.fast) {
        switch (p.mv) {
            case 1: return launch_scan_t<1>(a, p, stream);
            case 2: return launch_scan_t<2>(a, p, stream);
            case 4: return launch_scan_t<4>(a, p, stream);
            case 8: return launch_scan_t<8>(a, p, stream);
            default: return launch_scan_t<16>(a, p, stream);
        }
    }
    hipLaunchKernelGGL(k_assoc_scan_any, dim3((b.n_loci + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// step 3: filters + regression per locus
hipError_t launch_assoc_finalize(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                                 void* workspace, hipStream_t stream) {
    AssocPlan p;
    AssocArgs a;
    FinArgs f;
    double* full;
    assoc_build(b, prm, out, workspace, p, a, f, full);
    if (b.n_loci == 0) return hipSuccess;
    const int P = prm.n_vec + 1;
    int fin_t = FIN_T;
    while (fin_t > 8 && !f.wave_regress && (size_t)(P * (P + 1) / 2 + P) * fin_t * 8 > 150 * 1024) fin_t >>= 1;
    size_t fin_lds = f.wave_regress ? 0 : (size_t)(P * (P + 1) / 2 + P) * fin_t * 8;   // the normal matrix lives in k_assoc_regress_wave
    f.cc_lds = f.cc_lds_off = 0;
    if (b.max_alleles > 0 && fin_lds + (size_t)b.max_alleles * fin_t * 4 <= 64 * 1024) {   // class counts in LDS columns
        f.cc_lds = 1;
        f.cc_lds_off = f.wave_regress ? 0 : P * (P + 1) / 2 + P;
        fin_lds += (size_t)b.max_alleles * fin_t * 4;
    }
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assoc_finalize),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_assoc_finalize, dim3((b.n_loci + fin_t - 1) / fin_t), dim3(fin_t), fin_lds, stream, f);
    if ((err = hipGetLastError()) != hipSuccess) return err;
    return f.wave_regress ? launch_regress_wave(f, stream) : hipSuccess;
}

// one pass of the dosage kernels over the batch for the design `prm` (at most AS_MAXV rows): records into `workspace`
static hipError_t dosage_pass(const trk_batch& b, const trk_batch& bb, const trk_assoc_params& prm, const trk_assoc_dosage& dos,
                              const trk_assoc_out& out, double* class_sums, double* locus_sums, void* workspace,
                              hipStream_t stream) {
    AssocPlan p;
    AssocArgs a;
    FinArgs f;
    double* full;
    assoc_build(bb, prm, out, workspace, p, a, f, full);
    a.b = b;
    hipError_t err;
    if ((err = hipMemsetAsync(a.work_counter, 0, 1024, stream)) != hipSuccess) return err;
    if (b.n_alleles_total > 0) {
        if ((err = hipMemsetAsync(f.cc, 0, (size_t)b.n_alleles_total * 4, stream)) != hipSuccess) return err;
        if ((err = hipMemsetAsync(out.allele_count, 0, (size_t)b.n_alleles_total * 4, stream)) != hipSuccess) return err;
        if ((err = hipMemsetAsync(class_sums, 0, (size_t)b.n_alleles_total * TRK_ADC_COLS * 8, stream)) != hipSuccess)
            return err;
    }
    if (b.n_loci == 0) return hipSuccess;
    hipLaunchKernelGGL(k_assoc_gram<false>, dim3(a.NC), dim3(256), 0, stream, a, full);
    DosArgs q{dos, class_sums, locus_sums};
    // few alleles everywhere (and diploid, which Beagle output is): the single-pass kernel
    if (b.max_alleles > 0 && b.max_alleles <= DQ_A && b.ploidy == 2 && !b.locus_ploidy && !trk_opt("TRK_AS_DOSAGE_GENERIC"))
    {
        const int cmax = b.max_alleles;   // classes <= alleles
        hipLaunchKernelGGL(k_assoc_dosage_small, dim3((b.n_loci + 3) / 4), dim3(256), (size_t)4 * cmax * 4 * WAVE * 8, stream, a,
                           q, cmax);
    }
    else
        hipLaunchKernelGGL(k_assoc_dosage, dim3((b.n_loci + 3) / 4), dim3(256), 0, stream, a, q);
    return hipGetLastError();
}

hipError_t launch_assoc_dosage(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_dosage& dos,
                               const trk_assoc_out& out, double* class_sums, double* locus_sums, void* workspace,
                               hipStream_t stream) {
    AssocPlan p;
    AssocArgs a;
    FinArgs f;
    double* full;
    trk_batch bb = b;
    bb.max_alleles = 0;  // plan: the generic record layout (one chunk, no LDS-resident vectors)
    assoc_build(bb, prm, out, workspace, p, a, f, full);
    f.b = b;
    hipError_t err;
    const int M = prm.n_vec, S = b.n_samples;
    if (M <= AS_MAXV) {
        if ((err = dosage_pass(b, bb, prm, dos, out, class_sums, locus_sums, workspace, stream)) != hipSuccess) return err;
    } else {
        // wide designs (round 4): pair of 15-row groups by pair through the same kernels, as launch_assoc_scan_wide
        // does for the genotype scan (the dosages, the class and locus sums are the same in every pass)
        unsigned char* sub_ws = static_cast<unsigned char*>(workspace) + ((assoc_ws_one(bb, M) + 255) & ~(size_t)255);
        double* sub_vec = reinterpret_cast<double*>(sub_ws + ((assoc_ws_one(bb, 2 * AW_GROUP) + 255) & ~(size_t)255));
        int32_t* sub_cnt = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(sub_vec) + wide_vec_bytes(bb));
        const int ng = (M + AW_GROUP - 1) / AW_GROUP;
        bool first = true;
        for (int ga = 0; ga < ng; ++ga)
            for (int gb = ga + 1; gb < ng; ++gb) {
                const int a0 = ga * AW_GROUP, na = AW_GROUP;
                const int b0 = gb * AW_GROUP, nb = (M - b0 < AW_GROUP) ? M - b0 : AW_GROUP;
                const int m = na + nb;
                if ((err = hipMemcpyAsync(sub_vec, prm.vec + (size_t)a0 * S, (size_t)na * S * 8, hipMemcpyDeviceToDevice,
                                          stream)) != hipSuccess)
                    return err;
                if ((err = hipMemcpyAsync(sub_vec + (size_t)na * S, prm.vec + (size_t)b0 * S, (size_t)nb * S * 8,
                                          hipMemcpyDeviceToDevice, stream)) != hipSuccess)
                    return err;
                trk_assoc_params sp = prm;
                sp.n_vec = m;
                sp.vec = sub_vec;
                trk_assoc_out so = out;
                if (!first) so.allele_count = sub_cnt;
                if ((err = dosage_pass(b, bb, sp, dos, so, class_sums, locus_sums, sub_ws, stream)) != hipSuccess) return err;
                if (b.n_loci == 0) continue;
                AssocPlan q;
                AssocArgs sa;
                FinArgs sf;
                double* sfull;
                assoc_build(bb, sp, so, sub_ws, q, sa, sf, sfull);
                WideArgs w{};
                w.sub_partial = sa.partial;
                w.sub_full = sfull;
                w.partial = a.partial;
                w.full = full;
                w.L = b.n_loci;
                w.m = m;
                w.ns = sa.NS;
                w.nc = sa.NC;
                w.nchunks = q.nchunks;
                w.M = M;
                w.NS = a.NS;
                w.first = first ? 1 : 0;
                for (int r = 0; r < na; ++r) w.row[r] = (uint8_t)(a0 + r);
                for (int r = 0; r < nb; ++r) w.row[na + r] = (uint8_t)(b0 + r);
                w.row[m] = (uint8_t)M;
                for (int e = 0; e < sa.NC; ++e) {
                    w.pa[e] = sa.pa[e];
                    w.pb[e] = sa.pb[e];
                }
                hipLaunchKernelGGL(k_assoc_wide_gather, dim3(b.n_loci + 1), dim3(256), 0, stream, w);
                if ((err = hipGetLastError()) != hipSuccess) return err;
                first = false;
            }
    }
    if (b.n_loci == 0) return hipSuccess;
    f.dosage = 1;
    f.cc_lds = f.cc_lds_off = 0;
    const int P = M + 1;
    int fin_t = FIN_T;
    while (fin_t > 8 && !f.wave_regress && (size_t)(P * (P + 1) / 2 + P) * fin_t * 8 > 150 * 1024) fin_t >>= 1;
    const size_t fin_lds = f.wave_regress ? 0 : (size_t)(P * (P + 1) / 2 + P) * fin_t * 8;   // (the normal matrix lives in k_assoc_regress_wave)
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assoc_finalize),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds)) != hipSuccess)
        return err;
    hipLaunchKernelGGL(k_assoc_finalize, dim3((b.n_loci + fin_t - 1) / fin_t), dim3(fin_t), fin_lds, stream, f);
    if ((err = hipGetLastError()) != hipSuccess) return err;
    return f.wave_regress ? launch_regress_wave(f, stream) : hipSuccess;
}

hipError_t launch_dosages(const trk_batch& b, const double* allele_len, int type, const float* ap1, const float* ap2,
                          int n_alt_cols, float* out, int32_t* locus_err, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(locus_err, 0, (size_t)b.n_loci * 4, stream);
    if (err != hipSuccess || b.n_loci == 0 || b.n_samples == 0) return err;
    hipLaunchKernelGGL(k_dosages, dim3((b.n_samples + 255) / 256, b.n_loci), dim3(256), 0, stream, b, allele_len, type,
                       ap1, ap2, n_alt_cols, out, locus_err);
    return hipGetLastError();
}

}  // namespace trk
