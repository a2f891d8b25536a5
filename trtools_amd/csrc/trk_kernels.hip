// trk_kernels.hip -- gfx950 (MI355X / CDNA4) device code of libtrk.
//
// The hot path is HBM-bound integer work: stream the [L, S, P] int16 genotype
// tensor once, histogram allele indices per locus, fold a handful of per-row
// predicates, and (dumpSTR) evaluate the call-level filter predicates on the
// FORMAT planes.  No MFMA: nothing here is a contraction.  What matters on
// CDNA4 is 16-byte-per-lane coalesced streaming, enough loads in flight per CU,
// conflict-free LDS histogram updates and no atomics on the streaming path.
//
// Kernels
//   k_locus_count      one 64-lane wavefront per locus row (no __syncthreads on
//                      the streaming path; waves of a workgroup are independent).
//                      Allele histogram in LDS with K bank-private copies:
//                      lane i updates copy (i mod K) of bin b at word b*K + i%K,
//                      so the 32 lanes of a ds_add_u32 lane group always hit 32
//                      different banks no matter how skewed the allele
//                      frequencies are (a plain histogram serialises ~32-way on
//                      the major allele).
//   k_locus_finalize   one thread per (group, locus): O(A) float64 statistics in
//                      the reference's summation order + the exact binomial HWE
//                      test (trk_binom.h).
//   k_call_filter      column-owner tiling: a thread owns 4 consecutive samples
//                      and walks a block of loci, so the per-sample counters
//                      (dumpSTR sample_info) live in registers / thread-private
//                      LDS slots and are flushed once per block.
//   k_locus_filter     one thread per locus: filter bits + loc_info counters.
//   k_synth            counter-based synthetic genotype / FORMAT generator.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/trk.h"
#include "../../include/trk_test.h"
#include "trk_binom.h"
#include "trk_internal.h"

using trk::HweItem;

namespace {

constexpr int WAVE = 64;
constexpr int COUNT_WAVES_PER_WG = 4;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Whole-wave reductions with DPP row operations (six vector instructions, no LDS round trips; the shuffle form costs
// six ds_bpermute + their latencies).  Every lane must be active.  xor 1, xor 2, half-row mirror and row mirror leave
// the 16-lane row's result in each of its lanes; row_bcast:15 folds rows 0 -> 1 and 2 -> 3, row_bcast:31 rows 1 -> 3:
// lane 63 holds the wave's result.
#define TRK_DPP_STEP(op_, v_, ctrl_, rmask_, ident_) \
    v_ = op_(v_, __builtin_amdgcn_update_dpp(ident_, v_, ctrl_, rmask_, 0xf, false))
__device__ __forceinline__ int dpp_add(int a, int b) { return a + b; }
__device__ __forceinline__ int dpp_max(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int wave_sum(int v) {
    TRK_DPP_STEP(dpp_add, v, 0xB1, 0xf, 0);    // quad_perm [1,0,3,2]
    TRK_DPP_STEP(dpp_add, v, 0x4E, 0xf, 0);    // quad_perm [2,3,0,1]
    TRK_DPP_STEP(dpp_add, v, 0x141, 0xf, 0);   // row_half_mirror
    TRK_DPP_STEP(dpp_add, v, 0x140, 0xf, 0);   // row_mirror
    TRK_DPP_STEP(dpp_add, v, 0x142, 0xa, 0);   // row_bcast:15 into rows 1 and 3
    TRK_DPP_STEP(dpp_add, v, 0x143, 0xc, 0);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v) {
    TRK_DPP_STEP(dpp_max, v, 0xB1, 0xf, v);
    TRK_DPP_STEP(dpp_max, v, 0x4E, 0xf, v);
    TRK_DPP_STEP(dpp_max, v, 0x141, 0xf, v);
    TRK_DPP_STEP(dpp_max, v, 0x140, 0xf, v);
    TRK_DPP_STEP(dpp_max, v, 0x142, 0xa, v);
    TRK_DPP_STEP(dpp_max, v, 0x143, 0xc, v);
    return __builtin_amdgcn_readlane(v, 63);
}
#undef TRK_DPP_STEP
// order LDS traffic of one wavefront (the hardware executes a wave's DS
// operations in order; this stops the compiler from moving them)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// k_locus_count
// ---------------------------------------------------------------------------
// extra bins appended to each group's histogram in the grouped path
enum { XB_CALLED = 0, XB_LOW = 1, XB_HOML = 2, XB_HOMS = 3, XB_NSAMP = 4, XB_BAD = 5, XB_N = 6 };

struct RowCtx {
    uint32_t* hist;      // LDS, this wave's histogram region
    const uint32_t* lut; // LDS class LUT (lc | sc << 16) or nullptr
    const uint16_t* lc_g;  // global class tables (already offset to this locus)
    const uint16_t* sc_g;
    int A;
    int K;               // copies per bin (power of two)
    int kslot;           // lane & (K-1)
    int stride;          // bins per group = A + XB_N (grouped path)
    bool dup_len, dup_str;
    bool direct;         // histogram too large for LDS: global atomics
    int32_t* cnt_g;      // allele_count + off (group 0)
    int64_t sumA;        // group stride of allele_count
    int pl;              // ploidy of this locus
};

__device__ __forceinline__ void class_of(const RowCtx& c, int a, int& lc, int& sc) {
    if (c.lut) {
        uint32_t v = c.lut[a];
        lc = v & 0xffff;
        sc = v >> 16;
    } else {
        lc = c.lc_g[a];
        sc = c.sc_g[a];
    }
}

// One diploid call (two haplotypes), no sample groups: histogram to LDS,
// row predicates to registers.
__device__ __forceinline__ void cell_p2(const RowCtx& c, uint32_t w, int& n_called, int& n_low,
                                        int& n_hl, int& n_hs, int& n_bad) {
    int a0 = (int16_t)(w & 0xffffu);
    int a1 = (int16_t)(w >> 16);
    if (c.pl < 2) a1 = -3;  // haploid locus inside a P=2 batch: second column ignored
    bool miss = (a0 == -1) | (a1 == -1);
    bool low = (a0 == -2) | (a1 == -2);
    bool v0 = a0 >= 0, v1 = a1 >= 0;
    bool b0 = v0 & (a0 >= c.A), b1 = v1 & (a1 >= c.A);
    n_bad += (int)(b0 | b1);
    v0 &= !b0;
    v1 &= !b1;
    if (!c.direct) {
        if (v0 & v1 & (a0 == a1)) {
            atomicAdd(&c.hist[a0 * c.K + c.kslot], 2u);
        } else {
            if (v0) atomicAdd(&c.hist[a0 * c.K + c.kslot], 1u);
            if (v1) atomicAdd(&c.hist[a1 * c.K + c.kslot], 1u);
        }
    } else {
        if (v0) atomicAdd(&c.cnt_g[a0], 1);
        if (v1) atomicAdd(&c.cnt_g[a1], 1);
    }
    if (!miss) {
        n_called++;
        if (low) {
            n_low++;
        } else if (c.pl == 2 && v0 && v1) {
            bool same = a0 == a1;
            bool hl = same, hs = same;
            if (!same && (c.dup_len | c.dup_str)) {
                int l0, s0, l1, s1;
                class_of(c, a0, l0, s0);
                class_of(c, a1, l1, s1);
                hl = l0 == l1;
                hs = s0 == s1;
            }
            n_hl += hl;
            n_hs += hs;
        }
    }
}

// General call: `P` haplotypes in vals[], optional group bits.  Everything goes
// through the (LDS or global) bins; used for P != 2 and for stratified runs.
template <int PMAX>
__device__ __forceinline__ void cell_general(const RowCtx& c, const int* vals, int P, uint32_t gbits,
                                             int G, int32_t* li_g, int64_t li_gstride) {
    bool miss = false, low = false;
    int nbad = 0;
    int pl = c.pl;
    for (int j = 0; j < pl; ++j) {
        int a = vals[j];
        miss |= a == -1;
        low |= a == -2;
        nbad |= (a >= c.A);
    }
    bool hl = false, hs = false;
    bool called = !miss;
    if (called && !low && pl >= 2 && nbad == 0) {
        // sorted genotype tuple: gt[0] == gt[1]  <=>  the smallest class occurs twice
        int minl = 0x7fffffff, mins = 0x7fffffff, cl = 0, cs = 0;
        for (int j = 0; j < pl; ++j) {
            int l, s;
            class_of(c, vals[j], l, s);
            if (l < minl) { minl = l; cl = 1; } else if (l == minl) { cl++; }
            if (s < mins) { mins = s; cs = 1; } else if (s == mins) { cs++; }
        }
        hl = cl >= 2;
        hs = cs >= 2;
    }
    for (int g = 0; g < G; ++g) {
        if (!((gbits >> g) & 1u)) continue;
        if (!c.direct) {
            uint32_t* h = c.hist + (size_t)g * c.stride * c.K;
            for (int j = 0; j < pl; ++j) {
                int a = vals[j];
                if (a >= 0 && a < c.A) atomicAdd(&h[a * c.K + c.kslot], 1u);
            }
            uint32_t* x = h + c.A * c.K;
            atomicAdd(&x[XB_NSAMP * c.K + c.kslot], 1u);
            if (nbad) atomicAdd(&x[XB_BAD * c.K + c.kslot], (uint32_t)nbad);
            if (called) {
                atomicAdd(&x[XB_CALLED * c.K + c.kslot], 1u);
                if (low) atomicAdd(&x[XB_LOW * c.K + c.kslot], 1u);
                if (hl) atomicAdd(&x[XB_HOML * c.K + c.kslot], 1u);
                if (hs) atomicAdd(&x[XB_HOMS * c.K + c.kslot], 1u);
            }
        } else {
            int32_t* cg = c.cnt_g + (size_t)g * c.sumA;
            for (int j = 0; j < pl; ++j) {
                int a = vals[j];
                if (a >= 0 && a < c.A) atomicAdd(&cg[a], 1);
            }
            int32_t* li = li_g + g * li_gstride;
            atomicAdd(&li[TRK_LI_N_SAMPLES], 1);
            if (nbad) atomicAdd(&li[TRK_LI_N_BAD], nbad);
            if (called) {
                atomicAdd(&li[TRK_LI_N_CALLED], 1);
                if (low) atomicAdd(&li[TRK_LI_N_LOWPLOIDY], 1);
                if (hl) atomicAdd(&li[TRK_LI_N_HOM_LEN], 1);
                if (hs) atomicAdd(&li[TRK_LI_N_HOM_STR], 1);
            }
        }
    }
}

template <bool FAST2>  // FAST2: P == 2 and no sample groups
__global__ __launch_bounds__(WAVE* COUNT_WAVES_PER_WG) void k_locus_count(
    trk_batch b, int32_t* __restrict__ allele_count, int32_t* __restrict__ locus_int, int hist_entries,
    int lut_entries) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x >> 6;
    uint32_t* hist = lds + (size_t)wid * (hist_entries + lut_entries);
    uint32_t* lut = hist + hist_entries;
    const int total_waves = gridDim.x * COUNT_WAVES_PER_WG;
    const int L = b.n_loci, S = b.n_samples, P = b.ploidy;
    const int G = b.group_bits ? b.n_groups : 1;
    const int64_t sumA = b.n_alleles_total;

    for (int l = blockIdx.x * COUNT_WAVES_PER_WG + wid; l < L; l += total_waves) {
        const int off = b.allele_off[l];
        const int A = b.allele_off[l + 1] - off;
        RowCtx c;
        c.A = A;
        c.pl = b.locus_ploidy ? (int)b.locus_ploidy[l] : P;
        if (c.pl > P) c.pl = P;
        c.lc_g = b.len_class + off;
        c.sc_g = b.str_class + off;
        c.cnt_g = allele_count + off;
        c.sumA = sumA;
        c.hist = hist;
        // class LUT + duplicate detection (classes are dense ranks: a duplicate
        // exists iff max rank + 1 < A)
        const bool lut_lds = A <= lut_entries;
        int ml = 0, ms = 0;
        for (int a = lane; a < A; a += WAVE) {
            int lc = c.lc_g[a], sc = c.sc_g[a];
            if (lut_lds) lut[a] = (uint32_t)lc | ((uint32_t)sc << 16);
            ml = lc > ml ? lc : ml;
            ms = sc > ms ? sc : ms;
        }
        ml = wave_max(ml);
        ms = wave_max(ms);
        c.dup_len = ml + 1 < A;
        c.dup_str = ms + 1 < A;
        c.lut = lut_lds ? lut : nullptr;
        c.stride = FAST2 ? A : A + XB_N;
        const int bins = G * c.stride;
        int K = 32;
        while (K > 1 && bins * K > hist_entries) K >>= 1;
        c.direct = bins * K > hist_entries;
        c.K = K;
        c.kslot = lane & (K - 1);
        if (!c.direct)
            for (int i = lane; i < bins * K; i += WAVE) hist[i] = 0;
        wave_lds_fence();

        int n_called = 0, n_low = 0, n_hl = 0, n_hs = 0, n_bad = 0;
        const int64_t row0 = (int64_t)l * S;
        int32_t* li0 = locus_int + (int64_t)l * TRK_LI_COLS;
        const int64_t li_gstride = (int64_t)L * TRK_LI_COLS;
        if (FAST2) {
            // cells are 4-byte (2 x int16); stream 16-byte chunks of the global cell
            // array that cover [row0, row0 + S), masking cells outside the row
            const u32x4* g4 = reinterpret_cast<const u32x4*>(b.gt);
            const int64_t ch_first = row0 >> 2;
            const int64_t ch_last = (row0 + S - 1) >> 2;
            const int64_t row1 = row0 + S;
            constexpr int U = 4;
            for (int64_t ch = ch_first + lane; ch <= ch_last; ch += (int64_t)WAVE * U) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int64_t cc = ch + (int64_t)u * WAVE;
                    if (cc <= ch_last) v[u] = __builtin_nontemporal_load(&g4[cc]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int64_t cc = ch + (int64_t)u * WAVE;
                    if (cc > ch_last) break;
                    int64_t cell = cc << 2;
                    uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if (cell >= row0 && cell + 3 < row1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) cell_p2(c, w[j], n_called, n_low, n_hl, n_hs, n_bad);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cell + j >= row0 && cell + j < row1)
                                cell_p2(c, w[j], n_called, n_low, n_hl, n_hs, n_bad);
                    }
                }
            }
        } else {
            for (int s = lane; s < S; s += WAVE) {
                int vals[TRK_MAX_PLOIDY];
                const int16_t* p = b.gt + (row0 + s) * P;
                for (int j = 0; j < P && j < TRK_MAX_PLOIDY; ++j) vals[j] = p[j];
                uint32_t gb = b.group_bits ? b.group_bits[s] : 1u;
                cell_general<TRK_MAX_PLOIDY>(c, vals, P, gb, G, li0, li_gstride);
            }
        }
        wave_lds_fence();

        // fold the K copies of every bin; lane `bin` starts at copy (lane mod K) so
        // that the 32 lanes of a ds_read lane group touch 32 different banks
        if (!c.direct) {
            for (int bin = lane; bin < bins; bin += WAVE) {
                uint32_t s = 0;
                for (int k = 0; k < K; ++k) s += hist[bin * K + ((k + lane) & (K - 1))];
                int g = bin / c.stride;
                int r = bin - g * c.stride;
                if (r < A) {
                    allele_count[(int64_t)g * sumA + off + r] = (int32_t)s;
                } else {
                    int x = r - A;
                    int col = x == XB_CALLED ? TRK_LI_N_CALLED
                              : x == XB_LOW  ? TRK_LI_N_LOWPLOIDY
                              : x == XB_HOML ? TRK_LI_N_HOM_LEN
                              : x == XB_HOMS ? TRK_LI_N_HOM_STR
                              : x == XB_NSAMP ? TRK_LI_N_SAMPLES
                                              : TRK_LI_N_BAD;
                    li0[g * li_gstride + col] = (int32_t)s;
                }
            }
        }
        if (FAST2) {
            n_called = wave_sum(n_called);
            n_low = wave_sum(n_low);
            n_hl = wave_sum(n_hl);
            n_hs = wave_sum(n_hs);
            n_bad = wave_sum(n_bad);
            if (lane == 0) {
                li0[TRK_LI_N_CALLED] = n_called;
                li0[TRK_LI_N_LOWPLOIDY] = n_low;
                li0[TRK_LI_N_HOM_LEN] = n_hl;
                li0[TRK_LI_N_HOM_STR] = n_hs;
                li0[TRK_LI_N_BAD] = n_bad;
                li0[TRK_LI_N_SAMPLES] = S - b.n_pad_samples;
            }
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------
// k_locus_count_fast : the streaming fast path (P == 2, one sample group,
// rows 16-byte aligned, max_alleles known and small enough for LDS).
// Branch-free per call: invalid haplotypes (-1, -2, out of range) are steered
// into a trash bin instead of being predicated away, row predicates are folded
// into per-lane counters with add-with-carry.
// ---------------------------------------------------------------------------
template <bool DUP>
__device__ __forceinline__ void fast_cell(uint32_t w, int A, uint32_t* hist, const uint32_t* lut, int kshift,
                                          int kslot, int& n_miss, int& n_low, int& n_hom, int& n_hl, int& n_hs,
                                          int& n_bad) {
    const int a0 = (int)(int16_t)(w & 0xffffu);
    const int a1 = (int)w >> 16;
    const bool v0 = (unsigned)a0 < (unsigned)A;
    const bool v1 = (unsigned)a1 < (unsigned)A;
    const int i0 = v0 ? a0 : A;
    const int i1 = v1 ? a1 : A;
    atomicAdd(&hist[(i0 << kshift) + kslot], 1u);
    atomicAdd(&hist[(i1 << kshift) + kslot], 1u);
    const bool m = (a0 == -1) | (a1 == -1);
    const bool lo = ((a0 == -2) | (a1 == -2)) & !m;
    n_miss += m;
    n_low += lo;
    n_bad += (a0 > a1 ? a0 : a1) >= A;
    n_hom += (a0 == a1) & v0;
    if (DUP) {
        const uint32_t x = lut[i0] ^ lut[i1];  // lut[A] is a class no allele has
        const bool both = v0 & v1;
        n_hl += both & ((x & 0xffffu) == 0u);
        n_hs += both & ((x >> 16) == 0u);
    }
}

template <bool DUP, int U>
__device__ __forceinline__ void fast_row(const u32x4* __restrict__ row, int nchunks, int lane, int A,
                                         uint32_t* hist, const uint32_t* lut, int kshift, int kslot,
                                         uint32_t hap1_fix, int& n_miss, int& n_low, int& n_hom, int& n_hl,
                                         int& n_hs, int& n_bad) {
    int c = lane;
    // full groups of U chunks per lane
    for (; c + (U - 1) * WAVE < nchunks; c += U * WAVE) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&row[c + u * WAVE]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fast_cell<DUP>(hap1_fix ? (v[u][j] & 0xffffu) | hap1_fix : v[u][j], A, hist, lut, kshift, kslot, n_miss, n_low, n_hom, n_hl, n_hs,
                               n_bad);
        }
    }
    for (; c < nchunks; c += WAVE) {
        u32x4 v = __builtin_nontemporal_load(&row[c]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fast_cell<DUP>(hap1_fix ? (v[j] & 0xffffu) | hap1_fix : v[j], A, hist, lut, kshift, kslot, n_miss, n_low, n_hom, n_hl, n_hs, n_bad);
    }
}

// ---------------------------------------------------------------------------
// k_locus_count_v2 : same mapping as k_locus_count_fast with the per-call work cut to
// ~10 VALU + 2 LDS atomics.  The sentinels are histogrammed like alleles
// (packed 16-bit math: +2 then min with A+2 gives bin 0 = '-2', bin 1 = '-1',
// bins 2..A+1 = alleles, bin A+2 = out of range) and every row predicate is derived
// after the stream from those totals plus three cheap per-call facts:
//   eq    rows whose two bins are equal            (one compare)
//   c_xy  rows whose two haplotypes are BOTH sentinels, by combination (rare path)
// so that
//   rows with a '-1'      = hap(-1) - c(-1,-1)                    -> n_called = S - that
//   low-ploidy rows       = hap(-2) - c(-2,-2) - c(-1,-2) - c(-2,-1)
//   homozygous (by index) = eq - c(-1,-1) - c(-2,-2)
// ---------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Row streamer of the count kernels: U 16-byte chunks per lane in flight, the NEXT U already requested while the
// current ones are histogrammed (two register sets), and the first set requested by the caller at the very top of
// the kernel -- before the class LUT is built and the histogram zeroed -- so that the row's first bytes travel
// while the wave does its per-locus bookkeeping.  (Without this a wave alternates between waiting for its loads
// and issuing ALU/LDS work; on 1000-sample rows the whole row is 4 chunks per lane and the wait is most of the
// wave's life.)  Chunk indices past the row are clamped for the load and skipped by the consumer.
// g_gt_temporal (TRK_GT_TEMPORAL=1, experiments): genotype loads without the nontemporal hint, so that the tensor may
// stay in the 256 MiB Infinity Cache for a second reader that follows closely (count pass beside call-filter pass)
__device__ int g_gt_temporal = 0;
template <int LPL, int U>
__device__ __forceinline__ void row_fetch(const u32x4* __restrict__ row, int base, int sl, int last, u32x4 (&v)[U]) {
    const bool tmp = g_gt_temporal != 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = base + sl + u * LPL;
        const u32x4* p = &row[c < last ? c : last];
        v[u] = tmp ? *p : __builtin_nontemporal_load(p);
    }
}
template <int LPL, int U, typename Cell>
__device__ __forceinline__ void row_stream(const u32x4* __restrict__ row, int nchunks, int sl, u32x4 (&cur)[U],
                                           Cell&& cell) {
    const int last = nchunks > 0 ? nchunks - 1 : 0;
    for (int base = 0; base < nchunks; base += U * LPL) {
        u32x4 nxt[U];
        const bool more = base + U * LPL < nchunks;          // uniform over the wave
        if (more) row_fetch<LPL, U>(row, base + U * LPL, sl, last, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + sl + u * LPL < nchunks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = cur[u][j];
                    cell(w);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    }
}

// Whole-wave row streamer of k_locus_count_v2: full iterations (every lane live, no clamping) run two at a time with
// the two register sets swapping roles -- no copy of the prefetched set, one 64-bit address per iteration with the U
// loads at immediate offsets, no per-chunk liveness test; the row's last (partial) iterations go through the guarded
// form.  `cur` holds the row's first U chunks on entry (row_fetch at the top of the kernel).
template <int U, typename Cell>
__device__ __forceinline__ void row_stream_wave(const u32x4* __restrict__ row, int nchunks, int lane, u32x4 (&cur)[U],
                                                Cell&& cell) {
    constexpr int STEP = U * WAVE;
    const int nfull = nchunks / STEP;
    const u32x4* p = row + lane;
    int it = 0;
    for (; it + 3 <= nfull; it += 2) {   // iterations it, it + 1 and it + 2 are full
        u32x4 alt[U];
        const u32x4* p1 = p + (size_t)(it + 1) * STEP;
#pragma unroll
        for (int u = 0; u < U; ++u) alt[u] = __builtin_nontemporal_load(p1 + u * WAVE);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = cur[u][j];
                cell(w);
            }
        const u32x4* p2 = p + (size_t)(it + 2) * STEP;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(p2 + u * WAVE);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = alt[u][j];
                cell(w);
            }
    }
    const int last = nchunks > 0 ? nchunks - 1 : 0;
    for (int base = it * STEP; base < nchunks; base += STEP) {
        u32x4 nxt[U];
        const bool more = base + STEP < nchunks;          // uniform over the wave
        if (more) row_fetch<WAVE, U>(row, base + STEP, lane, last, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + lane + u * WAVE < nchunks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = cur[u][j];
                    cell(w);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    }
}

template <bool DUP>
__device__ __forceinline__ void v2_cell(uint32_t w, uint32_t amax2, uint32_t* hist, const uint32_t* lut,
                                        int kshift, int kslot, int combo_bin0, int& n_eq, int& n_hl, int& n_hs) {
    u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
    u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, amax2));
    const uint32_t t = __builtin_bit_cast(uint32_t, t2);
    const uint32_t lo = t & 0xffffu, hi = t >> 16;
    atomicAdd(&hist[(lo << kshift) + kslot], 1u);
    atomicAdd(&hist[(hi << kshift) + kslot], 1u);
    n_eq += lo == hi;
    if ((t & 0xfffefffeu) == 0u)  // both haplotypes are sentinels (a no-call): ~3 % of the rows
        atomicAdd(&hist[((combo_bin0 + (int)(lo + 2u * hi)) << kshift) + kslot], 1u);
    if (DUP) {
        const uint32_t x = lut[lo] ^ lut[hi];
        n_hl += (x & 0xffffu) == 0u;
        n_hs += (x >> 16) == 0u;
    }
}

// ---- the wave-per-locus cell in mask form (k_locus_count_v2) ----------------------------------------------------
// The SQ counters put k_locus_count_v2 at ~100 % VALU-busy with 27 vector instructions per 64 calls
// (profiles/r02_notes.md section 7): the kernel is bound by vector issue at 5.7 TB/s.  Per call it needs only the two
// histogram atomics per lane; everything else is a COUNT over the wave and can be a popcount of a lane mask, which
// costs one vector compare and scalar instructions:  equal bins (homozygous by index), the four sentinel pairs
// (t is one of four constants then), and with duplicate classes the LUT tests.  LDS byte addresses come straight from
// the packed 16-bit bins (v_mad_u32_u16, op_sel for the high half).
typedef __attribute__((address_space(3))) uint32_t* lds_u32p;
typedef __attribute__((address_space(3))) const uint32_t* lds_cu32p;
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
// lanes whose two 16-bit halves are equal
__device__ __forceinline__ uint64_t halves_equal(uint32_t t) {
    uint64_t m;
    asm("v_cmp_eq_u16_sdwa %0, %1, %1 src0_sel:WORD_0 src1_sel:WORD_1" : "=s"(m) : "v"(t));
    return m;
}
// x += (this lane's bit of the wave-wide mask): the mask goes in as the carry of an add-with-carry (one instruction;
// written as a select and an add the compiler emits two)
__device__ __forceinline__ void add_mask(uint32_t& x, uint64_t mask) {
    asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(x) : "s"(mask) : "vcc");
}

struct CountAcc {   // wave-uniform counters of one locus
    uint32_t n_eq = 0, c00 = 0, c10 = 0, c01 = 0, c11 = 0, n_hl = 0, n_hs = 0;
};
template <bool DUP>
__device__ __forceinline__ void v2m_cell(uint32_t w, uint32_t amax2, uint32_t hist_b, uint32_t kbytes, uint32_t lut_b,
                                         uint32_t four, CountAcc& c) {
    u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
    u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, amax2));
    const uint32_t t = __builtin_bit_cast(uint32_t, t2);
    __hip_atomic_fetch_add((lds_u32p)(uintptr_t)mad16_lo(t, kbytes, hist_b), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add((lds_u32p)(uintptr_t)mad16_hi(t, kbytes, hist_b), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    c.n_eq += (uint32_t)__popcll(halves_equal(t));
    // both haplotypes sentinels (bins 0 / 1): the four pairs, (-2,-2), lo -1 hi -2, lo -2 hi -1, (-1,-1)
    c.c00 += (uint32_t)__popcll(__ballot(t == 0x00000000u));
    c.c10 += (uint32_t)__popcll(__ballot(t == 0x00000001u));
    c.c01 += (uint32_t)__popcll(__ballot(t == 0x00010000u));
    c.c11 += (uint32_t)__popcll(__ballot(t == 0x00010001u));
    if (DUP) {
        const uint32_t x = *(lds_cu32p)(uintptr_t)mad16_lo(t, four, lut_b) ^ *(lds_cu32p)(uintptr_t)mad16_hi(t, four, lut_b);
        c.n_hl += (uint32_t)__popcll(__ballot((x & 0xffffu) == 0u));
        c.n_hs += (uint32_t)__popcll(__ballot(x < 0x10000u));
    }
}

// bins: 0 '-2', 1 '-1', 2..A+1 alleles, A+2 out of range, A+3.. A+6 sentinel pairs (lo + 2*hi)
// twin_delta != 0 (TRK_STATS_TWIN): every count is stored a second time, twin_ac / twin_li elements further on
template <int U>
__global__ __launch_bounds__(WAVE* COUNT_WAVES_PER_WG) void k_locus_count_v2(
    trk_batch b, int32_t* __restrict__ allele_count, int32_t* __restrict__ locus_int, int kshift,
    int wave_lds_words, int64_t twin_ac, int64_t twin_li) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x >> 6;
    const int l = blockIdx.x * COUNT_WAVES_PER_WG + wid;
    if (l >= b.n_loci) return;  // waves are independent: no workgroup barrier anywhere
    const int S = b.n_samples;
    const int64_t RS = b.row_stride ? b.row_stride : S;     // a column-range view: rows are RS samples apart
    const u32x4* row = reinterpret_cast<const u32x4*>(b.gt) + (((int64_t)l * RS) >> 2);
    const int nchunks = S >> 2;
    u32x4 cur[U];
    row_fetch<WAVE, U>(row, 0, lane, nchunks - 1, cur);   // the row's first chunks travel during the prologue
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    const int K = 1 << kshift;
    const int kslot = lane & (K - 1);
    const int nbins = A + 7;
    uint32_t* hist = lds + (size_t)wid * wave_lds_words;
    uint32_t* lut = hist + (nbins << kshift);  // indexed by BIN
    int ml = 0, ms = 0;
    for (int a = lane; a < A; a += WAVE) {
        int lc = b.len_class[off + a], sc = b.str_class[off + a];
        lut[a + 2] = (uint32_t)lc | ((uint32_t)sc << 16);
        ml = lc > ml ? lc : ml;
        ms = sc > ms ? sc : ms;
    }
    if (lane == 0) {  // sentinel bins: classes no allele has, distinct from each other
        lut[0] = 0xffffffffu;
        lut[1] = 0xfffefffeu;
        lut[A + 2] = 0xfffdfffdu;
    }
    ml = wave_max(ml);
    ms = wave_max(ms);
    const bool dup = (ml + 1 < A) | (ms + 1 < A);
    if (kshift >= 2) {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = lane; i < ((A + 3) << (kshift - 2)); i += WAVE) reinterpret_cast<u32x4*>(hist)[i] = z4;
    } else {
        for (int i = lane; i < ((A + 3) << kshift); i += WAVE) hist[i] = 0;
    }
    wave_lds_fence();

    CountAcc c;
    const uint32_t amax2 = (uint32_t)(A + 2) * 0x00010001u;
    const uint32_t hist_b = (uint32_t)(uintptr_t)(lds_u32p)hist + 4u * (uint32_t)kslot, kbytes = 4u << kshift;
    const uint32_t lut_b = (uint32_t)(uintptr_t)(lds_u32p)lut, four = 4u;
    if (dup)
        row_stream_wave<U>(row, nchunks, lane, cur, [&](uint32_t w) {
            v2m_cell<true>(w, amax2, hist_b, kbytes, lut_b, four, c);
        });
    else
        row_stream_wave<U>(row, nchunks, lane, cur, [&](uint32_t w) {
            v2m_cell<false>(w, amax2, hist_b, kbytes, lut_b, four, c);
        });
    wave_lds_fence();
    // fold the K copies of every bin: lanes = bins x parts, every part adds K / parts copies read 16 bytes at a time,
    // the parts of a bin are adjacent lanes
    uint32_t h_m2 = 0, h_m1 = 0, n_bad = 0;   // totals of the special bins, broadcast below
    if (kshift >= 2) {
        int pshift = kshift - 2;
        while (pshift > 0 && (nbins << pshift) > WAVE) --pshift;
        const int part = lane & ((1 << pshift) - 1);
        const int per4 = K >> (pshift + 2);   // 16-byte reads per part
        for (int bin0 = 0; bin0 < nbins; bin0 += WAVE >> pshift) {
            const int bin = bin0 + (lane >> pshift);
            uint32_t sum = 0;
            if (bin < A + 3) {                // (the sentinel-pair bins are not used by this kernel)
                const u32x4* hp = reinterpret_cast<const u32x4*>(hist + (bin << kshift)) + part * per4;
                for (int i = 0; i < per4; ++i) {
                    const u32x4 h = hp[i];
                    sum += (h.x + h.y) + (h.z + h.w);
                }
            }
            for (int o = (1 << pshift) >> 1; o > 0; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, WAVE);
            if (part == 0 && bin >= 2 && bin < A + 2) {
                allele_count[off + bin - 2] = (int32_t)sum;
                if (twin_ac) allele_count[twin_ac + off + bin - 2] = (int32_t)sum;
            }
            if (bin0 == 0) {                  // bins 0 and 1 sit in the first pass, at lanes 0 and 1 << pshift
                h_m2 = (uint32_t)__builtin_amdgcn_readlane((int)sum, 0);
                h_m1 = (uint32_t)__shfl((int)sum, 1 << pshift, WAVE);
            }
            const int lb = (A + 2 - bin0) << pshift;   // lane that holds the out-of-range bin in this pass
            if (lb >= 0 && lb < WAVE) n_bad = (uint32_t)__shfl((int)sum, lb, WAVE);
        }
    } else {
        for (int bin = lane; bin < A + 3; bin += WAVE) {
            uint32_t sum = 0;
            for (int k = 0; k < K; ++k) sum += hist[(bin << kshift) + k];
            if (bin >= 2 && bin < A + 2) {
                allele_count[off + bin - 2] = (int32_t)sum;
                if (twin_ac) allele_count[twin_ac + off + bin - 2] = (int32_t)sum;
            }
            hist[bin << kshift] = sum;
        }
        wave_lds_fence();
        h_m2 = hist[0 << kshift];
        h_m1 = hist[1 << kshift];
        n_bad = hist[(A + 2) << kshift];
    }
    if (lane == 0) {
        const int c00 = (int)c.c00, c10 = (int)c.c10, c01 = (int)c.c01, c11 = (int)c.c11;
        const int miss_rows = (int)h_m1 - c11;
        const int low_rows = (int)h_m2 - c00 - c10 - c01;
        const int hom_idx = (int)c.n_eq - c11 - c00;
        // the whole 48-byte row (the finaliser's columns zeroed): three 16-byte stores, no memset before the launch
        static_assert(TRK_LI_COLS == 12 && TRK_LI_N_CALLED == 0 && TRK_LI_N_HOM_STR == 3 && TRK_LI_N_BAD == 5 &&
                      TRK_LI_N_SAMPLES == 8, "row layout");
        const u32x4 r0 = {(uint32_t)(S - miss_rows), (uint32_t)low_rows,
                          (uint32_t)(dup ? (int)c.n_hl - c11 - c00 : hom_idx),
                          (uint32_t)(dup ? (int)c.n_hs - c11 - c00 : hom_idx)};
        const u32x4 r1 = {0u, n_bad, 0u, 0u};
        const u32x4 r2 = {(uint32_t)(S - b.n_pad_samples), 0u, 0u, 0u};
        for (int64_t tw = 0;; tw = twin_li) {
            u32x4* li0 = reinterpret_cast<u32x4*>(locus_int + tw + (int64_t)l * TRK_LI_COLS);
            li0[0] = r0;
            li0[1] = r1;
            li0[2] = r2;
            if (tw == twin_li) break;
        }
    }
    wave_lds_fence();
}

// ---------------------------------------------------------------------------
// k_locus_count_v3<R> : k_locus_count_v2 for SHORT rows -- R = 2 or 4 loci per wavefront, 64 / R lanes each.
// A 1000-sample row is 250 16-byte chunks: four loads per lane of a whole wave, against ~500 instructions of
// per-locus work that do not depend on the row length (class LUT, histogram zeroing and fold, reductions, the
// result row).  The v2 kernel is instruction-issue bound there (400k x 1k: 2.9 TB/s, 0.36 of the HBM peak;
// 100k x 10k: 5.8 TB/s).  With R loci side by side in one wave every one of those instructions serves R loci.
// Histogram rows stay bank-conflict free without more LDS per locus: a bin is one 32-word row, lane i adds at
// column i mod 32 -- with R = 2 each 32-lane group (= one locus) has its own rows, with R = 4 the two loci of a
// 32-lane group share rows and own 16 columns each (a ds_add_u32 is served in lane groups {0-31}, {32-63}).
// bins as in v2: 0 '-2', 1 '-1', 2..A+1 alleles, A+2 out of range, A+3..A+6 sentinel pairs.
// ---------------------------------------------------------------------------
template <int LPL>
__device__ __forceinline__ int seg_sum(int v) {
#pragma unroll
    for (int o = LPL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <int LPL>
__device__ __forceinline__ int seg_max(int v) {
#pragma unroll
    for (int o = LPL / 2; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, WAVE);
        v = t > v ? t : v;
    }
    return v;
}

// hcol_b: LDS byte address of this lane's histogram column (bins are 128 bytes apart), lut_b: of the locus's class LUT.
// Addresses straight from the packed bins (v_mad_u32_u16); per-lane counters (a wave holds R loci: no wave-wide counts).
template <bool DUP>
__device__ __forceinline__ void v3_cell(uint32_t w, uint32_t amax2, uint32_t hcol_b, uint32_t lut_b, uint32_t c128,
                                        uint32_t c4, uint32_t combo_b, uint32_t& n_eq, uint32_t& n_hl,
                                        uint32_t& n_hs) {
    u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
    u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, amax2));
    const uint32_t t = __builtin_bit_cast(uint32_t, t2);
    __hip_atomic_fetch_add((lds_u32p)(uintptr_t)mad16_lo(t, c128, hcol_b), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add((lds_u32p)(uintptr_t)mad16_hi(t, c128, hcol_b), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    add_mask(n_eq, halves_equal(t));
    if ((t & 0xfffefffeu) == 0u)   // both haplotypes sentinels: bin A + 3 + lo + 2 hi
        __hip_atomic_fetch_add((lds_u32p)(uintptr_t)(combo_b + (((t | (t >> 15)) & 3u) << 7)), 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    if (DUP) {
        const uint32_t x = *(lds_cu32p)(uintptr_t)mad16_lo(t, c4, lut_b) ^ *(lds_cu32p)(uintptr_t)mad16_hi(t, c4, lut_b);
        add_mask(n_hl, __ballot((x & 0xffffu) == 0u));
        add_mask(n_hs, __ballot(x < 0x10000u));
    }
}

// ---------------------------------------------------------------------------
// The finaliser as the count kernel's epilogue (small batches: BASELINE configs[1] is a latency chain of launches,
// the 40 MB stream itself is 5 us).  The LPL lanes of a locus share the work that k_locus_finalize does in one
// thread: the divisions and logarithms of the classes go side by side (class c on lane c mod LPL), and only the
// float64 SUMS -- whose order is the reference's (utils.py:139-296 iterate the sorted allele dict) -- stay serial:
// every sum is a chain  acc += X[c], c ascending,  over an LDS array X filled in parallel (an empty class holds +0.0:
// acc + 0.0 == acc), up to four chains side by side on the locus's first lanes.  Same operations on the same operands
// in the same order as mode_stats / k_locus_finalize: the results are the same bits.
// HWE tests go to FIXED slots (items[l] by length, items[L + l] by sequence; modes == 0: none) -- no counter to zero,
// no compaction: k_hwe_test_slots walks 2 L slots.
// ---------------------------------------------------------------------------
struct V3Fin {
    double* locus_f64;
    HweItem* items;
    double nalleles_thresh;
    int amax4;          // max_alleles rounded up to a multiple of four
    int fin_words;      // 32-bit words of one locus's scratch: 2 * amax4 (class counts) + 10 * amax4 (five f64 arrays)
};

// What one locus's lanes need for the cooperative finaliser (coop_finalize): counts by allele index and the
// alleles' classes (LDS or global), 12 * amax4 words of LDS scratch, the row predicates of the count pass.
struct CoopFin {
    const uint32_t* cnt;        // [A] called haplotypes per allele index
    const uint32_t* cls;        // [A] length class | sequence class << 16
    const double* cv;           // [A] class values (lengths), global
    uint32_t* scratch;          // LDS, 12 * amax4 words, 8-byte aligned, this locus's own
    int amax4;
    int row, n_rows;            // output row (g * L + l) and rows in all (slot addressing)
    int pl, ns_real;            // ploidy of the locus; samples of the group
    double nalleles_thresh;
    int32_t* locus_int;
    double* locus_f64;
    unsigned int* hwe_count;    // compact work list when not null, else fixed slots
    HweItem* items;
};

template <int LPL>
__device__ __forceinline__ void coop_finalize(const CoopFin& fa, bool live, bool any_dup, int sl, int lane0, int A,
                                              int n_called, int n_low, int n_homl, int n_homs, int n_bad) {
    const int A4 = (A + 3) & ~3, M4 = fa.amax4;
    int32_t* ccl = reinterpret_cast<int32_t*>(fa.scratch);
    int32_t* ccs = ccl + M4;
    double* Fl = reinterpret_cast<double*>(ccs + M4);   // f = n / total by class, alleles by length
    double* T1 = Fl + M4;                               // f^2, then -(pk ln pk), then f (v - mean)^2
    double* Fs = T1 + M4;                               // by sequence: f, then -(pk ln pk)
    double* T3 = Fs + M4;                               // by sequence f^2, then v f (the mean's terms)
    double* CV = T3 + M4;                               // class values (lengths)
    for (int a = sl; a < A4; a += LPL) {
        ccl[a] = 0;
        ccs[a] = 0;
    }
    wave_lds_fence();
    int tot = 0;
    for (int a = sl; a < A; a += LPL) {
        const int n = (int)fa.cnt[a];
        const uint32_t cls = fa.cls[a];
        __hip_atomic_fetch_add(&ccl[cls & 0xffffu], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&ccs[cls >> 16], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tot += n;
    }
    const int total = seg_sum<LPL>(tot);
    wave_lds_fence();
    const double nan = __builtin_nan("");
    const double ft = (double)total;      // float(sum(counts))   (tr_harmonizer.py:1539)
    const double* cvg = fa.cv;
    // pass 1 (parallel): f and f^2 per class; nalleles, last non-empty class, first most frequent class
    // the two partitions give the same class counts in the same order at most loci (then one set of sums serves
    // both, as in k_locus_finalize); equal only as multisets -- alleles sorted by sequence against by length --
    // means another summation order: both sets are taken
    int mism = n_homl != n_homs;
    for (int c = sl; c < A; c += LPL) mism |= ccl[c] != ccs[c];
    const bool same = seg_max<LPL>(mism) == 0;
    const bool both = __ballot(!same) != 0ull;
    int na_l = 0, na_s = 0, last = -1, bestn = 0;
    for (int c = sl; c < A4; c += LPL) {
        const int nl = c < A ? ccl[c] : 0, ns = c < A ? ccs[c] : 0;
        const double fl = nl ? (double)nl / ft : 0.0;
        Fl[c] = fl;
        T1[c] = fl * fl;
        CV[c] = c < A ? cvg[c] : 0.0;
        na_l += (nl != 0) & (fl >= fa.nalleles_thresh);     // statSTR.py:207
        if (nl) last = c;
        bestn = nl > bestn ? nl : bestn;
        if (both) {
            const double fs = ns ? (double)ns / ft : 0.0;
            Fs[c] = fs;
            T3[c] = fs * fs;
            na_s += (ns != 0) & (fs >= fa.nalleles_thresh);
        }
    }
    na_l = seg_sum<LPL>(na_l);
    last = seg_max<LPL>(last);
    bestn = seg_max<LPL>(bestn);
    int best = 0x7fffffff;    // first maximum == min over ties (utils.py:263-271)
    for (int c = sl; c < A; c += LPL)
        if (ccl[c] == bestn) { best = c; break; }
    best = -seg_max<LPL>(-best);
    na_s = both ? seg_sum<LPL>(na_s) : na_l;
    wave_lds_fence();
    // chains 1: sum f, sum f^2 (both partitions)
    const int j = sl & 3;
    const double* base1 = j == 0 ? Fl : j == 1 ? T1 : j == 2 ? Fs : T3;
    double fsum_l, sq_l, fsum_s, sq_s;
    {
        double acc = 0.0;
        if (sl < (both ? 4 : 2))
            for (int c = 0; c < A4; c += 4) {
                const double x0 = base1[c], x1 = base1[c + 1], x2 = base1[c + 2], x3 = base1[c + 3];
                acc += x0;
                acc += x1;
                acc += x2;
                acc += x3;
            }
        fsum_l = __shfl(acc, lane0, WAVE);
        sq_l = __shfl(acc, lane0 + 1, WAVE);
        fsum_s = both ? __shfl(acc, lane0 + 2, WAVE) : fsum_l;
        sq_s = both ? __shfl(acc, lane0 + 3, WAVE) : sq_l;
    }
    wave_lds_fence();
    // pass 2 (parallel): -(pk ln pk) with pk = f / sum f (scipy.stats.entropy normalises), v f
    for (int c = sl; c < A4; c += LPL) {
        const double fl = Fl[c];
        double e = 0.0;
        if (fl != 0.0) {
            const double pk = fl / fsum_l;
            e = -(pk * log(pk));
        }
        T1[c] = e;
        T3[c] = CV[c] * fl;           // utils.py:236
        if (both) {
            const double fs = Fs[c];
            double es = 0.0;
            if (fs != 0.0) {
                const double pk = fs / fsum_s;
                es = -(pk * log(pk));
            }
            Fs[c] = es;
        }
    }
    wave_lds_fence();
    const double* base2 = j == 0 ? T1 : j == 1 ? T3 : Fs;
    double ent_l, ent_s, mean;
    {
        double acc = 0.0;
        if (sl < (both ? 3 : 2))
            for (int c = 0; c < A4; c += 4) {
                const double x0 = base2[c], x1 = base2[c + 1], x2 = base2[c + 2], x3 = base2[c + 3];
                acc += x0;
                acc += x1;
                acc += x2;
                acc += x3;
            }
        ent_l = __shfl(acc, lane0, WAVE);
        mean = __shfl(acc, lane0 + 1, WAVE);
        ent_s = both ? __shfl(acc, lane0 + 2, WAVE) : ent_l;
    }
    wave_lds_fence();
    // pass 3: the variance's terms, f (v - mean)^2   (utils.py:296)
    for (int c = sl; c < A4; c += LPL) {
        const double d = CV[c] - mean;
        T1[c] = Fl[c] * (d * d);
    }
    wave_lds_fence();
    double var;
    {
        double acc = 0.0;
        if (sl == 0)
            for (int c = 0; c < A4; c += 4) {
                const double x0 = T1[c], x1 = T1[c + 1], x2 = T1[c + 2], x3 = T1[c + 3];
                acc += x0;
                acc += x1;
                acc += x2;
                acc += x3;
            }
        var = __shfl(acc, lane0, WAVE);
    }
    // the rows (mode_stats / k_locus_finalize)
    const bool have = total > 0;
    const bool ok_l = have && fabs(1.0 - fsum_l) <= 0.001, ok_s = have && fabs(1.0 - fsum_s) <= 0.001;   // utils.py:140
    ent_l /= 0.693147180559945309417232;
    ent_s /= 0.693147180559945309417232;
    ent_l = ent_l == 0.0 ? 0.0 : ent_l;
    ent_s = ent_s == 0.0 ? 0.0 : ent_s;
    const int st_base = n_called == 0 ? TRK_HWE_VALUE_ERROR : fa.pl < 2 ? TRK_HWE_INDEX_ERROR : n_low > 0 ? TRK_HWE_NAN : TRK_HWE_OK;
    const int st_l = ok_l ? st_base : TRK_HWE_NAN, st_s = ok_s ? st_base : TRK_HWE_NAN;
    if (live) {
        const int ns_real = fa.ns_real;
        double fv;
        switch (sl) {
            case TRK_LF_THRESH: fv = have && last >= 0 ? CV[last < 0 ? 0 : last] : nan; break;
            case TRK_LF_MEAN: fv = ok_l ? mean : nan; break;
            case TRK_LF_MODE: fv = ok_l ? CV[best >= A ? 0 : best] : nan; break;
            case TRK_LF_VAR: fv = ok_l ? var : nan; break;
            case TRK_LF_HET_LEN: fv = ok_l ? 1.0 - sq_l : nan; break;       // utils.py:175
            case TRK_LF_HET_STR: fv = ok_s ? 1.0 - sq_s : nan; break;
            case TRK_LF_ENTROPY_LEN: fv = ok_l ? ent_l : nan; break;
            case TRK_LF_ENTROPY_STR: fv = ok_s ? ent_s : nan; break;
            case TRK_LF_CALLRATE: fv = ns_real > 0 ? (double)n_called / (double)ns_real : nan; break;   // tr_harmonizer.py:946
            case 11: fv = 0.0; break;
            default: fv = nan; break;      // the two HWE p-values: k_hwe_test_slots
        }
        int iv;
        switch (sl) {
            case TRK_LI_N_CALLED: iv = n_called; break;
            case TRK_LI_N_LOWPLOIDY: iv = n_low; break;
            case TRK_LI_N_HOM_LEN: iv = n_homl; break;
            case TRK_LI_N_HOM_STR: iv = n_homs; break;
            case TRK_LI_N_ALLELES: iv = total; break;
            case TRK_LI_N_BAD: iv = n_bad; break;
            case TRK_LI_HWE_STATUS_LEN: iv = st_l; break;
            case TRK_LI_HWE_STATUS_STR: iv = st_s; break;
            case TRK_LI_N_SAMPLES: iv = ns_real; break;
            case TRK_LI_NALLELES_LEN: iv = have ? na_l : 0; break;
            case TRK_LI_NALLELES_STR: iv = have ? na_s : 0; break;
            default: iv = 0; break;
        }
        if (sl < TRK_LF_COLS) {
            fa.locus_f64[(int64_t)fa.row * TRK_LF_COLS + sl] = fv;
            fa.locus_int[(int64_t)fa.row * TRK_LI_COLS + sl] = iv;
        }
        if (fa.hwe_count) {   // compact work list (k_hwe_test<W>), as k_locus_finalize pushes it
            if (sl == 12 && st_l == TRK_HWE_OK) {
                const HweItem it = {fa.row, same ? 3 : 1, n_homl, n_called, sq_l};
                fa.items[atomicAdd(fa.hwe_count, 1u)] = it;
            }
            if (sl == 13 && !same && st_s == TRK_HWE_OK) {
                const HweItem it = {fa.row, 2, n_homs, n_called, sq_s};
                fa.items[atomicAdd(fa.hwe_count, 1u)] = it;
            }
        } else {              // fixed slots (k_hwe_test_slots)
            if (sl == 12) {
                const HweItem it = {fa.row, st_l == TRK_HWE_OK ? (same ? 3 : 1) : 0, n_homl, n_called, sq_l};
                fa.items[fa.row] = it;
            }
            if (sl == 13) {
                const HweItem it = {fa.row, (!same && st_s == TRK_HWE_OK) ? 2 : 0, n_homs, n_called, sq_s};
                fa.items[fa.n_rows + fa.row] = it;
            }
        }
    }
}


template <int R, int U, bool FIN = false>
__global__ __launch_bounds__(WAVE* COUNT_WAVES_PER_WG) void k_locus_count_v3(
    trk_batch b, int32_t* __restrict__ allele_count, int32_t* __restrict__ locus_int, int nbmax,
    int wave_lds_words, int64_t twin_ac, int64_t twin_li, V3Fin fin) {
    static_assert(R == 2 || R == 4, "loci per wave");
    constexpr int LPL = WAVE / R;                 // lanes per locus
    constexpr int KC = R == 4 ? 16 : 32;          // histogram columns of one locus
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x >> 6;
    const int sub = lane / LPL, sl = lane % LPL;
    const int l_raw = (blockIdx.x * COUNT_WAVES_PER_WG + wid) * R + sub;
    if ((blockIdx.x * COUNT_WAVES_PER_WG + wid) * R >= b.n_loci) return;   // whole wave beyond the batch
    const bool live = l_raw < b.n_loci;
    const int l = live ? l_raw : b.n_loci - 1;
    const int S = b.n_samples;
    const int64_t RS = b.row_stride ? b.row_stride : S;     // a column-range view: rows are RS samples apart
    const u32x4* row = reinterpret_cast<const u32x4*>(b.gt) + (((int64_t)l * RS) >> 2);
    const int nchunks = live ? (S >> 2) : 0;
    u32x4 cur[U];
    row_fetch<LPL, U>(row, 0, sl, (S >> 2) - 1, cur);     // the row's first chunks travel during the prologue
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    const int nbins = A + 7;
    uint32_t* wbase = lds + (size_t)wid * wave_lds_words;
    const int cbase = R == 4 ? (sub & 1) * 16 : 0;
    uint32_t* hrow = wbase + (size_t)(lane >> 5) * nbmax * 32;            // this 32-lane group's rows
    uint32_t* hcol = hrow + (lane & 31);
    uint32_t* lut = wbase + (size_t)2 * nbmax * 32 + (size_t)sub * nbmax;  // indexed by BIN
    int ml = 0, ms = 0;
    for (int a = sl; a < A; a += LPL) {
        const int lc = b.len_class[off + a], sc = b.str_class[off + a];
        lut[a + 2] = (uint32_t)lc | ((uint32_t)sc << 16);
        ml = lc > ml ? lc : ml;
        ms = sc > ms ? sc : ms;
    }
    if (sl == 0) {  // sentinel bins: classes no allele has, distinct from each other
        lut[0] = 0xffffffffu;
        lut[1] = 0xfffefffeu;
        lut[A + 2] = 0xfffdfffdu;
    }
    ml = seg_max<LPL>(ml);
    ms = seg_max<LPL>(ms);
    const bool dup = (ml + 1 < A) | (ms + 1 < A);
    const bool any_dup = __ballot(dup) != 0ull;
    {   // (2 * nbmax * 32 words: a multiple of four, 16-byte aligned)
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = lane; i < 2 * nbmax * 8; i += WAVE) reinterpret_cast<u32x4*>(wbase)[i] = z4;
    }
    wave_lds_fence();

    uint32_t n_eq = 0, n_hl = 0, n_hs = 0;
    const uint32_t amax2 = (uint32_t)(A + 2) * 0x00010001u;
    const uint32_t hcol_b = (uint32_t)(uintptr_t)(lds_u32p)hcol, lut_b = (uint32_t)(uintptr_t)(lds_u32p)lut;
    const uint32_t c128 = 128u, c4 = 4u, combo_b = hcol_b + ((uint32_t)(A + 3) << 7);
    // (every locus of the batch has S samples: the loop count is uniform over the wave; a wave's dead tail loci
    // -- nchunks == 0 -- only skip the consumer)
    if (any_dup)
        row_stream<LPL, U>(row, S >> 2, sl, cur, [&](uint32_t w) {
            if (nchunks) v3_cell<true>(w, amax2, hcol_b, lut_b, c128, c4, combo_b, n_eq, n_hl, n_hs);
        });
    else
        row_stream<LPL, U>(row, S >> 2, sl, cur, [&](uint32_t w) {
            if (nchunks) v3_cell<false>(w, amax2, hcol_b, lut_b, c128, c4, combo_b, n_eq, n_hl, n_hs);
        });
    wave_lds_fence();
    // fold this locus's KC columns of every bin (rotated start: conflict-free); the total is parked in the locus's
    // first column of the row (only this lane touches the bin's half row)
    for (int bin = sl; bin < nbins; bin += LPL) {
        uint32_t s = 0;
        const u32x4* hp = reinterpret_cast<const u32x4*>(hrow + (bin << 5) + cbase);
#pragma unroll
        for (int k = 0; k < KC / 4; ++k) {    // 16-byte reads, the start rotated by lane (fewer lanes on one bank group)
            const u32x4 h = hp[(k + sl) & (KC / 4 - 1)];
            s += (h.x + h.y) + (h.z + h.w);
        }
        if (live && bin >= 2 && bin < A + 2) {
            allele_count[off + bin - 2] = (int32_t)s;
            if (twin_ac) allele_count[twin_ac + off + bin - 2] = (int32_t)s;
        }
        if (FIN)   // (the histogram rows become the finaliser's scratch: the totals go to a row of their own)
            wbase[2 * nbmax * 32 + (R + sub) * nbmax + bin] = s;
        else
            hrow[(bin << 5) + cbase] = s;
    }
    wave_lds_fence();
    n_eq = (uint32_t)seg_sum<LPL>((int)n_eq);
    if (any_dup) {
        n_hl = (uint32_t)seg_sum<LPL>((int)n_hl);
        n_hs = (uint32_t)seg_sum<LPL>((int)n_hs);
    }
    if (FIN) {
        const uint32_t* h = wbase + 2 * nbmax * 32 + (R + sub) * nbmax;     // the locus's bin totals
        const int h_m2 = (int)h[0], h_m1 = (int)h[1];
        const int n_bad = (int)h[A + 2];
        const int c00 = (int)h[A + 3], c10 = (int)h[A + 4], c01 = (int)h[A + 5], c11 = (int)h[A + 6];
        const int hom_idx = (int)n_eq - c11 - c00;
        const int n_called = S - (h_m1 - c11), n_low = h_m2 - c00 - c10 - c01;
        const int n_homl = dup ? (int)n_hl - c11 - c00 : hom_idx, n_homs = dup ? (int)n_hs - c11 - c00 : hom_idx;
        CoopFin fa;
        fa.cnt = h + 2;
        fa.cls = lut + 2;
        fa.cv = b.len_class_value + off;
        fa.scratch = wbase + sub * fin.fin_words;       // (over the folded histogram rows)
        fa.amax4 = fin.amax4;
        fa.row = l;
        fa.n_rows = b.n_loci;
        fa.pl = 2;
        fa.ns_real = S - b.n_pad_samples;
        fa.nalleles_thresh = fin.nalleles_thresh;
        fa.locus_int = locus_int;
        fa.locus_f64 = fin.locus_f64;
        fa.hwe_count = nullptr;
        fa.items = fin.items;
        coop_finalize<LPL>(fa, live, any_dup, sl, sub * LPL, A, n_called, n_low, n_homl, n_homs, n_bad);
    } else if (sl == 0 && live) {
        const uint32_t* h = hrow + cbase;
        const int h_m2 = (int)h[0 << 5], h_m1 = (int)h[1 << 5];
        const int n_bad = (int)h[(A + 2) << 5];
        const int c00 = (int)h[(A + 3) << 5];  // (-2,-2)
        const int c10 = (int)h[(A + 4) << 5];  // lo = -1, hi = -2
        const int c01 = (int)h[(A + 5) << 5];  // lo = -2, hi = -1
        const int c11 = (int)h[(A + 6) << 5];  // (-1,-1)
        const int hom_idx = (int)n_eq - c11 - c00;
        const u32x4 r0 = {(uint32_t)(S - (h_m1 - c11)), (uint32_t)(h_m2 - c00 - c10 - c01),
                          (uint32_t)(dup ? (int)n_hl - c11 - c00 : hom_idx),
                          (uint32_t)(dup ? (int)n_hs - c11 - c00 : hom_idx)};
        const u32x4 r1 = {0u, (uint32_t)n_bad, 0u, 0u};
        const u32x4 r2 = {(uint32_t)(S - b.n_pad_samples), 0u, 0u, 0u};
        for (int64_t tw = 0;; tw = twin_li) {   // the whole 48-byte row, as in k_locus_count_v2
            u32x4* li0 = reinterpret_cast<u32x4*>(locus_int + tw + (int64_t)l * TRK_LI_COLS);
            li0[0] = r0;
            li0[1] = r1;
            li0[2] = r2;
            if (tw == twin_li) break;
        }
    }
    wave_lds_fence();
}

// ---------------------------------------------------------------------------
// k_locus_count_v2g : k_locus_count_v2 for sample groups (statSTR --samples a,b,...: statSTR.py:520-542).
// A sample's group bits (<= 3 groups here) are its CLASS; every class has its own histogram (all bins of
// k_locus_count_v2 plus "both bins equal" / "same length class" / "same sequence class" rows), so the stream
// does the same work per call as the ungrouped kernel plus one byte of group bits per sample, and a group's
// totals are the sums over the classes that contain it -- taken once per locus after the stream.
// bins per class: 0 '-2', 1 '-1', 2..A+1 alleles, A+2 out of range, A+3..A+6 sentinel pairs, A+7 eq, A+8 hl, A+9 hs
// ---------------------------------------------------------------------------
constexpr int V2G_EXTRA = 10;

// (round 3: LDS byte addresses straight from the packed 16-bit bins, as in v2m_cell -- one multiply-add for the class's
// histogram, one v_mad_u32_u16 per half; the rare rows as predicated adds on precomputed bin addresses.  The first form
// indexed `hist + cls * cstride + kslot` by words and shifted every bin by kshift: ~30 vector instructions per cell.)
struct V2gBase {
    uint32_t hist_b;     // LDS byte address of hist + 4 * kslot
    uint32_t kbytes;     // bytes from one bin to the next (4 << kshift)
    uint32_t cstride_b;  // bytes from one class's histogram to the next
    uint32_t pairs_b;    // byte offset of bin A + 3 (the four sentinel pairs)
    uint32_t eq_b;       // byte offset of bin A + 7 ('both bins equal'; A + 8 / A + 9: same length / sequence class)
    uint32_t lut_b;      // LDS byte address of the class LUT
};
__device__ __forceinline__ void lds_inc(uint32_t addr_b) {
    __hip_atomic_fetch_add((lds_u32p)(uintptr_t)addr_b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool DUP>
__device__ __forceinline__ void v2g_cell(uint32_t w, uint32_t cls, uint32_t amax2, const V2gBase& g) {
    u16x2 u = __builtin_bit_cast(u16x2, w) + (u16x2){2, 2};
    u16x2 t2 = __builtin_elementwise_min(u, __builtin_bit_cast(u16x2, amax2));
    const uint32_t t = __builtin_bit_cast(uint32_t, t2);
    const uint32_t base = cls * g.cstride_b + g.hist_b;
    lds_inc(mad16_lo(t, g.kbytes, base));
    lds_inc(mad16_hi(t, g.kbytes, base));
    if ((t & 0xfffefffeu) == 0u)    // both haplotypes sentinels: pair lo + 2 * hi
        lds_inc(((t | (t >> 15)) & 3u) * g.kbytes + (base + g.pairs_b));
    if (DUP) {
        const uint32_t x = *(lds_cu32p)(uintptr_t)mad16_lo(t, 4u, g.lut_b) ^ *(lds_cu32p)(uintptr_t)mad16_hi(t, 4u, g.lut_b);
        if ((x & 0xffffu) == 0u) lds_inc(base + g.eq_b + g.kbytes);
        if (x < 0x10000u) lds_inc(base + g.eq_b + 2u * g.kbytes);
    } else if ((t & 0xffffu) == (t >> 16)) {
        lds_inc(base + g.eq_b);
    }
}

template <bool DUP>
__device__ __forceinline__ void v2g_row(const u32x4* __restrict__ row, const uint32_t* __restrict__ gbits4,
                                        int nchunks, int lane, uint32_t amax2, const V2gBase& g, uint32_t cmask) {
    constexpr int U = 2;
    int c = lane;
    for (; c + (U - 1) * WAVE < nchunks; c += U * WAVE) {
        u32x4 v[U];
        uint32_t gb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = __builtin_nontemporal_load(&row[c + u * WAVE]);
            gb[u] = gbits4[c + u * WAVE];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) v2g_cell<DUP>(v[u][j], (gb[u] >> (8 * j)) & cmask, amax2, g);
    }
    for (; c < nchunks; c += WAVE) {
        const u32x4 v = __builtin_nontemporal_load(&row[c]);
        const uint32_t gb = gbits4[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) v2g_cell<DUP>(v[j], (gb >> (8 * j)) & cmask, amax2, g);
    }
}

__global__ __launch_bounds__(WAVE* COUNT_WAVES_PER_WG) void k_locus_count_v2g(
    trk_batch b, int32_t* __restrict__ allele_count, int32_t* __restrict__ locus_int, int kshift,
    int wave_lds_words) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x >> 6;
    const int l = blockIdx.x * COUNT_WAVES_PER_WG + wid;
    if (l >= b.n_loci) return;  // waves are independent: no workgroup barrier anywhere
    const int S = b.n_samples, G = b.n_groups;
    const int ncls = 1 << G;
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    const int K = 1 << kshift;
    const int kslot = lane & (K - 1);
    const int nbins = A + V2G_EXTRA;
    const int cstride = nbins << kshift;
    uint32_t* hist = lds + (size_t)wid * wave_lds_words;
    uint32_t* lut = hist + ncls * cstride;  // indexed by BIN
    int ml = 0, ms = 0;
    for (int a = lane; a < A; a += WAVE) {
        int lc = b.len_class[off + a], sc = b.str_class[off + a];
        lut[a + 2] = (uint32_t)lc | ((uint32_t)sc << 16);
        ml = lc > ml ? lc : ml;
        ms = sc > ms ? sc : ms;
    }
    if (lane == 0) {  // sentinel bins: classes no allele has, distinct from each other
        lut[0] = 0xffffffffu;
        lut[1] = 0xfffefffeu;
        lut[A + 2] = 0xfffdfffdu;
    }
    ml = wave_max(ml);
    ms = wave_max(ms);
    const bool dup = (ml + 1 < A) | (ms + 1 < A);
    for (int i = lane; i < ncls * cstride; i += WAVE) hist[i] = 0;
    wave_lds_fence();

    const u32x4* row = reinterpret_cast<const u32x4*>(b.gt) + (((int64_t)l * S) >> 2);
    const uint32_t* gbits4 = reinterpret_cast<const uint32_t*>(b.group_bits);
    const uint32_t amax2 = (uint32_t)(A + 2) * 0x00010001u;
    V2gBase gb;
    gb.kbytes = 4u << kshift;
    gb.hist_b = (uint32_t)(uintptr_t)(lds_u32p)hist + 4u * (uint32_t)kslot;
    gb.cstride_b = 4u * (uint32_t)cstride;
    gb.pairs_b = (uint32_t)(A + 3) * gb.kbytes;
    gb.eq_b = (uint32_t)(A + 7) * gb.kbytes;
    gb.lut_b = (uint32_t)(uintptr_t)(lds_u32p)lut;
    if (dup)
        v2g_row<true>(row, gbits4, S >> 2, lane, amax2, gb, (uint32_t)ncls - 1u);
    else
        v2g_row<false>(row, gbits4, S >> 2, lane, amax2, gb, (uint32_t)ncls - 1u);
    wave_lds_fence();
    // fold: per bin the K copies of every class, then per group the classes that contain it.  Group g's total of
    // a bin is parked in copy 0 of class g's bin (only this lane touches the bin's words).
    int hap[3] = {0, 0, 0};   // haplotypes (alleles + sentinels + out of range) of the group: 2 per sample
    for (int bin = lane; bin < nbins; bin += WAVE) {
        uint32_t cs[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            cs[c] = 0;
            if (c < ncls)
                for (int k = 0; k < K; ++k) cs[c] += hist[c * cstride + (bin << kshift) + ((k + lane) & (K - 1))];
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (g >= G) break;
            uint32_t tot = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((c >> g) & 1) tot += cs[c];
            if (bin >= 2 && bin < A + 2) allele_count[(int64_t)g * b.n_alleles_total + off + bin - 2] = (int32_t)tot;
            if (bin <= A + 2) hap[g] += (int)tot;
            hist[g * cstride + (bin << kshift)] = tot;
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if (g >= G) break;
        const int n_samp = wave_sum(hap[g]) >> 1;
        if (lane == 0) {
            const uint32_t* h = hist + g * cstride;
            const int h_m2 = (int)h[0 << kshift], h_m1 = (int)h[1 << kshift];
            const int n_bad = (int)h[(A + 2) << kshift];
            const int c00 = (int)h[(A + 3) << kshift];  // (-2,-2)
            const int c10 = (int)h[(A + 4) << kshift];  // lo = -1, hi = -2
            const int c01 = (int)h[(A + 5) << kshift];  // lo = -2, hi = -1
            const int c11 = (int)h[(A + 6) << kshift];  // (-1,-1)
            const int hom_idx = (int)h[(A + 7) << kshift] - c11 - c00;
            int32_t* li0 = locus_int + ((int64_t)g * b.n_loci + l) * TRK_LI_COLS;
            li0[TRK_LI_N_CALLED] = n_samp - (h_m1 - c11);
            li0[TRK_LI_N_LOWPLOIDY] = h_m2 - c00 - c10 - c01;
            li0[TRK_LI_N_HOM_LEN] = dup ? (int)h[(A + 8) << kshift] - c11 - c00 : hom_idx;
            li0[TRK_LI_N_HOM_STR] = dup ? (int)h[(A + 9) << kshift] - c11 - c00 : hom_idx;
            li0[TRK_LI_N_BAD] = n_bad;
            li0[TRK_LI_N_SAMPLES] = n_samp;
        }
    }
    wave_lds_fence();
}

template <int U>
__global__ __launch_bounds__(WAVE* COUNT_WAVES_PER_WG) void k_locus_count_fast(
    trk_batch b, int32_t* __restrict__ allele_count, int32_t* __restrict__ locus_int, int kshift,
    int wave_lds_words) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x >> 6;
    const int l = blockIdx.x * COUNT_WAVES_PER_WG + wid;
    if (l >= b.n_loci) return;  // waves are independent: no workgroup barrier anywhere
    const int S = b.n_samples;
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    const int K = 1 << kshift;
    const int kslot = lane & (K - 1);
    uint32_t* hist = lds + (size_t)wid * wave_lds_words;
    uint32_t* lut = hist + ((A + 1) << kshift);
    int pl = b.locus_ploidy ? (int)b.locus_ploidy[l] : 2;
    // a haploid record inside a diploid batch: the second column is not part of the record (it holds the -2
    // padding): REPLACE it by -3 (neither a no-call nor padding, never counted).  [OR-ing -3 over the -2 gave -1,
    // i.e. every call of such a record read as missing -- found by tests/test_gpu_property.py]
    const uint32_t hap1_fix = pl < 2 ? 0xfffd0000u : 0u;
    // dense class ranks: a duplicate (two indices, one class) exists iff max rank + 1 < A
    int ml = 0, ms = 0;
    for (int a = lane; a < A; a += WAVE) {
        int lc = b.len_class[off + a], sc = b.str_class[off + a];
        lut[a] = (uint32_t)lc | ((uint32_t)sc << 16);
        ml = lc > ml ? lc : ml;
        ms = sc > ms ? sc : ms;
    }
    if (lane == 0) lut[A] = 0xffffffffu;
    ml = wave_max(ml);
    ms = wave_max(ms);
    const bool dup = (ml + 1 < A) | (ms + 1 < A);
    for (int i = lane; i < ((A + 1) << kshift); i += WAVE) hist[i] = 0;
    wave_lds_fence();

    int n_miss = 0, n_low = 0, n_hom = 0, n_hl = 0, n_hs = 0, n_bad = 0;
    const u32x4* row = reinterpret_cast<const u32x4*>(b.gt) + (((int64_t)l * S) >> 2);
    const int nchunks = S >> 2;
    if (dup)
        fast_row<true, U>(row, nchunks, lane, A, hist, lut, kshift, kslot, hap1_fix, n_miss, n_low, n_hom, n_hl, n_hs,
                          n_bad);
    else
        fast_row<false, U>(row, nchunks, lane, A, hist, lut, kshift, kslot, hap1_fix, n_miss, n_low, n_hom, n_hl,
                           n_hs, n_bad);
    wave_lds_fence();
    for (int bin = lane; bin < A; bin += WAVE) {
        uint32_t s = 0;
        for (int k = 0; k < K; ++k) s += hist[(bin << kshift) + ((k + lane) & (K - 1))];
        allele_count[off + bin] = (int32_t)s;
    }
    n_miss = wave_sum(n_miss);
    n_low = wave_sum(n_low);
    n_hom = wave_sum(n_hom);
    n_bad = wave_sum(n_bad);
    if (dup) {
        n_hl = wave_sum(n_hl);
        n_hs = wave_sum(n_hs);
    } else {
        n_hl = n_hs = n_hom;
    }
    if (lane == 0) {
        int32_t* li0 = locus_int + (int64_t)l * TRK_LI_COLS;
        li0[TRK_LI_N_CALLED] = S - n_miss;
        li0[TRK_LI_N_LOWPLOIDY] = n_low;
        li0[TRK_LI_N_HOM_LEN] = pl == 2 ? n_hl : 0;
        li0[TRK_LI_N_HOM_STR] = pl == 2 ? n_hs : 0;
        li0[TRK_LI_N_BAD] = n_bad;
        li0[TRK_LI_N_SAMPLES] = S - b.n_pad_samples;
    }
}

// ---------------------------------------------------------------------------
// k_locus_finalize : one thread per (group, locus)
// ---------------------------------------------------------------------------
struct ModeStats {
    double het, entropy, hwep, sq;
    int nalleles, status;
};

// class counts are in cc[0..ncls), ascending class order == the dict order the
// reference iterates in (np.unique sorts keys; tr_harmonizer.py:1495-1499)
__device__ __forceinline__ void mode_stats(const int32_t* cc, int cstride, int ncls, int64_t total,
                                           double nalleles_thresh, int n_called, int n_low, int n_hom, int pl,
                                           ModeStats& o) {
    const double nan = __builtin_nan("");
    o.het = o.entropy = o.hwep = nan;
    o.sq = 0.0;
    o.nalleles = 0;
    o.status = TRK_HWE_NAN;
    if (total <= 0) return;  // ValidateAlleleFreqs: empty dict  (utils.py:139)
    const double ft = (double)total;  // float(sum(counts))   (tr_harmonizer.py:1539)
    double fsum = 0.0, sq = 0.0;
    int na = 0;
    for (int c = 0; c < ncls; ++c) {
        int n = cc[c * cstride];
        if (n == 0) continue;
        double f = (double)n / ft;
        fsum += f;
        sq += f * f;
        na += f >= nalleles_thresh;
    }
    o.nalleles = na;  // statSTR.py:207 (no validity check there)
    if (!(fabs(1.0 - fsum) <= 0.001)) return;  // utils.py:140
    o.het = 1.0 - sq;                          // utils.py:175
    // scipy.stats.entropy(pk, base=2): pk /= sum(pk); -sum(pk ln pk) / ln 2
    double ent = 0.0;
    for (int c = 0; c < ncls; ++c) {
        int n = cc[c * cstride];
        if (n == 0) continue;
        double pk = ((double)n / ft) / fsum;
        ent -= pk * log(pk);
    }
    ent /= 0.693147180559945309417232;
    o.entropy = ent == 0.0 ? 0.0 : ent;  // normalise -0.0
    // GetHardyWeinbergBinomialTest utils.py:325-338
    if (n_called == 0) {
        o.status = TRK_HWE_VALUE_ERROR;  // binomtest(0, n=0): ValueError
    } else if (pl < 2) {
        o.status = TRK_HWE_INDEX_ERROR;  // gt[1] on a 1-tuple
    } else if (n_low > 0) {
        o.status = TRK_HWE_NAN;  // a -2 / ',' haplotype is not in allele_freqs
    } else {
        o.status = TRK_HWE_OK;  // p-value computed by k_hwe_test
        o.sq = sq;
    }
}

constexpr int FIN_THREADS = 64;
// LDS_CC: class counts live in thread-private LDS columns ([class][thread], so a
// wave's accesses are bank-conflict free); otherwise in a global scratch segment.
template <bool LDS_CC>
__global__ __launch_bounds__(FIN_THREADS) void k_locus_finalize(trk_batch b, const int32_t* __restrict__ allele_count,
                                                               int32_t* __restrict__ locus_int,
                                                               double* __restrict__ locus_f64,
                                                               int32_t* __restrict__ scratch,
                                                               double nalleles_thresh, int max_alleles,
                                                               unsigned int* __restrict__ hwe_count,
                                                               HweItem* __restrict__ hwe_items) {
    extern __shared__ uint32_t fin_lds[];
    const int L = b.n_loci;
    const int G = b.group_bits ? b.n_groups : 1;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * L) return;
    const int g = (int)(t / L);
    const int l = (int)(t - (int64_t)g * L);
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    const int64_t sumA = b.n_alleles_total;
    const int32_t* cnt = allele_count + (int64_t)g * sumA + off;
    int32_t* ccl;
    int32_t* ccs;
    int cs;
    if (LDS_CC) {
        ccl = reinterpret_cast<int32_t*>(fin_lds) + threadIdx.x;
        ccs = ccl + max_alleles * FIN_THREADS;
        cs = FIN_THREADS;
    } else {
        ccl = scratch + ((int64_t)g * 2 + 0) * sumA + off;
        ccs = scratch + ((int64_t)g * 2 + 1) * sumA + off;
        cs = 1;
    }
    int32_t* li = locus_int + ((int64_t)g * L + l) * TRK_LI_COLS;
    double* lf = locus_f64 + ((int64_t)g * L + l) * TRK_LF_COLS;
    const double nan = __builtin_nan("");

    for (int a = 0; a < A; ++a) {
        ccl[a * cs] = 0;
        ccs[a * cs] = 0;
    }
    int64_t total = 0;
    for (int a = 0; a < A; ++a) {
        int n = cnt[a];
        ccl[b.len_class[off + a] * cs] += n;
        ccs[b.str_class[off + a] * cs] += n;
        total += n;
    }
    li[TRK_LI_N_ALLELES] = (int32_t)total;
    const int n_called = li[TRK_LI_N_CALLED];
    const int n_low = li[TRK_LI_N_LOWPLOIDY];
    int pl = b.locus_ploidy ? (int)b.locus_ploidy[l] : b.ploidy;

    ModeStats ml, ms;
    mode_stats(ccl, cs, A, total, nalleles_thresh, n_called, n_low, li[TRK_LI_N_HOM_LEN], pl, ml);
    // identical partitions of the allele indices (the common case: no two alleles
    // share a length or a sequence) give identical class counts in both modes
    bool same = li[TRK_LI_N_HOM_LEN] == li[TRK_LI_N_HOM_STR];
    for (int a = 0; a < A && same; ++a) same = ccl[a * cs] == ccs[a * cs];
    if (same)
        ms = ml;
    else
        mode_stats(ccs, cs, A, total, nalleles_thresh, n_called, n_low, li[TRK_LI_N_HOM_STR], pl, ms);
    {
        const int32_t slot = (int32_t)((int64_t)g * L + l);
        if (ml.status == TRK_HWE_OK) {
            HweItem it = {slot, same ? 3 : 1, li[TRK_LI_N_HOM_LEN], n_called, ml.sq};
            hwe_items[atomicAdd(hwe_count, 1u)] = it;
        }
        if (!same && ms.status == TRK_HWE_OK) {
            HweItem it = {slot, 2, li[TRK_LI_N_HOM_STR], n_called, ms.sq};
            hwe_items[atomicAdd(hwe_count, 1u)] = it;
        }
    }
    li[TRK_LI_HWE_STATUS_LEN] = ml.status;
    li[TRK_LI_HWE_STATUS_STR] = ms.status;
    li[TRK_LI_NALLELES_LEN] = ml.nalleles;
    li[TRK_LI_NALLELES_STR] = ms.nalleles;
    lf[TRK_LF_HET_LEN] = ml.het;
    lf[TRK_LF_HET_STR] = ms.het;
    lf[TRK_LF_ENTROPY_LEN] = ml.entropy;
    lf[TRK_LF_ENTROPY_STR] = ms.entropy;
    lf[TRK_LF_HWEP_LEN] = ml.hwep;
    lf[TRK_LF_HWEP_STR] = ms.hwep;

    // always-by-length statistics (statSTR.py:126,347,375,402)
    double thresh = nan, mean = nan, mode = nan, var = nan;
    if (total > 0) {
        const double* cv = b.len_class_value + off;
        const double ft = (double)total;
        double fsum = 0.0;
        int best = -1, bestn = 0;
        for (int c = 0; c < A; ++c) {
            int n = ccl[c * cs];
            if (n == 0) continue;
            fsum += (double)n / ft;
            thresh = cv[c];  // ascending classes: the last non-empty one is the max
            if (n > bestn) {  // first maximum == min over ties (utils.py:263-271)
                bestn = n;
                best = c;
            }
        }
        if (fabs(1.0 - fsum) <= 0.001) {
            double m = 0.0;
            for (int c = 0; c < A; ++c) {
                int n = ccl[c * cs];
                if (n == 0) continue;
                m += cv[c] * ((double)n / ft);  // utils.py:236
            }
            double v = 0.0;
            for (int c = 0; c < A; ++c) {
                int n = ccl[c * cs];
                if (n == 0) continue;
                double d = cv[c] - m;
                v += ((double)n / ft) * (d * d);  // utils.py:296
            }
            mean = m;
            var = v;
            mode = cv[best];
        }
    }
    lf[TRK_LF_THRESH] = thresh;
    lf[TRK_LF_MEAN] = mean;
    lf[TRK_LF_MODE] = mode;
    lf[TRK_LF_VAR] = var;
    int ns = li[TRK_LI_N_SAMPLES];
    lf[TRK_LF_CALLRATE] = ns > 0 ? (double)n_called / (double)ns : nan;  // tr_harmonizer.py:946
    lf[11] = 0.0;
}

// ---------------------------------------------------------------------------
// k_locus_finalize_coop : k_locus_finalize by SIXTEEN lanes per (group, locus) -- coop_finalize, the epilogue of the fused
// small-batch pass, on counts that are already in memory (after any count kernel, after the call filters' delta
// outputs).  Same bits as the one-thread kernel (same operations in the same order); the HWE tests go to the same
// compact work list.  One thread per locus walks ~37 classes with a float64 division and a logarithm each, serially;
// here the classes go side by side -- a shorter chain for a small batch (command-line batches of large cohorts are
// 800-4000 loci), more instructions in all: launch_locus_finalize takes it up to 4096 rows.
// ---------------------------------------------------------------------------
constexpr int FC_LPL = 16, FC_THREADS = 256;
__global__ __launch_bounds__(FC_THREADS) void k_locus_finalize_coop(trk_batch b, const int32_t* __restrict__ allele_count,
                                                                    int32_t* __restrict__ locus_int,
                                                                    double* __restrict__ locus_f64,
                                                                    double nalleles_thresh, int amax4,
                                                                    unsigned int* __restrict__ hwe_count,
                                                                    HweItem* __restrict__ hwe_items) {
    extern __shared__ uint32_t fc_lds[];
    const int L = b.n_loci;
    const int G = b.group_bits ? b.n_groups : 1;
    const int64_t n_units = (int64_t)G * L;
    const int lane = threadIdx.x & (WAVE - 1);
    const int sl = lane % FC_LPL, sub = lane / FC_LPL;
    const int64_t wave_unit0 = ((int64_t)blockIdx.x * FC_THREADS + (threadIdx.x & ~(WAVE - 1))) / FC_LPL;
    if (wave_unit0 >= n_units) return;     // whole wave beyond the batch
    const int64_t unit = wave_unit0 + sub;
    const bool live = unit < n_units;
    const int t = (int)(live ? unit : n_units - 1);
    const int g = t / L, l = t - g * L;
    const int off = b.allele_off[l];
    const int A = b.allele_off[l + 1] - off;
    uint32_t* cnt = fc_lds + (size_t)(threadIdx.x / FC_LPL) * 14 * amax4;
    uint32_t* cls = cnt + amax4;
    const int32_t* ac = allele_count + (int64_t)g * b.n_alleles_total + off;
    for (int a = sl; a < A; a += FC_LPL) {
        cnt[a] = (uint32_t)ac[a];
        cls[a] = (uint32_t)b.len_class[off + a] | ((uint32_t)b.str_class[off + a] << 16);
    }
    const int32_t* li = locus_int + (int64_t)t * TRK_LI_COLS;
    const int n_called = li[TRK_LI_N_CALLED], n_low = li[TRK_LI_N_LOWPLOIDY], n_homl = li[TRK_LI_N_HOM_LEN],
              n_homs = li[TRK_LI_N_HOM_STR], n_bad = li[TRK_LI_N_BAD], ns = li[TRK_LI_N_SAMPLES];
    wave_lds_fence();
    CoopFin fa;
    fa.cnt = cnt;
    fa.cls = cls;
    fa.cv = b.len_class_value + off;
    fa.scratch = cls + amax4;
    fa.amax4 = amax4;
    fa.row = t;
    fa.n_rows = (int)n_units;
    fa.pl = b.locus_ploidy ? (int)b.locus_ploidy[l] : b.ploidy;
    fa.ns_real = ns;
    fa.nalleles_thresh = nalleles_thresh;
    fa.locus_int = locus_int;
    fa.locus_f64 = locus_f64;
    fa.hwe_count = hwe_count;
    fa.items = hwe_items;
    coop_finalize<FC_LPL>(fa, live, true, sl, sub * FC_LPL, A, n_called, n_low, n_homl, n_homs, n_bad);
}

// ---------------------------------------------------------------------------
// k_call_filter : dumpSTR call-level filters, column-owner tiling
// ---------------------------------------------------------------------------
constexpr int CF_THREADS = 256;
constexpr int CF_V = 4;  // samples per thread

// element (cell, col) of a plane: [L*S, ncol] interleaved, or -- TRK_DT_PLANAR -- [ncol, L*S]
__device__ __forceinline__ int64_t pidx(const trk_plane& p, int64_t cell, int col, int64_t nc) {
    return (p.dtype & TRK_DT_PLANAR) ? (int64_t)col * nc + cell : cell * p.ncol + col;
}
__device__ __forceinline__ bool p_is_f32(const trk_plane& p) { return (p.dtype & 0xff) == TRK_DT_F32; }
__device__ __forceinline__ int32_t p_i32(const trk_plane& p, int64_t cell, int col, int64_t nc) {
    return reinterpret_cast<const int32_t*>(p.data)[pidx(p, cell, col, nc)];
}
__device__ __forceinline__ float p_f32(const trk_plane& p, int64_t cell, int col, int64_t nc) {
    return reinterpret_cast<const float*>(p.data)[pidx(p, cell, col, nc)];
}
__device__ __forceinline__ double plane_val(const trk_plane& p, int64_t cell, int col, int64_t nc) {
    if (!p_is_f32(p)) return (double)p_i32(p, cell, col, nc);
    return (double)p_f32(p, cell, col, nc);
}

// evaluate one filter on one call; returns true when the filter fires (nc = L * S)
__device__ __forceinline__ bool eval_filter(const trk_call_filter& f, const trk_plane* planes, int64_t cell,
                                            bool called, const int* gtv, int P, int pl, int64_t nc) {
    const trk_plane& pa = planes[f.plane_a];
    switch (f.op) {
        case TRK_F_LT:
        case TRK_F_CALLED_LT: {
            if (f.op == TRK_F_CALLED_LT && !called) return false;
            if (p_is_f32(pa)) return p_f32(pa, cell, f.col_a, nc) < (float)f.thr;  // numpy compares float32 arrays in float32
            return (double)p_i32(pa, cell, f.col_a, nc) < f.thr;
        }
        case TRK_F_GT: {
            if (p_is_f32(pa)) return p_f32(pa, cell, f.col_a, nc) > (float)f.thr;
            return (double)p_i32(pa, cell, f.col_a, nc) > f.thr;
        }
        case TRK_F_RATIO_GT: {
            double a = plane_val(pa, cell, f.col_a, nc);
            double bb = plane_val(planes[f.plane_b], cell, f.col_b, nc);
            return (a / bb) > f.thr;  // int32/int32 -> float64 true division
        }
        case TRK_F_CALLED_SUM_LT: {
            if (!called) return false;
            if (p_is_f32(pa)) {
                float s = p_f32(pa, cell, f.col_a, nc) + p_f32(pa, cell, f.col_a2, nc);
                return s < (float)f.thr;
            }
            return (double)((int64_t)p_i32(pa, cell, f.col_a, nc) + (int64_t)p_i32(pa, cell, f.col_a2, nc)) < f.thr;
        }
        case TRK_F_CALLED_EQ: {
            if (!called) return false;
            return p_i32(pa, cell, f.col_a, nc) == p_i32(planes[f.plane_b], cell, f.col_b, nc);
        }
        case TRK_F_CALLED_SUM_EQ: {
            if (!called) return false;
            return (int64_t)p_i32(pa, cell, f.col_a, nc) + (int64_t)p_i32(pa, cell, f.col_a2, nc) ==
                   (int64_t)p_i32(planes[f.plane_b], cell, f.col_b, nc);
        }
        case TRK_F_CALLED_OUTSIDE_CI: {
            if (!called) return false;
            const trk_plane& pb = planes[f.plane_b];
            bool hit = false;
            for (int j = 0; j < pa.ncol; ++j) {
                const int32_t ml = p_i32(pa, cell, j, nc);
                hit = hit || ml < p_i32(pb, cell, 2 * j, nc) || p_i32(pb, cell, 2 * j + 1, nc) < ml;
            }
            return hit;
        }
        case TRK_F_AD_SUPPORT_LT: {
            // read_support[sample, gt_idx] with numpy negative indexing (filters.py:865)
            bool hit = false;
            for (int j = 0; j < pl; ++j) {
                int a = gtv[j];
                if (a < 0) a += pa.ncol;
                if (a < 0 || a >= pa.ncol) continue;
                hit |= (double)p_i32(pa, cell, a, nc) < f.thr;
            }
            return hit;
        }
        default:
            return false;
    }
}

constexpr int CF_MAX_GROUPS = 6;
struct CallArgs {
    trk_batch b;
    trk_plane planes[TRK_MAX_PLANES];
    trk_call_filter filters[TRK_MAX_FILTERS];
    int n_planes, n_filters, dp_plane, loci_per_block;
    int loci_per_wg;            // streaming kernel: loci of one workgroup, walked in sub-blocks of loci_per_block
    int64_t n_cells;            // L * S (column stride of planar planes)
    // vector sources of the streaming kernel: single-column planes and single columns of planar planes,
    // each a [L*S] array of 4-byte elements fetched as one 16-byte vector per thread and locus
    const void* src_ptr[16];
    uint32_t src_f32_mask;      // source holds float32
    int32_t n_src;
    // interleaved planes [L*S, k] (k = 2..4): the k vectors of a thread's four calls are fetched together and
    // de-interleaved into k consecutive sources (their src_ptr is NULL)
    const void* grp_ptr[CF_MAX_GROUPS];
    int8_t grp_base[CF_MAX_GROUPS], grp_k[CF_MAX_GROUPS];
    int32_t grp_n;
    // the same sources as LOAD SLOTS of the register-vector path (cf_load_r): slot s is one 16-byte vector per thread
    // at  slot_ptr[s] + ((cell0 / 4) * slot_mult[s] + slot_chunk[s]) * 16  (a planar source: mult 1, chunk 0; chunk i of
    // an interleaved plane with k columns: mult k, chunk i), landing in elements 4 s .. 4 s + 3 of the vector; the value
    // of logical source q for call j is element  src_off[q] + j * src_stride[q]  (planar: 4 q + j; column c of an
    // interleaved plane whose first slot is b: 4 b + c + j k)
    const void* slot_ptr[16];
    int8_t slot_mult[16], slot_chunk[16], src_off[16], src_stride[16];
    int8_t f_src_a[TRK_MAX_FILTERS], f_src_a2[TRK_MAX_FILTERS], f_src_b[TRK_MAX_FILTERS];  // operands of filter k
    int8_t f_ci[TRK_MAX_FILTERS][6];  // OUTSIDE_CI: sources of ml_j, lo_j, hi_j (j < f_ci_n <= 2)
    int8_t f_ci_n[TRK_MAX_FILTERS];
    int8_t dp_src;              // source of the DP/LC plane, -1: read per call
    uint32_t reg_filter_mask;   // filters all of whose operands are sources: evaluated on the registers
    uint32_t int_thr_mask;      // LT/GT on an int32 source with the threshold folded to an integer (f_ithr)
    int32_t f_ithr[TRK_MAX_FILTERS];
    int32_t delta_stride;       // LDS words per locus of the delta table (max_alleles + 4), 0 = no delta
    trk_call_out out;
};

// what one filtered call (called, now masked to no-call) removes from the locus counts
enum { DX_CALLED = 0, DX_LOW = 1, DX_HOML = 2, DX_HOMS = 3, DX_N = 4 };

// gtv: the call's haplotypes; tab: this locus's delta table (LDS) or nullptr -> global atomics
__device__ __forceinline__ void delta_filtered_call(const CallArgs& a, int l, const int* gtv, int pl, int32_t* tab,
                                                    int tab_alleles) {
    const int off = a.b.allele_off[l];
    const int A = a.b.allele_off[l + 1] - off;
    bool low = false, bad = false;
    for (int j = 0; j < pl; ++j) {
        const int v = gtv[j];
        low |= v == -2;
        bad |= v >= A;
        if (v >= 0 && v < A) {
            if (tab) atomicAdd(&tab[v], 1);
            else atomicSub(&a.out.delta_allele_count[off + v], 1);
        }
    }
    bool hl = false, hs = false;
    if (!low && !bad && pl >= 2) {
        int minl = 0x7fffffff, mins = 0x7fffffff, cl = 0, cs = 0;
        for (int j = 0; j < pl; ++j) {
            const int lc = a.b.len_class[off + gtv[j]], sc = a.b.str_class[off + gtv[j]];
            if (lc < minl) { minl = lc; cl = 1; } else if (lc == minl) { cl++; }
            if (sc < mins) { mins = sc; cs = 1; } else if (sc == mins) { cs++; }
        }
        hl = cl >= 2;
        hs = cs >= 2;
    }
    if (tab) {
        int32_t* x = tab + tab_alleles;
        atomicAdd(&x[DX_CALLED], 1);
        if (low) atomicAdd(&x[DX_LOW], 1);
        if (hl) atomicAdd(&x[DX_HOML], 1);
        if (hs) atomicAdd(&x[DX_HOMS], 1);
    } else {
        int32_t* li = a.out.delta_locus_int + (int64_t)l * TRK_LI_COLS;
        atomicSub(&li[TRK_LI_N_CALLED], 1);
        if (low) atomicSub(&li[TRK_LI_N_LOWPLOIDY], 1);
        if (hl) atomicSub(&li[TRK_LI_N_HOM_LEN], 1);
        if (hs) atomicSub(&li[TRK_LI_N_HOM_STR], 1);
    }
}

template <bool VEC>  // VEC: P == 2 and S % 4 == 0 -> every thread's 4 cells are one aligned 16-byte chunk
__global__ __launch_bounds__(CF_THREADS) void k_call_filter(const CallArgs a) {
    extern __shared__ uint32_t fcount[];  // [n_filters][CF_THREADS * CF_V]
    const int tid = threadIdx.x;
    const int S = a.b.n_samples, L = a.b.n_loci, P = a.b.ploidy;
    const int nf = a.n_filters;
    const int64_t s0 = ((int64_t)blockIdx.x * CF_THREADS + tid) * CF_V;
    const int l_begin = blockIdx.y * a.loci_per_block;
    const int l_end = min(L, l_begin + a.loci_per_block);
    for (int k = 0; k < nf; ++k)
#pragma unroll
        for (int j = 0; j < CF_V; ++j) fcount[(k * CF_THREADS + tid) * CF_V + j] = 0;
    uint32_t numcalls[CF_V] = {0, 0, 0, 0};
    uint32_t dpmiss[CF_V] = {0, 0, 0, 0};
    int64_t totaldp[CF_V] = {0, 0, 0, 0};
    if (s0 < S) {
        const int nvalid = (int)min((int64_t)CF_V, (int64_t)S - s0);
        for (int l = l_begin; l < l_end; ++l) {
            const int64_t cell0 = (int64_t)l * S + s0;
            const int pl = a.b.locus_ploidy ? min((int)a.b.locus_ploidy[l], P) : P;
            uint32_t w[CF_V];
            if (VEC) {
                u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.b.gt) + (cell0 >> 2));
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            }
            uint32_t mask[CF_V];
#pragma unroll
            for (int j = 0; j < CF_V; ++j) {
                mask[j] = 0;
                if (j >= nvalid) continue;
                const int64_t cell = cell0 + j;
                int gtv[TRK_MAX_PLOIDY];
                if (VEC) {
                    gtv[0] = (int16_t)(w[j] & 0xffffu);
                    gtv[1] = (int16_t)(w[j] >> 16);
                } else {
                    for (int q = 0; q < P && q < TRK_MAX_PLOIDY; ++q) gtv[q] = a.b.gt[cell * P + q];
                }
                bool miss = false;
                for (int q = 0; q < pl; ++q) miss |= gtv[q] == -1;
                const bool called = !miss;  // GetCalledSamples (dumpSTR.py:651)
                uint32_t m = 0;
                for (int k = 0; k < nf; ++k) {
                    if (eval_filter(a.filters[k], a.planes, cell, called, gtv, P, pl, a.n_cells)) {
                        m |= 1u << k;
                        if (called) fcount[(k * CF_THREADS + tid) * CF_V + j]++;  // dumpSTR.py:661
                    }
                }
                if (!called) m |= TRK_MASK_NOCALL;
                mask[j] = m;
                if (m == 0) {  // 'PASS'  (dumpSTR.py:686-713)
                    numcalls[j]++;
                    if (a.dp_plane >= 0 && p_is_f32(a.planes[a.dp_plane])) {
                        // Float depth (dumpSTR.py:688-713 on a float32 array): nan is neither negative nor
                        // positive; sums of float32 values are exact in float64 at these magnitudes, so the
                        // order of the atomics does not show
                        const trk_plane& dp = a.planes[a.dp_plane];
                        const float d = p_f32(dp, cell, 0, a.n_cells);
                        if (d < 0.f) {
                            if (atomicCAS(&a.out.error[0], 0, 1) == 0) {
                                a.out.error[1] = l;
                                a.out.error[2] = (int32_t)(s0 + j);
                            }
                        } else if (d > 0.f) {
                            atomicAdd(a.out.sample_totaldp_f64 + s0 + j, (double)d);
                        }
                    } else if (a.dp_plane >= 0) {
                        const trk_plane& dp = a.planes[a.dp_plane];
                        int32_t d = p_i32(dp, cell, 0, a.n_cells);
                        if (d == INT32_MIN) {
                            dpmiss[j]++;
                        } else if (d < 0) {
                            if (atomicCAS(&a.out.error[0], 0, 1) == 0) {
                                a.out.error[1] = l;
                                a.out.error[2] = (int32_t)(s0 + j);
                            }
                        } else {
                            totaldp[j] += d;
                        }
                    }
                } else if (called) {  // filtered call: genotype := no-call (dumpSTR.py:721-727)
                    if (a.out.delta_allele_count) delta_filtered_call(a, l, gtv, pl, nullptr, 0);
                    if (VEC) {
                        w[j] = 0xffffffffu;
                    } else if (a.out.gt_out) {
                        for (int q = 0; q < pl; ++q) gtv[q] = -1;
                    }
                }
                if (!VEC && a.out.gt_out)
                    for (int q = 0; q < P; ++q) a.out.gt_out[cell * P + q] = (int16_t)gtv[q];
            }
            if (VEC) {
                if (a.out.gt_out)
                    __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]},
                                                reinterpret_cast<u32x4*>(a.out.gt_out) + (cell0 >> 2));
                if (a.out.filter_mask)
                    __builtin_nontemporal_store(u32x4{mask[0], mask[1], mask[2], mask[3]},
                                                reinterpret_cast<u32x4*>(a.out.filter_mask) + (cell0 >> 2));
            } else if (a.out.filter_mask) {
                for (int j = 0; j < nvalid; ++j) a.out.filter_mask[cell0 + j] = mask[j];
            }
            if (a.out.filter_mask8)
                for (int j = 0; j < nvalid; ++j)
                    a.out.filter_mask8[cell0 + j] = (uint8_t)((mask[j] & 0x7fu) | ((mask[j] >> 24) & 0x80u));
        }
        // flush the per-sample counters of this block of loci
        for (int j = 0; j < nvalid; ++j) {
            const int64_t s = s0 + j;
            if (numcalls[j])
                atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_counters + s),
                          (unsigned long long)numcalls[j]);
            if (totaldp[j])
                atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_totaldp + s),
                          (unsigned long long)totaldp[j]);
            if (dpmiss[j])
                atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_dp_missing + s),
                          (unsigned long long)dpmiss[j]);
            for (int k = 0; k < nf; ++k) {
                uint32_t c = fcount[(k * CF_THREADS + tid) * CF_V + j];
                if (c)
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_counters + (int64_t)(1 + k) * S + s),
                              (unsigned long long)c);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_call_filter_fast : P == 2, S % 4 == 0.  Plane-major evaluation: the (up to
// four) single-column FORMAT planes are fetched once per locus as 16-byte
// vectors, then every simple threshold filter that reads plane p is evaluated
// on the registers.  Filters that need two planes or multi-column planes take
// the per-call path (eval_filter).  U loci are in flight per thread.
// ---------------------------------------------------------------------------
constexpr int CF_NSRC = 16;  // most vector sources of one launch; the kernel is built for 4, 8, 12 and 16

template <int NS>
struct CfLocus {
    u32x4 gt;
    u32x4 sv[NS];
};

// K vectors = 4 calls x K interleaved columns -> sources Q .. Q+K-1 (column c of call j is word j*K + c).
// Every register index is a template constant, so the sources stay in VGPRs.
template <int K, int NS, int Q>
__device__ __forceinline__ void cf_deinterleave(const u32x4* __restrict__ p, CfLocus<NS>& d) {
    if constexpr (Q + K <= NS) {
        u32x4 r[K];
#pragma unroll
        for (int i = 0; i < K; ++i) r[i] = __builtin_nontemporal_load(p + i);
#pragma unroll
        for (int c = 0; c < K; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) d.sv[Q + c][j] = r[(j * K + c) / 4][(j * K + c) % 4];
    }
}
// uniform dispatch of (first slot, k) to the static form
template <int NS, int Q>
__device__ __forceinline__ void cf_load_group(const u32x4* __restrict__ p, CfLocus<NS>& d, int base, int k) {
    if constexpr (Q < NS) {
        if (base == Q) {
            if (k == 2) cf_deinterleave<2, NS, Q>(p, d);
            else if (k == 3) cf_deinterleave<3, NS, Q>(p, d);
            else cf_deinterleave<4, NS, Q>(p, d);
        } else {
            cf_load_group<NS, Q + 1>(p, d, base, k);
        }
    }
}

template <int NS>
__device__ __forceinline__ void cf_load(const CallArgs& a, int64_t cell0, CfLocus<NS>& d) {
    d.gt = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.b.gt) + (cell0 >> 2));
#pragma unroll
    for (int q = 0; q < NS; ++q)
        if (q < a.n_src && a.src_ptr[q])
            d.sv[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.src_ptr[q]) + (cell0 >> 2));
    for (int g = 0; g < a.grp_n; ++g)   // interleaved planes
        cf_load_group<NS, 0>(reinterpret_cast<const u32x4*>(a.grp_ptr[g]) + (cell0 >> 2) * a.grp_k[g], d,
                             a.grp_base[g], a.grp_k[g]);
}

// the four values of source `idx` (uniform across the wave): static register indexing behind uniform guards
template <int NS>
__device__ __forceinline__ void cf_gather(const CfLocus<NS>& d, int idx, uint32_t* out) {
#pragma unroll
    for (int q = 0; q < NS; ++q)
        if (q == idx) {
            out[0] = d.sv[q][0];
            out[1] = d.sv[q][1];
            out[2] = d.sv[q][2];
            out[3] = d.sv[q][3];
        }
}

// Class LUT of a locus block in LDS, built by the whole workgroup: lutb[li][q] = len_class | str_class << 16 of
// allele q, linfo[li] = {A, flag, allele_off}; flag: some alleles share a class (max class + 1 < A), so that the
// homozygosity of a filtered call needs the LUT.  Ends with a barrier.
// Two global round trips per block whatever its size: (1) the block's allele offsets, (2) every class entry -- all
// of a thread's entries are fetched before the first is used, the duplicate test is a packed 16-bit max over the
// aligned lane group that owns a locus (no LDS atomics).  [The first form -- a loop over entries with the offset
// and the two class loads dependent inside every iteration plus two ds_max per entry -- cost a workgroup ~50 us:
// nothing at 100k loci, where other workgroups stream meanwhile, but at a strong-scaling shard of 12.5k loci
// every workgroup of the single round sits in its prologue at the same time: 0.76 ms against 0.54 ms without the
// delta outputs, profiles/r02_notes.md]
constexpr int CF_LINFO = 3;
constexpr int CF_LUT_UNROLL = 8;
template <bool STORE = true>   // STORE = false: only linfo (k_call_filter_v4 reads the classes of its rare duplicate-class
                                // loci from global memory when it drains its queue; lutb is not touched)
__device__ __forceinline__ void cf_build_lut(const trk_batch& b, int l_begin, int nl, int nal, int tid,
                                             uint32_t* lutb, int32_t* linfo) {
    for (int li = tid; li < nl; li += CF_THREADS) {
        const int o0 = b.allele_off[l_begin + li], o1 = b.allele_off[l_begin + li + 1];
        linfo[CF_LINFO * li] = o1 - o0;
        linfo[CF_LINFO * li + 1] = 0;
        linfo[CF_LINFO * li + 2] = o0;
    }
    __syncthreads();
    int gshift = 0;                       // lane group of a locus: the power of two >= nal
    while ((1 << gshift) < nal) ++gshift;
    if (gshift <= 6) {
        const int gsz = 1 << gshift, q = tid & (gsz - 1), lpp = CF_THREADS >> gshift;   // loci per pass
        for (int li0 = tid >> gshift; li0 < nl; li0 += lpp * CF_LUT_UNROLL) {
            uint32_t cls[CF_LUT_UNROLL];
#pragma unroll
            for (int u = 0; u < CF_LUT_UNROLL; ++u) {
                const int li = li0 + u * lpp;
                cls[u] = 0;
                if (li < nl && q < linfo[CF_LINFO * li]) {
                    const int e = linfo[CF_LINFO * li + 2] + q;
                    cls[u] = (uint32_t)b.len_class[e] | ((uint32_t)b.str_class[e] << 16);
                }
            }
#pragma unroll
            for (int u = 0; u < CF_LUT_UNROLL; ++u) {
                const int li = li0 + u * lpp;
                // uniform over the lane group, and a group never straddles a wave: the shuffles below are safe
                if (li >= nl) break;
                const int A = linfo[CF_LINFO * li];
                if (STORE && q < nal) lutb[li * nal + q] = cls[u];
                u16x2 m = __builtin_bit_cast(u16x2, cls[u]);
                for (int o = gsz >> 1; o > 0; o >>= 1) {
                    const uint32_t t = (uint32_t)__shfl_xor((int)__builtin_bit_cast(uint32_t, m), o, WAVE);
                    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, t));
                }
                if (q == 0) linfo[CF_LINFO * li + 1] = (((int)m.x + 1 < A) | ((int)m.y + 1 < A)) ? 1 : 0;
            }
        }
        __syncthreads();
        return;
    }
    if (STORE) {
        for (int i = tid; i < nl * nal; i += CF_THREADS) {     // allele sets beyond a wave: one entry at a time
            const int li = i / nal, q = i - li * nal;
            if (q >= linfo[CF_LINFO * li]) continue;
            const int e = linfo[CF_LINFO * li + 2] + q;
            lutb[i] = (uint32_t)b.len_class[e] | ((uint32_t)b.str_class[e] << 16);
        }
        __syncthreads();
    }
    // dense ranks: a duplicate exists iff the largest rank + 1 < A, i.e. iff no allele has rank A - 1
    for (int li = tid; li < nl; li += CF_THREADS) {
        const int A = linfo[CF_LINFO * li];
        bool top_l = false, top_s = false;
        for (int q = 0; q < A; ++q) {
            const int e = linfo[CF_LINFO * li + 2] + q;
            const uint32_t v = STORE ? lutb[li * nal + q] : ((uint32_t)b.len_class[e] | ((uint32_t)b.str_class[e] << 16));
            top_l |= (int)(v & 0xffffu) == A - 1;
            top_s |= (int)(v >> 16) == A - 1;
        }
        linfo[CF_LINFO * li + 1] = (A > 0 && !(top_l && top_s)) ? 1 : 0;
    }
    __syncthreads();
}
__device__ __forceinline__ bool cf_lut_needed(const int32_t* linfo, int li) { return linfo[CF_LINFO * li + 1] != 0; }

// per-locus delta context of the streaming call-filter kernel (all in LDS, filled at block start)
struct CfDelta {
    int32_t* tab;         // [max_alleles + DX_N] counts removed from this locus
    const uint32_t* lut;  // [max_alleles] lc | sc << 16
    int A;                // alleles of this locus
    int nal;              // max_alleles (offset of the DX_* entries)
    bool dup;             // some alleles share a class: homozygosity needs the LUT
};

// words after the allele bins of a locus's delta table: a bin for indices outside the locus's alleles,
// W0 = filtered calls | low-ploidy calls << 16, W1 = length-homozygous | sequence-homozygous << 16
enum { V2_TRASH = 0, V2_W0 = 1, V2_W1 = 2, V2_EXTRA = 3 };

// a called genotype that a filter masks to no-call: what it removes from the locus counts.  (The packed,
// branch-free form of k_call_filter_v4 costs this kernel five more VGPRs and a wave of occupancy.)
__device__ __forceinline__ void cf_delta_call(const CfDelta& c, uint32_t w, int pl) {
    const int a0 = (int)(int16_t)(w & 0xffffu);
    const int a1 = pl > 1 ? (int)(int16_t)(w >> 16) : -3;
    const bool v0 = (unsigned)a0 < (unsigned)c.A, v1 = (unsigned)a1 < (unsigned)c.A;
    if (v0) atomicAdd(&c.tab[a0], 1);
    if (v1) atomicAdd(&c.tab[a1], 1);
    int32_t* x = c.tab + c.nal;
    atomicAdd(&x[DX_CALLED], 1);
    if ((a0 == -2) | (a1 == -2)) {
        atomicAdd(&x[DX_LOW], 1);
    } else if (v0 & v1) {
        bool hl = a0 == a1, hs = hl;
        if (c.dup & !hl) {
            const uint32_t q = c.lut[a0] ^ c.lut[a1];
            hl = (q & 0xffffu) == 0u;
            hs = (q >> 16) == 0u;
        }
        if (hl) atomicAdd(&x[DX_HOML], 1);
        if (hs) atomicAdd(&x[DX_HOMS], 1);
    }
}

// ---- every operand in registers (ALLREG): sources as four vectors, decisions as lane masks -------------------------
// The sources of a thread's four calls are held as four NS-element vectors (c[j][q] = source q of call j): a filter
// picks its operands with a wave-uniform DYNAMIC index, which the compiler turns into one indexed v_mov each
// (s_set_gpr_idx).  The select chain of cf_gather costs 4 x NS instructions per operand; with nine filters the
// GangSTR set spent ~250 vector instructions per call on 60 bytes (SQ counters, profiles/r02_notes.md section 8).
// Decisions are wave-wide lane masks in scalar registers as in k_call_filter_v4.
template <int NS, bool GEN>
struct CfRegs {
    static constexpr int NLO = (NS < 8 ? NS : 8) * 4, NHI = (NS > 8 ? NS - 8 : 1) * 4;
    typedef uint32_t lo_t __attribute__((ext_vector_type(NLO)));
    typedef uint32_t hi_t __attribute__((ext_vector_type(NHI)));
    u32x4 gt;
    lo_t lo;   // element 4 q + j: source q (< 8) of call j -- a 16-byte load lands in four consecutive elements
    hi_t hi;   // sources 8 ..
    // element e of the (lo, hi) pair, e uniform across the wave: one indexed move
    __device__ __forceinline__ uint32_t elem(int e) const {
        if (NS <= 8 || e < NLO) return lo[e];
        return hi[e - NLO];
    }
    // The four values of logical source `idx`.  idx is the same in every lane, but where it comes out of a byte
    // table of the kernel arguments the compiler fetches it with a vector load and then wraps every indexed move in
    // a loop over the distinct values of the lanes -- readfirstlane tells it there is one.
    // GEN: interleaved planes among the sources (element offset and stride from the tables); else every source is
    // planar and sits in elements 4 idx .. 4 idx + 3
    __device__ __forceinline__ void get(const CallArgs& a, int idx_, uint32_t (&v)[CF_V]) const {
        const int idx = __builtin_amdgcn_readfirstlane(idx_);
        if constexpr (GEN) {
            const int off = __builtin_amdgcn_readfirstlane((int)a.src_off[idx]);
            const int st = __builtin_amdgcn_readfirstlane((int)a.src_stride[idx]);
#pragma unroll
            for (int j = 0; j < CF_V; ++j) v[j] = elem(off + j * st);
        } else if (NS <= 8 || idx < 8) {
#pragma unroll
            for (int j = 0; j < CF_V; ++j) v[j] = lo[4 * idx + j];
        } else {
#pragma unroll
            for (int j = 0; j < CF_V; ++j) v[j] = hi[4 * (idx - 8) + j];
        }
    }
};
// Every slot is loaded unconditionally -- a conditional insert makes the whole vector a phi and the register
// allocator copies it; slots past the last one re-read the genotype row (an L2 hit).
template <int NS, bool GEN>
__device__ __forceinline__ void cf_load_r(const CallArgs& a, int64_t cell0, CfRegs<NS, GEN>& d) {
    const int64_t c4 = cell0 >> 2;
    d.gt = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.b.gt) + c4);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const u32x4* sp = reinterpret_cast<const u32x4*>(a.b.gt) + c4;
        if (q < a.n_src) {
            if constexpr (GEN)
                sp = reinterpret_cast<const u32x4*>(a.slot_ptr[q]) + (c4 * a.slot_mult[q] + a.slot_chunk[q]);
            else
                sp = reinterpret_cast<const u32x4*>(a.slot_ptr[q]) + c4;
        }
        const u32x4 t = __builtin_nontemporal_load(sp);
#pragma unroll
        for (int j = 0; j < CF_V; ++j) {
            if (q < 8) d.lo[4 * q + j] = t[j];
            else d.hi[4 * (q - 8) + j] = t[j];
        }
    }
}

template <int NS, bool GEN>
__device__ __forceinline__ void cf_process_r(const CallArgs& a, int l, int64_t cell0, int64_t s0, int tid, bool leader,
                                             const CfRegs<NS, GEN>& d, uint32_t* fcount, uint32_t* numcalls,
                                             int64_t* totaldp, uint32_t* dpmiss, const CfDelta dc,
                                             const bool has_delta) {
    const int nf = a.n_filters;
    const int pl = a.b.locus_ploidy ? min((int)a.b.locus_ploidy[l], 2) : 2;
    const uint64_t livem = __ballot(1);
    uint32_t w[CF_V] = {d.gt[0], d.gt[1], d.gt[2], d.gt[3]};
    uint32_t mask[CF_V];
    uint64_t calledm[CF_V], anyhit[CF_V];
#pragma unroll
    for (int j = 0; j < CF_V; ++j) {
        uint64_t miss = __ballot((w[j] & 0xffffu) == 0xffffu);
        if (pl > 1) miss |= __ballot(w[j] >= 0xffff0000u);
        calledm[j] = livem & ~miss;
        mask[j] = __builtin_amdgcn_inverse_ballot_w64(calledm[j]) ? 0u : TRK_MASK_NOCALL;
        anyhit[j] = 0;
    }
    for (int k = 0; k < nf; ++k) {
        const trk_call_filter& f = a.filters[k];
        const int ia = a.f_src_a[k], ia2 = a.f_src_a2[k], ib = a.f_src_b[k];
        const bool af = ia >= 0 && ((a.src_f32_mask >> ia) & 1u);
        const float thrf = (float)f.thr;
        uint64_t hm[CF_V] = {0, 0, 0, 0};
        uint32_t oa[CF_V] = {0, 0, 0, 0}, oa2[CF_V] = {0, 0, 0, 0}, ob[CF_V] = {0, 0, 0, 0};
        if (ia >= 0) d.get(a, ia, oa);
        if (ia2 >= 0) d.get(a, ia2, oa2);
        if (ib >= 0) d.get(a, ib, ob);
        switch (f.op) {
            case TRK_F_LT:
            case TRK_F_CALLED_LT:
            case TRK_F_GT: {
                const bool gt_op = f.op == TRK_F_GT;
                if ((a.int_thr_mask >> k) & 1u) {
                    const int32_t ithr = a.f_ithr[k];
                    if (gt_op) {
#pragma unroll
                        for (int j = 0; j < CF_V; ++j) hm[j] = __ballot((int32_t)oa[j] > ithr);
                    } else {
#pragma unroll
                        for (int j = 0; j < CF_V; ++j) hm[j] = __ballot((int32_t)oa[j] < ithr);
                    }
                } else if (af) {   // numpy compares float32 arrays in float32
                    if (gt_op) {
#pragma unroll
                        for (int j = 0; j < CF_V; ++j) hm[j] = __ballot(__uint_as_float(oa[j]) > thrf);
                    } else {
#pragma unroll
                        for (int j = 0; j < CF_V; ++j) hm[j] = __ballot(__uint_as_float(oa[j]) < thrf);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) {
                        const double v = (double)(int32_t)oa[j];
                        hm[j] = gt_op ? __ballot(v > f.thr) : __ballot(v < f.thr);
                    }
                }
                if (f.op == TRK_F_CALLED_LT) {
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) hm[j] &= calledm[j];
                }
                break;
            }
            case TRK_F_RATIO_GT: {
                const bool bf = (a.src_f32_mask >> ib) & 1u;
#pragma unroll
                for (int j = 0; j < CF_V; ++j) {
                    const uint32_t xa = oa[j], xb = ob[j];
                    const double x = af ? (double)__uint_as_float(xa) : (double)(int32_t)xa;
                    const double y = bf ? (double)__uint_as_float(xb) : (double)(int32_t)xb;
                    hm[j] = __ballot((x / y) > f.thr);
                }
                break;
            }
            case TRK_F_CALLED_SUM_LT: {
#pragma unroll
                for (int j = 0; j < CF_V; ++j) {
                    const uint32_t xa = oa[j], xa2 = oa2[j];
                    if (af)
                        hm[j] = __ballot(__uint_as_float(xa) + __uint_as_float(xa2) < thrf);
                    else
                        hm[j] = __ballot((double)((int64_t)(int32_t)xa + (int64_t)(int32_t)xa2) < f.thr);
                    hm[j] &= calledm[j];
                }
                break;
            }
            case TRK_F_CALLED_EQ: {
#pragma unroll
                for (int j = 0; j < CF_V; ++j)
                    hm[j] = calledm[j] & __ballot((int32_t)oa[j] == (int32_t)ob[j]);
                break;
            }
            case TRK_F_CALLED_SUM_EQ: {
#pragma unroll
                for (int j = 0; j < CF_V; ++j)
                    hm[j] = calledm[j] & __ballot((int64_t)(int32_t)oa[j] + (int64_t)(int32_t)oa2[j] ==
                                                  (int64_t)(int32_t)ob[j]);
                break;
            }
            case TRK_F_CALLED_OUTSIDE_CI: {
                for (int c = 0; c < a.f_ci_n[k]; ++c) {
                    uint32_t vm[CF_V], vl[CF_V], vh[CF_V];
                    d.get(a, a.f_ci[k][3 * c], vm);
                    d.get(a, a.f_ci[k][3 * c + 1], vl);
                    d.get(a, a.f_ci[k][3 * c + 2], vh);
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) {
                        const int32_t ml = (int32_t)vm[j];
                        hm[j] |= calledm[j] & (__ballot(ml < (int32_t)vl[j]) | __ballot((int32_t)vh[j] < ml));
                    }
                }
                break;
            }
            default:
                break;
        }
        const uint32_t bit = 1u << k;
        uint32_t inc[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < CF_V; ++j) {
            mask[j] |= __builtin_amdgcn_inverse_ballot_w64(hm[j]) ? bit : 0u;
            anyhit[j] |= hm[j];
            inc[j >> 1] |= __builtin_amdgcn_inverse_ballot_w64(hm[j] & calledm[j]) ? (1u << (16 * (j & 1))) : 0u;
        }
        // per-thread counters, two samples per LDS word (a block holds <= 4096 loci: 16 bits are enough)
        uint32_t* fc = fcount + (k * CF_THREADS + tid) * 2;
        atomicAdd(&fc[0], inc[0]);
        atomicAdd(&fc[1], inc[1]);
    }
    uint64_t passm[CF_V], filtm[CF_V];
#pragma unroll
    for (int j = 0; j < CF_V; ++j) {
        passm[j] = calledm[j] & ~anyhit[j];  // mask word == 0: dumpSTR.py:686
        filtm[j] = calledm[j] & anyhit[j];   // dumpSTR.py:715-727
        add_mask(numcalls[j], passm[j]);
    }
    if (a.dp_plane >= 0) {
        uint64_t bad = 0;
        uint32_t dpv[CF_V];
        d.get(a, a.dp_src, dpv);
#pragma unroll
        for (int j = 0; j < CF_V; ++j) {
            const int32_t dv = (int32_t)dpv[j];
            add_mask(dpmiss[j], passm[j] & __ballot(dv == INT32_MIN));
            const int32_t dpos = dv > 0 ? dv : 0;
            totaldp[j] += __builtin_amdgcn_inverse_ballot_w64(passm[j]) ? dpos : 0;
            bad |= passm[j] & __ballot((uint32_t)dv > 0x80000000u);   // negative, not the missing marker
        }
        if (bad) {  // a negative depth on a call that passes (cold): dumpSTR.py:698-706
#pragma unroll
            for (int j = CF_V - 1; j >= 0; --j) {
                const int32_t dv = (int32_t)dpv[j];
                if ((mask[j] == 0u) & (dv < 0) & (dv != INT32_MIN)) {
                    if (atomicCAS(&a.out.error[0], 0, 1) == 0) {
                        a.out.error[1] = l;
                        a.out.error[2] = (int32_t)(s0 + j);
                    }
                }
            }
        }
    }
    if (has_delta) {
        // what the filtered calls remove from the locus counts: allele bins per call, the four per-locus counters
        // from the masks -- one LDS atomic each per wave and locus
        uint32_t n_filt = 0, n_low = 0, n_hl = 0, n_hs = 0;
        const uint32_t A = (uint32_t)dc.A;
#pragma unroll
        for (int j = 0; j < CF_V; ++j) {
            const uint32_t a0 = w[j] & 0xffffu, a1 = pl > 1 ? (w[j] >> 16) : 0xfffdu;   // unsigned halves: -2 = 0xfffe
            const bool filtered = __builtin_amdgcn_inverse_ballot_w64(filtm[j]);
            const bool v0 = a0 < A, v1 = a1 < A;
            if (filtered) {
                if (v0) atomicAdd(&dc.tab[a0], 1);
                if (v1) atomicAdd(&dc.tab[a1], 1);
            }
            const uint64_t lowm = filtm[j] & (__ballot(a0 == 0xfffeu) | __ballot(a1 == 0xfffeu));
            const uint64_t bothm = filtm[j] & ~lowm & __ballot(v0) & __ballot(v1);
            uint64_t hlm = bothm & __ballot(a0 == a1), hsm = hlm;
            if (dc.dup) {   // alleles that share a class (uniform per locus, rare): the LUT decides for a0 != a1
                const bool need = __builtin_amdgcn_inverse_ballot_w64(bothm & ~hlm);
                uint32_t q = 0xffffffffu;
                if (need) q = dc.lut[a0] ^ dc.lut[a1];
                hlm |= __ballot((q & 0xffffu) == 0u);
                hsm |= __ballot((q >> 16) == 0u);
            }
            n_filt += (uint32_t)__popcll(filtm[j]);
            n_low += (uint32_t)__popcll(lowm);
            n_hl += (uint32_t)__popcll(hlm);
            n_hs += (uint32_t)__popcll(hsm);
        }
        if (leader) {
            int32_t* x = dc.tab + dc.nal;
            if (n_filt) atomicAdd(&x[DX_CALLED], (int32_t)n_filt);
            if (n_low) atomicAdd(&x[DX_LOW], (int32_t)n_low);
            if (n_hl) atomicAdd(&x[DX_HOML], (int32_t)n_hl);
            if (n_hs) atomicAdd(&x[DX_HOMS], (int32_t)n_hs);
        }
    }
#pragma unroll
    for (int j = 0; j < CF_V; ++j)
        if (__builtin_amdgcn_inverse_ballot_w64(filtm[j])) w[j] = pl > 1 ? 0xffffffffu : (w[j] | 0xffffu);
    if (a.out.gt_out)
        __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]},
                                    reinterpret_cast<u32x4*>(a.out.gt_out) + (cell0 >> 2));
    if (a.out.filter_mask)
        __builtin_nontemporal_store(u32x4{mask[0], mask[1], mask[2], mask[3]},
                                    reinterpret_cast<u32x4*>(a.out.filter_mask) + (cell0 >> 2));
    if (a.out.filter_mask8) {
        uint32_t m8 = 0;
#pragma unroll
        for (int j = 0; j < CF_V; ++j) m8 |= ((mask[j] & 0x7fu) | ((mask[j] >> 24) & 0x80u)) << (8 * j);
        __builtin_nontemporal_store(m8, reinterpret_cast<uint32_t*>(a.out.filter_mask8) + (cell0 >> 2));
    }
}

template <int NS, bool ALLREG>
__device__ __forceinline__ void cf_process(const CallArgs& a, int l, int64_t cell0, int64_t s0, int tid,
                                           const CfLocus<NS>& d, uint32_t* fcount, uint32_t* numcalls,
                                           int64_t* totaldp, uint32_t* dpmiss, const CfDelta dc,
                                           const bool has_delta) {
    const int nf = a.n_filters;
    const int pl = a.b.locus_ploidy ? min((int)a.b.locus_ploidy[l], 2) : 2;
    uint32_t w[CF_V] = {d.gt[0], d.gt[1], d.gt[2], d.gt[3]};
    uint32_t mask[CF_V];
    bool called[CF_V];
#pragma unroll
    for (int j = 0; j < CF_V; ++j) {
        const bool m0 = (w[j] & 0xffffu) == 0xffffu;
        const bool m1 = (pl > 1) & ((w[j] >> 16) == 0xffffu);
        called[j] = !(m0 | m1);
        mask[j] = called[j] ? 0u : TRK_MASK_NOCALL;
    }
    for (int k = 0; k < nf; ++k) {
        const trk_call_filter& f = a.filters[k];
        bool hit[CF_V];
        if (ALLREG || ((a.reg_filter_mask >> k) & 1u)) {
            // every operand is a vector source already in registers
            const bool af = (a.src_f32_mask >> a.f_src_a[k]) & 1u;
            uint32_t oa[CF_V] = {0, 0, 0, 0}, oa2[CF_V] = {0, 0, 0, 0}, ob[CF_V] = {0, 0, 0, 0};
            if (a.f_src_a[k] >= 0) cf_gather(d, a.f_src_a[k], oa);
            if (a.f_src_a2[k] >= 0) cf_gather(d, a.f_src_a2[k], oa2);
            if (a.f_src_b[k] >= 0) cf_gather(d, a.f_src_b[k], ob);
            const float thrf = (float)f.thr;
            switch (f.op) {
                case TRK_F_LT:
                case TRK_F_CALLED_LT:
                case TRK_F_GT: {
                    const bool gt_op = f.op == TRK_F_GT, need = f.op == TRK_F_CALLED_LT;
                    const bool ithr_ok = (a.int_thr_mask >> k) & 1u;
                    const int32_t ithr = a.f_ithr[k];
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) {
                        bool h;
                        if (ithr_ok) {
                            const int32_t v = (int32_t)oa[j];
                            h = gt_op ? (v > ithr) : (v < ithr);
                        } else if (af) {
                            const float v = __uint_as_float(oa[j]);
                            h = gt_op ? (v > thrf) : (v < thrf);   // numpy compares float32 arrays in float32
                        } else {
                            const double v = (double)(int32_t)oa[j];
                            h = gt_op ? (v > f.thr) : (v < f.thr);
                        }
                        hit[j] = h & (called[j] | !need);
                    }
                    break;
                }
                case TRK_F_RATIO_GT: {
                    const bool bf = (a.src_f32_mask >> a.f_src_b[k]) & 1u;
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) {
                        const double x = af ? (double)__uint_as_float(oa[j]) : (double)(int32_t)oa[j];
                        const double y = bf ? (double)__uint_as_float(ob[j]) : (double)(int32_t)ob[j];
                        hit[j] = (x / y) > f.thr;
                    }
                    break;
                }
                case TRK_F_CALLED_SUM_LT: {
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) {
                        bool h;
                        if (af) {
                            const float sum = __uint_as_float(oa[j]) + __uint_as_float(oa2[j]);
                            h = sum < thrf;
                        } else {
                            h = (double)((int64_t)(int32_t)oa[j] + (int64_t)(int32_t)oa2[j]) < f.thr;
                        }
                        hit[j] = h & called[j];
                    }
                    break;
                }
                case TRK_F_CALLED_EQ: {
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) hit[j] = called[j] & ((int32_t)oa[j] == (int32_t)ob[j]);
                    break;
                }
                case TRK_F_CALLED_SUM_EQ: {
#pragma unroll
                    for (int j = 0; j < CF_V; ++j)
                        hit[j] = called[j] & ((int64_t)(int32_t)oa[j] + (int64_t)(int32_t)oa2[j] == (int64_t)(int32_t)ob[j]);
                    break;
                }
                case TRK_F_CALLED_OUTSIDE_CI: {
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) hit[j] = false;
                    for (int c = 0; c < a.f_ci_n[k]; ++c) {
                        uint32_t ml[CF_V] = {0, 0, 0, 0}, lo[CF_V] = {0, 0, 0, 0}, hi[CF_V] = {0, 0, 0, 0};
                        cf_gather(d, a.f_ci[k][3 * c], ml);
                        cf_gather(d, a.f_ci[k][3 * c + 1], lo);
                        cf_gather(d, a.f_ci[k][3 * c + 2], hi);
#pragma unroll
                        for (int j = 0; j < CF_V; ++j)
                            hit[j] |= called[j] & (((int32_t)ml[j] < (int32_t)lo[j]) | ((int32_t)hi[j] < (int32_t)ml[j]));
                    }
                    break;
                }
                default:
#pragma unroll
                    for (int j = 0; j < CF_V; ++j) hit[j] = false;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CF_V; ++j) {
                int gtv[2] = {(int)(int16_t)(w[j] & 0xffffu), (int)(int16_t)(w[j] >> 16)};
                hit[j] = eval_filter(f, a.planes, cell0 + j, called[j], gtv, 2, pl, a.n_cells);
            }
        }
#pragma unroll
        for (int j = 0; j < CF_V; ++j) mask[j] |= hit[j] ? (1u << k) : 0u;
        // per-thread counters, two samples per LDS word (a block holds <= 4096 loci: 16 bits are enough)
        uint32_t* fc = fcount + (k * CF_THREADS + tid) * 2;
        atomicAdd(&fc[0], (uint32_t)(hit[0] & called[0]) | ((uint32_t)(hit[1] & called[1]) << 16));
        atomicAdd(&fc[1], (uint32_t)(hit[2] & called[2]) | ((uint32_t)(hit[3] & called[3]) << 16));
    }
    uint32_t dpv[CF_V] = {0, 0, 0, 0};
    if (a.dp_src >= 0) cf_gather(d, a.dp_src, dpv);
#pragma unroll
    for (int j = 0; j < CF_V; ++j) {
        if (mask[j] == 0) {  // dumpSTR.py:686
            numcalls[j]++;
            if (a.dp_plane >= 0) {
                const int32_t dv = (ALLREG || a.dp_src >= 0) ? (int32_t)dpv[j]
                                                             : p_i32(a.planes[a.dp_plane], cell0 + j, 0, a.n_cells);
                if (dv == INT32_MIN) {
                    dpmiss[j]++;
                } else if (dv < 0) {  // dumpSTR.py:698-706
                    if (atomicCAS(&a.out.error[0], 0, 1) == 0) {
                        a.out.error[1] = l;
                        a.out.error[2] = (int32_t)(s0 + j);
                    }
                } else {
                    totaldp[j] += dv;
                }
            }
        } else if (called[j]) {  // dumpSTR.py:715-727
            if (has_delta) cf_delta_call(dc, w[j], pl);
            w[j] = pl > 1 ? 0xffffffffu : (w[j] | 0xffffu);
        }
    }
    if (a.out.gt_out)
        __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]},
                                    reinterpret_cast<u32x4*>(a.out.gt_out) + (cell0 >> 2));
    if (a.out.filter_mask)
        __builtin_nontemporal_store(u32x4{mask[0], mask[1], mask[2], mask[3]},
                                    reinterpret_cast<u32x4*>(a.out.filter_mask) + (cell0 >> 2));
    if (a.out.filter_mask8) {
        uint32_t m8 = 0;
#pragma unroll
        for (int j = 0; j < CF_V; ++j) m8 |= ((mask[j] & 0x7fu) | ((mask[j] >> 24) & 0x80u)) << (8 * j);
        __builtin_nontemporal_store(m8, reinterpret_cast<uint32_t*>(a.out.filter_mask8) + (cell0 >> 2));
    }
}

// ALLREG: every filter and the depth plane read vector sources only (the per-call path is compiled out).
// A workgroup owns loci [y * loci_per_wg, ...) and walks them in sub-blocks of loci_per_block, the unit of the
// LDS delta table; the per-sample counters live across sub-blocks and are flushed once.
// (the instantiation that sits one register above 128 VGPRs is held to four waves per SIMD)
// VEC (with ALLREG): the sources sit in indexable register vectors, cf_process_r -- 1: every source planar, 2: interleaved
// planes among them (offsets and strides from tables); 0: the select-chain form.
template <int NS, bool ALLREG, int VEC>
__global__ __launch_bounds__(CF_THREADS, (NS == 12 && ALLREG) ? 4 : 1) void k_call_filter_fast(const CallArgs a) {
    extern __shared__ uint32_t fcount[];  // [n_filters][CF_THREADS][2] (16-bit pairs), then the delta table
    const int tid = threadIdx.x;
    const int S = a.b.n_samples, L = a.b.n_loci;
    const int nf = a.n_filters;
    const int64_t s0 = ((int64_t)blockIdx.x * CF_THREADS + tid) * CF_V;
    const bool live = s0 < S;  // S % 4 == 0: a thread's 4 samples are all in range or all out
    const int wg_begin = blockIdx.y * a.loci_per_wg;
    const int wg_end = min(L, wg_begin + a.loci_per_wg);
    for (int k = 0; k < nf; ++k) fcount[(k * CF_THREADS + tid) * 2] = fcount[(k * CF_THREADS + tid) * 2 + 1] = 0;
    // delta context in LDS: table [loci][max_alleles + 4], class LUT [loci][max_alleles], per-locus (A, dup)
    const int dstride = a.delta_stride;
    const int nal = dstride - DX_N;
    int32_t* const dbase = reinterpret_cast<int32_t*>(fcount + (size_t)nf * CF_THREADS * 2);
    uint32_t* const lutb = reinterpret_cast<uint32_t*>(dbase + (size_t)a.loci_per_block * dstride);
    int32_t* const linfo = reinterpret_cast<int32_t*>(lutb + (size_t)a.loci_per_block * nal);
    uint32_t numcalls[CF_V] = {0, 0, 0, 0};
    uint32_t dpmiss[CF_V] = {0, 0, 0, 0};
    int64_t totaldp[CF_V] = {0, 0, 0, 0};
    for (int l_begin = wg_begin; l_begin < wg_end; l_begin += a.loci_per_block) {
        const int l_end = min(wg_end, l_begin + a.loci_per_block);
        const int nl = l_end - l_begin;
        if (dstride) {
            for (int i = tid; i < nl * dstride; i += CF_THREADS) dbase[i] = 0;
            cf_build_lut(a.b, l_begin, nl, nal, tid, lutb, linfo);
        }
        if (live) {
            auto delta_of = [&](int l) {
                CfDelta dc = {nullptr, nullptr, 0, 0, false};
                if (dstride) {
                    const int li = l - l_begin;
                    dc.tab = dbase + li * dstride;
                    dc.lut = lutb + li * nal;
                    dc.A = linfo[CF_LINFO * li];
                    dc.dup = cf_lut_needed(linfo, li);
                    dc.nal = nal;
                }
                return dc;
            };
            // (loading locus l + 1 into a second register set while l is evaluated was measured: the registers
            // cost more occupancy than the overlap buys, profiles/r01_notes.md)
            if constexpr (ALLREG && VEC != 0) {
                const bool leader = (tid & 63) == __ffsll((unsigned long long)__ballot(1)) - 1;  // first live lane
                for (int l = l_begin; l < l_end; ++l) {
                    CfRegs<NS, VEC == 2> d;
                    cf_load_r(a, (int64_t)l * S + s0, d);
                    cf_process_r<NS, VEC == 2>(a, l, (int64_t)l * S + s0, s0, tid, leader, d, fcount, numcalls, totaldp, dpmiss,
                                     delta_of(l), dstride != 0);
                }
            } else {
                for (int l = l_begin; l < l_end; ++l) {
                    CfLocus<NS> d;
                    cf_load(a, (int64_t)l * S + s0, d);
                    cf_process<NS, ALLREG>(a, l, (int64_t)l * S + s0, s0, tid, d, fcount, numcalls, totaldp, dpmiss,
                                           delta_of(l), dstride != 0);
                }
            }
        }
        if (dstride) {  // flush the sub-block's delta table: one global atomic per non-zero entry
            __syncthreads();
            for (int i = tid; i < nl * dstride; i += CF_THREADS) {
                    const int v = dbase[i];
                    if (!v) continue;
                    const int li_f = i / dstride;
                    const int l = l_begin + li_f;
                    const int r = i - li_f * dstride;
                    if (r < nal) {
                        atomicSub(&a.out.delta_allele_count[linfo[CF_LINFO * li_f + 2] + r], v);
                    } else {
                        const int x = r - nal;
                        const int col = x == DX_CALLED ? TRK_LI_N_CALLED
                                        : x == DX_LOW  ? TRK_LI_N_LOWPLOIDY
                                        : x == DX_HOML ? TRK_LI_N_HOM_LEN
                                                       : TRK_LI_N_HOM_STR;
                        atomicSub(&a.out.delta_locus_int[(int64_t)l * TRK_LI_COLS + col], v);
                    }
                }
            __syncthreads();
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < CF_V; ++j) {
            const int64_t s = s0 + j;
            if (numcalls[j])
                atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_counters + s),
                          (unsigned long long)numcalls[j]);
            if (totaldp[j])
                atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_totaldp + s),
                          (unsigned long long)totaldp[j]);
            if (dpmiss[j])
                atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_dp_missing + s),
                          (unsigned long long)dpmiss[j]);
            for (int k = 0; k < nf; ++k) {
                const uint32_t c = (fcount[(k * CF_THREADS + tid) * 2 + (j >> 1)] >> (16 * (j & 1))) & 0xffffu;
                if (c)
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_counters + (int64_t)(1 + k) * S + s),
                              (unsigned long long)c);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Launch geometry of the column-owner call-filter kernels (k_call_filter_v4 / _gs): a workgroup owns one column tile
// (CF_THREADS x CF_V samples) and ONE contiguous range of `walk` locus blocks (sub-blocks of loci_per_block loci: the
// LDS delta table is rebuilt block by block while the per-sample counters stay in registers over the whole range).
//   map 0   grid (gx, n_ranges): column tiles are neighbours in launch order
//   map 1   grid (n_ranges, gx): locus-major launch order
//   map 2   1-D grid, XCD-aware: block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"), so
//           range = b % 8 + 8 * ((b / 8) / gx), tile = (b / 8) % gx puts the gx tiles of a range -- whole rows -- on
//           ONE XCD and gives each XCD every eighth range.
// What the round-4 sweeps with the output planes' placement pinned say (profiles/r04_notes.md section 1): the bare
// stream of this shape is worth 3.3-3.45 ms at 100k x 10k whatever the map, the block size or the occupancy (+-3 %);
// persistent ranges save k_cf_reduce its rows (0.075 -> 0.015 ms) and the product kernel 1-3 %; the 10-18 % between
// a fast and a slow PAIR of output planes (r03_notes section 22) is not a matter of geometry.
// A speed choice only: every workgroup computes the same thing wherever it runs.
// ---------------------------------------------------------------------------
struct CfGeom {
    int gx, n_ranges, walk, map;
};
__device__ __forceinline__ bool cf_place(const CfGeom& g, int& tile, int& range) {
    if (g.map == 2) {
        const int slot = (int)blockIdx.x >> 3;
        range = ((int)blockIdx.x & 7) + 8 * (slot / g.gx);
        tile = slot % g.gx;
    } else if (g.map == 1) {
        range = blockIdx.x;
        tile = blockIdx.y;
    } else {
        tile = blockIdx.x;
        range = blockIdx.y;
    }
    return range < g.n_ranges;
}

constexpr int V2_MAX_FILTERS = 6;
// ---------------------------------------------------------------------------
// k_call_filter_v4 : the streaming kernel for the common dumpSTR filter sets -- every filter a plain threshold on a
// single-column plane (min / max DP, min Q, pre-parsed min supporting reads ...) or a HipSTR ratio over the depth
// plane; diploid records, S % 4 == 0.  Column-owner tiling: a thread owns four consecutive samples and walks the loci
// of its workgroup's range, the per-sample counters of dumpSTR's sample_info in registers (add-with-carry, the
// wave-wide decision mask as the carry), decisions as 64-bit lane masks in scalar registers.
// Round 4 rewrote the round-2/3 kernel (k_call_filter_v2, git history) for a third of its instructions: the v2 loop issued 246 vector + 358 scalar instructions per wave and locus (tools/loop_census.py): the filter
// kind was a run-time switch (57 branches, four duplicated compare arms per filter), the kernel arguments were
// re-read from the constant cache every iteration (14 s_load), twelve scalars lived in vector lanes, and the delta
// block -- what a FILTERED call removes from its locus's counts -- ran its ~20 vector instructions and two LDS
// atomics per call slot for the two or three lanes of a wave that have a filtered call.  Here:
//   * the compare TYPE is static: the host orders the filters integer planes first, float planes last, and NFLT
//     (the number of trailing float filters) is a template argument; LT / GT is folded into the operands --
//     integers: x > t <=> !(x < t + 1), the inversion applied to the wave-wide lane mask (one s_xor); floats:
//     x > t <=> -x < -t, the sign flipped on the value (exact, NaN stays false);
//   * everything about call j is finished before call j + 1 starts, so only a handful of 64-bit lane masks are live
//     at any time (no spills, no argument re-loads);
//   * FILTERED CALLS ARE QUEUED, not handled where they are met (the trick of k_assoc_scan_few's missing-call
//     queue): a wave appends {genotype word, locus} to its own LDS queue (ballot rank = v_mbcnt, one masked
//     ds_write_b64 per call slot) and drains the queue in bulk -- lane = queued call, 64 at a time, the allele /
//     called / homozygosity updates of the block's delta table as before -- when it runs full and at the end of
//     the block.  The drain's instructions serve 64 calls instead of two.
// 184 vector + 167 scalar instructions per wave and locus (v2: 246 + 358); same-process A/B at 100k x 10k: 3.58 ->
// 3.46 ms, every locus and 1.2e8 calls bit for bit against the compiled oracle under both (profiles/r04_notes.md).
// ---------------------------------------------------------------------------
struct V4Filter {
    const void* plane;    // [L,S] int32 or float32
    int32_t thr;          // integer: fires when (x < thr) != invert; float: bits of the (sign-flipped) threshold
    uint32_t flip;        // float: 0x80000000 for GT (the value's sign is flipped), else 0
    int32_t invert;       // integer GT: thr = t + 1 and the lane mask is inverted
    int32_t need_called;  // TRK_F_CALLED_LT
    int32_t bit;          // bit of this filter in the mask / row of sample_counters - 1
    int32_t ratio;        // 1: HipSTR-style ratio over the depth plane: (double)x / (double)depth > dthr
    int32_t is_float;     // the plane's type (read by the NFLT = -1 builds only: there the type is a run-time flag)
    int32_t pad;
    double dthr;
    double dthr_up;       // ratio: the next double above dthr (x / d above it rounds to a quotient above dthr)
};
constexpr int V4_QCAP = 320;            // queue entries per wave: drained when fewer than 4 x 64 are free
constexpr int V4_PD = 2;                // input sets of the compact build's prefetch ring (k_call_filter_v4, OUT == 2)
constexpr int V4_CV_MAXNF = 0;          // filters up to which the compact build takes two chunks per thread: none (see the kernel's header)
struct V4Args {
    trk_batch b;
    V4Filter f[V2_MAX_FILTERS];
    const int32_t* dp;
    int loci_per_block;
    int delta_nal;
    CfGeom geom;
    uint16_t* part16;
    unsigned long long* part64;
    trk_call_out out;
};

// one queued call: what it removes from its locus's counts (dumpSTR.py:715-727 masks it to a no-call)
__device__ __forceinline__ void v4_drain_one(uint32_t w, uint32_t li, uint32_t* dtab, const trk_batch& b,
                                             const int32_t* linfo, int nal, int dstride) {
    uint32_t* tab = dtab + li * dstride;
    const uint32_t A = (uint32_t)linfo[CF_LINFO * li];
    const uint32_t a0 = w & 0xffffu, a1 = w >> 16;   // unsigned halves: -1 = 0xffff, -2 = 0xfffe
    const bool v0 = a0 < A, v1 = a1 < A;             // (-2 / out of range: not below A)
    atomicAdd(&tab[v0 ? (int)a0 : nal + V2_TRASH], 1u);
    atomicAdd(&tab[v1 ? (int)a1 : nal + V2_TRASH], 1u);
    const bool low = (a0 == 0xfffeu) | (a1 == 0xfffeu);
    // homozygous by index (same allele twice); by length / sequence class only where the locus has alleles that
    // share a class (rare): then the LUT decides for a0 != a1
    bool hl = (a0 == a1) & v0, hs = hl;
    if (v0 && v1 && !hl && cf_lut_needed(linfo, (int)li)) {
        const int o = linfo[CF_LINFO * li + 2];   // (rare: the classes come from global memory, not from an LDS copy)
        hl = b.len_class[o + a0] == b.len_class[o + a1];
        hs = b.str_class[o + a0] == b.str_class[o + a1];
    }
    atomicAdd(&tab[nal + V2_W0], 1u + (low ? 0x10000u : 0u));
    const uint32_t w1 = (hl ? 1u : 0u) + (hs ? 0x10000u : 0u);
    if (w1) atomicAdd(&tab[nal + V2_W1], w1);
}

// OUT: the output set as a template argument.  0: whichever of gt_out / filter_mask / filter_mask8 the caller set (run-time
// pointer tests, wave-uniform); 2: filter_mask8 ALONE -- what dumpSTR's command line asks for (compute.dumpstr_batch): no
// masked-genotype words, no 32-bit mask, the four bytes of a chunk assembled directly, and the NEXT locus's loads in
// flight while this one is worked on.  With 13 B per call instead of 20 the pass is no longer bound by the memory
// system but by how many bytes four waves per SIMD keep in flight (one locus each: the loop waited for its loads
// before it started on them); the 20 B builds gain nothing from the prefetch (r04_notes section 2) and do not carry it.
// gt_out == b.gt (IN PLACE, OUT == 0): the genotype tensor is updated where it lies, as the reference does with its
// record (dumpSTR.py:721-727) -- only the 16-byte chunks that hold a filtered call are written, the second write stream
// of the pass nearly vanishes (and with it the output-pair effect of profiles/r03_notes.md section 22).
// CV: 16-byte chunks (four samples each) a thread owns -- chunk c of thread t is chunk c * CF_THREADS + t of the tile, so
// that every load and store instruction of a wave stays one contiguous kilobyte.  ONE in every build that is launched.
// Why it exists: the compact build's one byte per call is a 4-byte store per lane and locus, and the bare stream of that
// shape -- three input planes and a million one-kilobyte row segments of mask bytes -- takes 2.62 ms where the three
// planes alone take 1.79 (tools/stream_probe pinned_probe `compact`, profiles/r05_pinned_compact.txt: eight per cent more
// bytes, 46 % more time; 16-byte stores exchanged through LDS 2.52, deeper prefetch nothing).  Two chunks per thread
// halve the number of segments; same-process A/B of the product kernel (tools/cf_variants_probe.py, TRK_CF_CV2):
// 3.15 ms against 2.74 with one chunk -- twice the counters and inputs in registers cost the occupancy more than the
// wider segments bring.  The compact pass runs at 0.95 of its own stream; the stream's shape is the bound.
template <int NF, int NFLT, bool DELTA, bool RATIO, int ALIAS, int OUT = 0, int CV = (OUT == 2 && NF <= V4_CV_MAXNF ? 2 : 1)>
__global__ __launch_bounds__(CF_THREADS) void k_call_filter_v4(const V4Args a) {
    extern __shared__ uint32_t v2lds[];
    const int tid = threadIdx.x;
    const int S = a.b.n_samples, L = a.b.n_loci;
    int bix, biy;
    if (!cf_place(a.geom, bix, biy)) return;
    // first sample of chunk c; a chunk beyond the row's end (the last tile's tail) reads chunk 0's cells and has no effect
    int64_t s0c[CV];
    uint64_t lm[CV];
    bool livec[CV];
#pragma unroll
    for (int c = 0; c < CV; ++c) {
        const int64_t sc = (((int64_t)bix * CV + c) * CF_THREADS + tid) * CF_V;
        livec[c] = sc < S;
        s0c[c] = livec[c] ? sc : ((int64_t)bix * CV * CF_THREADS + tid) * CF_V;
    }
    const int64_t s0 = s0c[0];
    const int nal = a.delta_nal;
    const int dstride = nal + V2_EXTRA;
    uint32_t* dtab = v2lds;                                        // [loci][nal + 3]
    int32_t* linfo = reinterpret_cast<int32_t*>(dtab + (size_t)a.loci_per_block * dstride);  // [loci][3]
    // this wave's queue [V4_QCAP] of {genotype word, locus of the block}, behind the tables on an 8-byte boundary (the
    // wave's base as a scalar: a push's address is one shift-and-add on the lane's rank)
    const uint32_t qbase_w = (uint32_t)(((((size_t)a.loci_per_block * (dstride + CF_LINFO)) + 1) & ~(size_t)1)) +
                             (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6) * (uint32_t)(2 * V4_QCAP);
    uint2* queue = reinterpret_cast<uint2*>(v2lds + qbase_w);
    struct Acc {
        uint32_t numcalls[CF_V], dpmiss[CF_V];
        uint32_t fc[NF][CF_V];
        int64_t totaldp[CF_V];
    };
    Acc acc[CV] = {};
    uint64_t nn[NF], inv[NF];   // all lanes when the filter also applies to calls that are not made / when inverted
    uint32_t bitv[NF];          // the filter's bit of the mask word, in a vector register (a select's constant)
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        nn[k] = a.f[k].need_called == 0 ? ~0ull : 0ull;
        inv[k] = a.f[k].invert ? ~0ull : 0ull;
        bitv[k] = 1u << a.f[k].bit;
        asm volatile("" : "+v"(bitv[k]));
    }
    // the no-call flag: bit 31 of the mask word, bit 7 of the mask byte
    uint32_t nocallv = OUT == 2 ? 0x80u : TRK_MASK_NOCALL;
    asm volatile("" : "+v"(nocallv));
    uint32_t qtail = 0;         // wave-uniform
    const bool live = livec[0];
    const bool has_dp = (ALIAS & 1) || a.dp != nullptr;
    const bool in_place = OUT == 0 && a.out.gt_out != nullptr && a.out.gt_out == a.b.gt;
    // (the last wave of a row may be partly beyond S: its live lanes share the queue among themselves)
    const uint64_t exm = __ballot(live);
#pragma unroll
    for (int c = 0; c < CV; ++c) lm[c] = __ballot(livec[c]);
    const uint32_t n_lanes = (uint32_t)__popcll(exm);
    const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(exm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)exm, 0u));
    auto drain = [&]() {
        if (!DELTA) return;
        wave_lds_fence();
        for (uint32_t i = my_rank; i < qtail; i += n_lanes) {
            const uint2 r = queue[i];
            v4_drain_one(r.x, r.y, dtab, a.b, linfo, nal, dstride);
        }
        wave_lds_fence();
        qtail = 0;
    };
    struct In {
        u32x4 g, dv;
        u32x4 pv[NF];
    };
    // (plane sharing -- ALIAS bit 0: the depth vector is filter 0's plane; bit 1: filter 1 reads filter 0's plane -- is
    // resolved where a value is USED: a copy made at the load would wait for the load)
    auto fetch = [&](int l, In& in, int cv) {
        const int64_t c4 = ((int64_t)l * S + s0c[cv]) >> 2;
        in.g = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.b.gt) + c4);
#pragma unroll
        for (int k = 0; k < NF; ++k)
            if (!(k > 0 && ((ALIAS >> k) & 1)))
                in.pv[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.f[k].plane) + c4);
        if (!(ALIAS & 1) && a.dp) in.dv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.dp) + c4);
    };
    auto process = [&](int l, int l_begin, const In& in, int cv) {
        const int64_t c4 = ((int64_t)l * S + s0c[cv]) >> 2;
        Acc& ac = acc[cv];
        const u32x4& g = in.g;
        const u32x4& dv = (ALIAS & 1) ? in.pv[0] : in.dv;
#define TRK_PV(k) in.pv[((k) > 0 && ((ALIAS >> (k)) & 1)) ? (k) - 1 : (k)]
        u32x4 wout, mout;
        uint64_t bad = 0, touched = 0;
        const uint32_t li = (uint32_t)(l - l_begin);
#pragma unroll
        for (int j = 0; j < CF_V; ++j) {
            const uint32_t w = g[j];
            // called: neither half is the missing marker (one ballot per compare: the ballot of a combined
            // condition goes through a 0/1 register)
            uint64_t calledm = __ballot((w & 0xffffu) != 0xffffu) & __ballot(w < 0xffff0000u);
            if (cv > 0) calledm &= lm[cv];     // (a chunk beyond the row's end: its lanes are no calls, filtered by nothing)
            uint32_t m = __builtin_amdgcn_inverse_ballot_w64(calledm) ? 0u : nocallv;
            uint64_t anyhit = 0;
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const V4Filter& f = a.f[k];
                uint64_t c;
                if (RATIO && f.ratio) {
                    // RN(x / d) > t without the division (filters.py:415-484 divides two int32 columns in float64
                    // for every call; the decision is almost never close).  With the sign of d moved onto x the
                    // sign of x - t |d| -- one fused multiply-add, exact in sign -- says on which side of t the
                    // exact quotient lies: at or below t the rounded quotient is not above it either; above the
                    // NEXT double after t it is.  Only a quotient in between (an exact tie of the threshold such as
                    // 3/20 against 0.15, which rounds DOWN to t) and d == 0 (inf / nan) take the division, in a
                    // wave-uniform branch.
                    const int32_t xi = (int32_t)TRK_PV(k)[j], di = (int32_t)dv[j];
                    const double ad = __builtin_fabs((double)di);
                    const double xs = __longlong_as_double(__double_as_longlong((double)xi) ^
                                                           ((long long)((uint32_t)di & 0x80000000u) << 32));
                    const uint64_t yes = __ballot(__builtin_fma(-f.dthr_up, ad, xs) > 0.0);
                    const uint64_t no = __ballot(__builtin_fma(-f.dthr, ad, xs) <= 0.0);
                    const uint64_t unsure = (~(yes | no) | __ballot(di == 0)) & exm;
                    c = yes & ~unsure;
                    if (__builtin_expect(unsure != 0, 0)) c |= unsure & __ballot(((double)xi / (double)di) > f.dthr);
                } else if (NFLT < 0 ? f.is_float != 0 : k >= NF - NFLT)
                    c = __ballot(__uint_as_float(TRK_PV(k)[j] ^ f.flip) < __uint_as_float((uint32_t)f.thr));
                else
                    c = __ballot((int32_t)TRK_PV(k)[j] < f.thr) ^ inv[k];
                const uint64_t h = c & (calledm | nn[k]);
                m |= __builtin_amdgcn_inverse_ballot_w64(h) ? bitv[k] : 0u;
                add_mask(ac.fc[k][j], h & calledm);   // dumpSTR.py:661
                anyhit |= h;
            }
            const uint64_t passm = calledm & ~anyhit;   // mask word == 0: dumpSTR.py:686
            const uint64_t filtm = calledm & anyhit;    // called and not passing: dumpSTR.py:715-727
            add_mask(ac.numcalls[j], passm);
            if (OUT != 2) wout[j] = __builtin_amdgcn_inverse_ballot_w64(filtm) ? 0xffffffffu : w;
            mout[j] = m;
            touched |= filtm;
            if (has_dp) {
                // a passing call's depth is added when it is not negative; a negative one is the missing marker
                // (counted) or an error -- both rare, decided off the stream (dumpSTR.py:696-713)
                const int32_t d = (int32_t)dv[j];
                const uint64_t pneg = passm & __ballot(d < 0);
                ac.totaldp[j] += __builtin_amdgcn_inverse_ballot_w64(passm & ~pneg) ? d : 0;
                if (__builtin_expect(pneg != 0, 0)) {   // (wave-uniform test; rare: out of line)
                    const uint64_t missm = __ballot(d == INT32_MIN);
                    add_mask(ac.dpmiss[j], pneg & missm);
                    bad |= pneg & ~missm;
                }
            }
            if (DELTA && filtm) {   // (wave-uniform test) queue the filtered calls of slot j
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(filtm >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)filtm, 0u));
                if (__builtin_amdgcn_inverse_ballot_w64(filtm)) queue[qtail + rank] = make_uint2(w, li);
                qtail += (uint32_t)__popcll(filtm);
            }
        }
        if (__builtin_expect(has_dp && bad, 0)) {   // a negative depth on a call that passes (cold): dumpSTR.py:698-706
#pragma unroll
            for (int j = CF_V - 1; j >= 0; --j) {
                const int32_t d = (int32_t)dv[j];
                if ((mout[j] == 0u) & (d < 0) & (d != INT32_MIN)) {
                    if (atomicCAS(&a.out.error[0], 0, 1) == 0) {
                        a.out.error[1] = l;
                        a.out.error[2] = (int32_t)(s0c[cv] + j);
                    }
                }
            }
        }
        if (OUT == 2) {   // one byte per call, bit 7 = no-call
            const uint32_t m8 = mout[0] | (mout[1] << 8) | (mout[2] << 16) | (mout[3] << 24);
            if (cv == 0 || livec[cv]) __builtin_nontemporal_store(m8, reinterpret_cast<uint32_t*>(a.out.filter_mask8) + c4);
        } else {
            if (in_place) {   // only the chunks with a filtered call change
                if (__builtin_amdgcn_inverse_ballot_w64(touched))
                    __builtin_nontemporal_store(wout, reinterpret_cast<u32x4*>(a.out.gt_out) + c4);
            } else if (a.out.gt_out) {
                __builtin_nontemporal_store(wout, reinterpret_cast<u32x4*>(a.out.gt_out) + c4);
            }
            if (a.out.filter_mask) __builtin_nontemporal_store(mout, reinterpret_cast<u32x4*>(a.out.filter_mask) + c4);
            if (a.out.filter_mask8) {   // one byte per call: bit 7 = no-call
                uint32_t m8 = 0;
#pragma unroll
                for (int j = 0; j < CF_V; ++j) m8 |= ((mout[j] & 0x7fu) | ((mout[j] >> 24) & 0x80u)) << (8 * j);
                __builtin_nontemporal_store(m8, reinterpret_cast<uint32_t*>(a.out.filter_mask8) + c4);
            }
        }
        if (DELTA && __builtin_expect(qtail > (uint32_t)(V4_QCAP - CF_V * WAVE), 0)) drain();
#undef TRK_PV
    };
    const int n_blocks = (L + a.loci_per_block - 1) / a.loci_per_block;
    const int by_end = min(n_blocks, (biy + 1) * a.geom.walk);
    const int l_last = min(L, by_end * a.loci_per_block);   // end of this workgroup's range
    // OUT == 2: a ring of V4_PD input sets; set k holds locus (range start + k) mod V4_PD.  The loop is unrolled V4_PD
    // times so that every set is a fixed group of registers; a set is refilled -- with the locus V4_PD further on,
    // across the blocks of the range -- as soon as its locus has been worked on, so V4_PD - 1 loci are in flight while
    // one is worked on.  The host makes the block length a multiple of V4_PD (only the last block of the batch may be
    // shorter: its tail runs without refills).  Refills are UNCONDITIONAL (beyond the range's end the last locus is
    // fetched again): a branch around a fetch would turn the wait for a set's loads into a wait for everything (the
    // counter of outstanding loads is merged pessimistically where two paths meet).
    In ring[V4_PD][CV] = {};
    const int r_begin = biy * a.geom.walk * a.loci_per_block;
    if (OUT == 2 && live && r_begin < l_last) {
#pragma unroll
        for (int k = 0; k < V4_PD; ++k)
#pragma unroll
            for (int c = 0; c < CV; ++c) fetch(min(r_begin + k, l_last - 1), ring[k][c], c);
    }
    for (int by = biy * a.geom.walk; by < by_end; ++by) {
        const int l_begin = by * a.loci_per_block;
        const int l_end = min(L, l_begin + a.loci_per_block);
        const int nl = l_end - l_begin;
        if (DELTA) {
            for (int i = tid; i < nl * dstride; i += CF_THREADS) dtab[i] = 0;
            cf_build_lut<false>(a.b, l_begin, nl, nal, tid, nullptr, linfo);
        }
        if (live) {
            if (OUT == 2) {
                int l = l_begin;
                for (; l + V4_PD <= l_end; l += V4_PD) {
#pragma unroll
                    for (int k = 0; k < V4_PD; ++k)
#pragma unroll
                        for (int c = 0; c < CV; ++c) {
                            process(l + k, l_begin, ring[k][c], c);
                            fetch(min(l + k + V4_PD, l_last - 1), ring[k][c], c);
                        }
                }
#pragma unroll
                for (int k = 0; k < V4_PD - 1; ++k)     // (the batch's last block: fewer than V4_PD loci left)
                    if (l + k < l_end) {
#pragma unroll
                        for (int c = 0; c < CV; ++c) process(l + k, l_begin, ring[k][c], c);
                    }
            } else {
                for (int l = l_begin; l < l_end; ++l) {
#pragma unroll
                    for (int c = 0; c < CV; ++c) {
                        fetch(l, ring[0][c], c);
                        process(l, l_begin, ring[0][c], c);
                    }
                }
            }
            drain();
        }
        if (DELTA) {  // one global atomic per non-zero entry of the block's delta table
            __syncthreads();
            const uint32_t rcp = (uint32_t)((0x100000000ull + (uint32_t)dstride - 1u) / (uint32_t)dstride);  // i / dstride
            for (int i = tid; i < nl * dstride; i += CF_THREADS) {
                const uint32_t v = dtab[i];
                if (!v) continue;
                const int li = (int)__umulhi((uint32_t)i, rcp);
                const int r = i - li * dstride;
                const int l = l_begin + li;
                if (r < nal) {
                    atomicSub(&a.out.delta_allele_count[linfo[CF_LINFO * li + 2] + r], (int)v);
                } else if (r == nal + V2_W0) {
                    int32_t* li_ = a.out.delta_locus_int + (int64_t)l * TRK_LI_COLS;
                    atomicSub(&li_[TRK_LI_N_CALLED], (int)(v & 0xffffu));
                    if (v >> 16) atomicSub(&li_[TRK_LI_N_LOWPLOIDY], (int)(v >> 16));
                } else if (r == nal + V2_W1) {
                    int32_t* li_ = a.out.delta_locus_int + (int64_t)l * TRK_LI_COLS;
                    if (v & 0xffffu) atomicSub(&li_[TRK_LI_N_HOM_LEN], (int)(v & 0xffffu));
                    if (v >> 16) atomicSub(&li_[TRK_LI_N_HOM_STR], (int)(v >> 16));
                }
            }
            __syncthreads();   // the table is re-initialised for the next block
        }
    }
#pragma unroll
    for (int cv = 0; cv < CV; ++cv) {
        if (!livec[cv]) continue;
        const Acc& ac = acc[cv];
        const int64_t sc = s0c[cv];
        if (a.part16) {
            // this workgroup's counters of the chunk's 4 samples: one 8-byte store per counter row, 32 bytes of depth sums
            typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u16x4* p16 = reinterpret_cast<u16x4*>(a.part16 + ((size_t)biy * (2 + NF)) * S + sc);
            const size_t rs = (size_t)S / 4;   // row stride in u16x4
            p16[0] = (u16x4){(unsigned short)ac.numcalls[0], (unsigned short)ac.numcalls[1], (unsigned short)ac.numcalls[2],
                             (unsigned short)ac.numcalls[3]};
            p16[rs] = (u16x4){(unsigned short)ac.dpmiss[0], (unsigned short)ac.dpmiss[1], (unsigned short)ac.dpmiss[2],
                              (unsigned short)ac.dpmiss[3]};
#pragma unroll
            for (int k = 0; k < NF; ++k)
                p16[(2 + k) * rs] = (u16x4){(unsigned short)ac.fc[k][0], (unsigned short)ac.fc[k][1], (unsigned short)ac.fc[k][2],
                                            (unsigned short)ac.fc[k][3]};
            u64x2* p64 = reinterpret_cast<u64x2*>(a.part64 + (size_t)biy * S + sc);
            p64[0] = (u64x2){(unsigned long long)ac.totaldp[0], (unsigned long long)ac.totaldp[1]};
            p64[1] = (u64x2){(unsigned long long)ac.totaldp[2], (unsigned long long)ac.totaldp[3]};
        } else {
#pragma unroll
            for (int j = 0; j < CF_V; ++j) {
                const int64_t s = sc + j;
                if (ac.numcalls[j])
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_counters + s),
                              (unsigned long long)ac.numcalls[j]);
                if (ac.totaldp[j])
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_totaldp + s),
                              (unsigned long long)ac.totaldp[j]);
                if (ac.dpmiss[j])
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_dp_missing + s),
                              (unsigned long long)ac.dpmiss[j]);
#pragma unroll
                for (int k = 0; k < NF; ++k)
                    if (ac.fc[k][j])
                        atomicAdd(reinterpret_cast<unsigned long long*>(a.out.sample_counters +
                                                                        (int64_t)(1 + a.f[k].bit) * S + s),
                                  (unsigned long long)ac.fc[k][j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_call_filter_gs : dumpSTR's CLOSED list of GangSTR call filters (dumpSTR.py:819-836, filters.py:327-409 and
// 573-757) with static operand slots -- no interpreter loop, no indexed register moves, no scratch.  The nine
// predicates in a fixed role order, each switched on or off by a wave-uniform flag:
//   0 min DP   DP < t                 (every call)          5 QEXP total  QEXP[1] + QEXP[2] < t (float32 sum; called)
//   1 max DP   DP > t                 (every call)          6 span only   RC[1] == DP                     (called)
//   2 min Q    Q < t (float32)        (every call)          7 span+bound  RC[1] + RC[3] == DP (int64 sum; called)
//   3 QEXP het QEXP[1] < t            (called)              8 bad CI      REPCN[j] outside REPCI[2j .. 2j+1], j = 0, 1
//   4 QEXP hom QEXP[2] < t            (called)
// Same tiling, counters, delta table and outputs as k_call_filter_v4 (column-owner: thread = 4 samples x the loci of
// its workgroup).  PLANAR: every column its own [L, S] array (13 16-byte loads per locus, unused columns never
// read); otherwise the planes as cyvcf2 / Engine.upload_plane hands them, [L, S, k]: the k vectors of a thread's
// four calls are loaded whole and the columns picked out of the registers with compile-time indices (16 loads).
// ---------------------------------------------------------------------------
constexpr int GS_NF = 9;
struct GsFilter {
    int32_t on, bit, ithr;
    float fthr;
};
struct GsArgs {
    trk_batch b;
    GsFilter f[GS_NF];
    // PLANAR: dp, q, qexp[1], qexp[2], rc[1], rc[3], repcn[0], repcn[1], repci[0..3]; interleaved: dp, q, then the
    // bases of QEXP [.,3], RC [.,4], REPCN [.,2], REPCI [.,4] in slots 2..5
    const void* p[12];
    int use_qexp, use_rc, use_ci, has_dp;
    int loci_per_block, delta_nal;
    uint16_t* part16;            // [n_ranges][2 + GS_NF][S]
    unsigned long long* part64;  // [n_ranges][S]
    CfGeom geom;                 // workgroup -> (column tile, range of locus blocks), as in k_call_filter_v4
    trk_call_out out;
};

// MASK >= 0: the enabled roles are a compile-time set and filter `role` owns mask bit / counter row
// popcount(MASK & ((1 << role) - 1)) -- the order dumpSTR builds its GangSTR list in (dumpSTR.py:819-836); no flag
// or bit registers, no branches.  MASK < 0: flags and bits from the arguments.
template <bool PLANAR, bool DELTA, int MASK>
__global__ __launch_bounds__(CF_THREADS) void k_call_filter_gs(const GsArgs a) {
    extern __shared__ uint32_t v2lds[];
    const int tid = threadIdx.x;
    const int S = a.b.n_samples, L = a.b.n_loci;
    int bix, biy;
    if (!cf_place(a.geom, bix, biy)) return;
    const int64_t s0 = ((int64_t)bix * CF_THREADS + tid) * CF_V;
    const int nal = a.delta_nal;
    const int dstride = nal + V2_EXTRA;
    uint32_t* dtab = v2lds;
    int32_t* linfo = reinterpret_cast<int32_t*>(dtab + (size_t)a.loci_per_block * dstride);
    // this wave's queue of filtered calls {genotype word, locus of the block}, behind the tables (k_call_filter_v4)
    uint2* queue = reinterpret_cast<uint2*>(v2lds + ((((size_t)a.loci_per_block * (dstride + CF_LINFO)) + 1) & ~(size_t)1)) +
                   (size_t)(tid >> 6) * V4_QCAP;
    uint32_t qtail = 0;
    const uint64_t exm = __ballot(s0 < S);
    const uint32_t n_lanes = (uint32_t)__popcll(exm);
    const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(exm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)exm, 0u));
    auto drain = [&]() {
        if (!DELTA) return;
        wave_lds_fence();
        for (uint32_t i = my_rank; i < qtail; i += n_lanes) {
            const uint2 r = queue[i];
            v4_drain_one(r.x, r.y, dtab, a.b, linfo, nal, dstride);
        }
        wave_lds_fence();
        qtail = 0;
    };
    // per-sample counters as 16-bit pairs (samples 2 i and 2 i + 1 share a register: a workgroup's loci stay below
    // 65536): half the registers of one counter per sample -- 22 instead of 44 VGPRs, the difference between three
    // and four waves per SIMD.  The low half takes the mask as the carry of an add, the high half through the SDWA
    // form of the same instruction (carry in vcc).
    uint32_t numcalls[2] = {0, 0}, dpmiss[2] = {0, 0};
    uint32_t fc[GS_NF][2];
    int64_t totaldp[CF_V] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < GS_NF; ++k) fc[k][0] = fc[k][1] = 0;
    uint32_t vzero = 0;
    asm volatile("" : "+v"(vzero));     // (a register that holds 0: SDWA takes no inline constants)
    auto add16 = [&](uint32_t (&c)[2], int j, uint64_t mask) {
        if (j & 1)
            asm("s_mov_b64 vcc, %2\n\tv_addc_co_u32_sdwa %0, vcc, %1, %0, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE "
                "src0_sel:DWORD src1_sel:WORD_1"
                : "+v"(c[j >> 1]) : "v"(vzero), "s"(mask) : "vcc");
        else
            add_mask(c[j >> 1], mask);
    };
    const int n_blocks = (L + a.loci_per_block - 1) / a.loci_per_block;
    const int by_end = min(n_blocks, (biy + 1) * a.geom.walk);
    for (int by = biy * a.geom.walk; by < by_end; ++by) {
    const int l_begin = by * a.loci_per_block;
    const int l_end = min(L, l_begin + a.loci_per_block);
    const int nl = l_end - l_begin;
    if (DELTA) {
        for (int i = tid; i < nl * dstride; i += CF_THREADS) dtab[i] = 0;
        cf_build_lut<false>(a.b, l_begin, nl, nal, tid, nullptr, linfo);
    }
    if (s0 < S) {
        for (int l = l_begin; l < l_end; ++l) {
            const int64_t c4 = ((int64_t)l * S + s0) >> 2;
            // (interleaved planes: a lane's k vectors are k x 16 consecutive bytes, so every load instruction of the
            // group touches every cache line of the wave's 64 k x 16 bytes -- plain loads, which the L1 may keep
            // between them, instead of the streaming hint: r03_ab_gangstr_interleaved_loads.txt)
            auto ld = [&](int slot, int64_t idx) {
                const u32x4* q = reinterpret_cast<const u32x4*>(a.p[slot]) + idx;
                return (PLANAR || slot < 2) ? __builtin_nontemporal_load(q) : *q;
            };
            const u32x4 g = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.b.gt) + c4);
            u32x4 dv = {0, 0, 0, 0}, qv = {0, 0, 0, 0};
            const bool use_q = MASK >= 0 ? ((MASK >> 2) & 1) != 0 : a.f[2].on != 0;
            const bool use_qexp = MASK >= 0 ? (MASK & 0x38) != 0 : a.use_qexp != 0;
            const bool use_rc = MASK >= 0 ? (MASK & 0xc0) != 0 : a.use_rc != 0;
            const bool use_ci = MASK >= 0 ? (MASK & 0x100) != 0 : a.use_ci != 0;
            const bool has_dp = MASK >= 0 ? true : a.has_dp != 0;
            if (has_dp) dv = ld(0, c4);
            if (use_q) qv = ld(1, c4);
            // operands by call j: e1 / e2 = QEXP[1] / QEXP[2], r1 / r3 = RC[1] / RC[3], cn / lo / hi by haplotype
            u32x4 e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
            u32x4 cn0 = {0, 0, 0, 0}, cn1 = {0, 0, 0, 0}, lo0 = {0, 0, 0, 0}, hi0 = {0, 0, 0, 0}, lo1 = {0, 0, 0, 0},
                  hi1 = {0, 0, 0, 0};
            if (PLANAR) {
                if (use_qexp) { e1 = ld(2, c4); e2 = ld(3, c4); }
                if (use_rc) { r1 = ld(4, c4); r3 = ld(5, c4); }
                if (use_ci) { cn0 = ld(6, c4); cn1 = ld(7, c4); lo0 = ld(8, c4); hi0 = ld(9, c4); lo1 = ld(10, c4); hi1 = ld(11, c4); }
            } else {
                if (use_qexp) {           // 4 calls x 3 floats = 3 vectors: element 3 j + c of the 12
                    const u32x4 x0 = ld(2, c4 * 3), x1 = ld(2, c4 * 3 + 1), x2 = ld(2, c4 * 3 + 2);
                    e1 = (u32x4){x0[1], x1[0], x1[3], x2[2]};
                    e2 = (u32x4){x0[2], x1[1], x2[0], x2[3]};
                }
                if (use_rc) {             // call j = vector j: {enclosing, spanning, flanking, bound}
                    const u32x4 x0 = ld(3, c4 * 4), x1 = ld(3, c4 * 4 + 1), x2 = ld(3, c4 * 4 + 2), x3 = ld(3, c4 * 4 + 3);
                    r1 = (u32x4){x0[1], x1[1], x2[1], x3[1]};
                    r3 = (u32x4){x0[3], x1[3], x2[3], x3[3]};
                }
                if (use_ci) {
                    const u32x4 n0 = ld(4, c4 * 2), n1 = ld(4, c4 * 2 + 1);
                    cn0 = (u32x4){n0[0], n0[2], n1[0], n1[2]};
                    cn1 = (u32x4){n0[1], n0[3], n1[1], n1[3]};
                    const u32x4 x0 = ld(5, c4 * 4), x1 = ld(5, c4 * 4 + 1), x2 = ld(5, c4 * 4 + 2), x3 = ld(5, c4 * 4 + 3);
                    lo0 = (u32x4){x0[0], x1[0], x2[0], x3[0]};
                    hi0 = (u32x4){x0[1], x1[1], x2[1], x3[1]};
                    lo1 = (u32x4){x0[2], x1[2], x2[2], x3[2]};
                    hi1 = (u32x4){x0[3], x1[3], x2[3], x3[3]};
                }
            }
            uint64_t calledm[CF_V], anyhit[CF_V];
            u32x4 mout;
#pragma unroll
            for (int j = 0; j < CF_V; ++j) {
                const uint32_t w = g[j];
                calledm[j] = __ballot((w & 0xffffu) != 0xffffu) & __ballot(w < 0xffff0000u);
                mout[j] = __builtin_amdgcn_inverse_ballot_w64(calledm[j]) ? 0u : TRK_MASK_NOCALL;
                anyhit[j] = 0;
            }
            // one predicate: the lane mask of the calls it fires on (`hit`), its bit into the mask word, its per-sample
            // counter bumped where the call is made (dumpSTR.py:661), the union for the pass / filtered decision
#define TRK_GS_FILTER(K, CALLED_ONLY, EXPR)                                                   \
            if (MASK >= 0 ? ((MASK >> K) & 1) != 0 : a.f[K].on != 0) {                        \
                const uint32_t bitv = MASK >= 0 ? 1u << __builtin_popcount(MASK & ((1 << K) - 1)) : 1u << a.f[K].bit; \
                _Pragma("unroll") for (int j = 0; j < CF_V; ++j) {                            \
                    uint64_t hit = __ballot(EXPR);                                            \
                    if (CALLED_ONLY) hit &= calledm[j];                                       \
                    mout[j] |= __builtin_amdgcn_inverse_ballot_w64(hit) ? bitv : 0u;          \
                    add16(fc[K], j, CALLED_ONLY ? hit : (hit & calledm[j]));                  \
                    anyhit[j] |= hit;                                                         \
                }                                                                             \
            }
            TRK_GS_FILTER(0, false, (int32_t)dv[j] < a.f[0].ithr)
            TRK_GS_FILTER(1, false, (int32_t)dv[j] > a.f[1].ithr)
            TRK_GS_FILTER(2, false, __uint_as_float(qv[j]) < a.f[2].fthr)
            TRK_GS_FILTER(3, true, __uint_as_float(e1[j]) < a.f[3].fthr)
            TRK_GS_FILTER(4, true, __uint_as_float(e2[j]) < a.f[4].fthr)
            TRK_GS_FILTER(5, true, (__uint_as_float(e1[j]) + __uint_as_float(e2[j])) < a.f[5].fthr)
            TRK_GS_FILTER(6, true, (int32_t)r1[j] == (int32_t)dv[j])
            TRK_GS_FILTER(7, true, ((int64_t)(int32_t)r1[j] + (int64_t)(int32_t)r3[j]) == (int64_t)(int32_t)dv[j])
            TRK_GS_FILTER(8, true, ((int32_t)cn0[j] < (int32_t)lo0[j]) | ((int32_t)hi0[j] < (int32_t)cn0[j]) |
                                       ((int32_t)cn1[j] < (int32_t)lo1[j]) | ((int32_t)hi1[j] < (int32_t)cn1[j]))
#undef TRK_GS_FILTER
            uint64_t passm[CF_V], filtm[CF_V];
            u32x4 wout;
#pragma unroll
            for (int j = 0; j < CF_V; ++j) {
                passm[j] = calledm[j] & ~anyhit[j];   // dumpSTR.py:686
                filtm[j] = calledm[j] & anyhit[j];    // dumpSTR.py:715-727
                add16(numcalls, j, passm[j]);
                wout[j] = __builtin_amdgcn_inverse_ballot_w64(filtm[j]) ? 0xffffffffu : g[j];
            }
            if (has_dp) {
                uint64_t bad = 0;
#pragma unroll
                for (int j = 0; j < CF_V; ++j) {
                    const int32_t d = (int32_t)dv[j];
                    add16(dpmiss, j, passm[j] & __ballot(d == INT32_MIN));
                    const int32_t dpos = d > 0 ? d : 0;
                    totaldp[j] += __builtin_amdgcn_inverse_ballot_w64(passm[j]) ? dpos : 0;
                    bad |= passm[j] & __ballot((uint32_t)d > 0x80000000u);
                }
                if (__builtin_expect(bad != 0, 0)) {   // a negative depth on a call that passes (cold): dumpSTR.py:698-706
#pragma unroll
                    for (int j = CF_V - 1; j >= 0; --j) {
                        const int32_t d = (int32_t)dv[j];
                        if ((mout[j] == 0u) & (d < 0) & (d != INT32_MIN)) {
                            if (atomicCAS(&a.out.error[0], 0, 1) == 0) {
                                a.out.error[1] = l;
                                a.out.error[2] = (int32_t)(s0 + j);
                            }
                        }
                    }
                }
            }
            if (DELTA) {    // the filtered calls are queued (as in k_call_filter_v4) and drained in bulk
                const uint32_t li = (uint32_t)(l - l_begin);
#pragma unroll
                for (int j = 0; j < CF_V; ++j) {
                    if (!filtm[j]) continue;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(filtm[j] >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)filtm[j], 0u));
                    if (__builtin_amdgcn_inverse_ballot_w64(filtm[j])) queue[qtail + rank] = make_uint2(g[j], li);
                    qtail += (uint32_t)__popcll(filtm[j]);
                }
            }
            if (a.out.gt_out == a.b.gt) {   // in place (k_call_filter_v4's header): only the chunks with a filtered call
                if (a.out.gt_out && __builtin_amdgcn_inverse_ballot_w64(filtm[0] | filtm[1] | filtm[2] | filtm[3]))
                    __builtin_nontemporal_store(wout, reinterpret_cast<u32x4*>(a.out.gt_out) + c4);
            } else if (a.out.gt_out) {
                __builtin_nontemporal_store(wout, reinterpret_cast<u32x4*>(a.out.gt_out) + c4);
            }
            if (a.out.filter_mask) __builtin_nontemporal_store(mout, reinterpret_cast<u32x4*>(a.out.filter_mask) + c4);
            if (a.out.filter_mask8) {
                uint32_t m8 = 0;
#pragma unroll
                for (int j = 0; j < CF_V; ++j) m8 |= ((mout[j] & 0x7fu) | ((mout[j] >> 24) & 0x80u)) << (8 * j);
                __builtin_nontemporal_store(m8, reinterpret_cast<uint32_t*>(a.out.filter_mask8) + c4);
            }
            if (DELTA && __builtin_expect(qtail > (uint32_t)(V4_QCAP - CF_V * WAVE), 0)) drain();
        }
        drain();
    }
    if (DELTA) {
        __syncthreads();
        const uint32_t rcp = (uint32_t)((0x100000000ull + (uint32_t)dstride - 1u) / (uint32_t)dstride);
        for (int i = tid; i < nl * dstride; i += CF_THREADS) {
            const uint32_t v = dtab[i];
            if (!v) continue;
            const int li = (int)__umulhi((uint32_t)i, rcp);
            const int r = i - li * dstride;
            const int l = l_begin + li;
            if (r < nal) {
                atomicSub(&a.out.delta_allele_count[linfo[CF_LINFO * li + 2] + r], (int)v);
            } else if (r == nal + V2_W0) {
                int32_t* li_ = a.out.delta_locus_int + (int64_t)l * TRK_LI_COLS;
                atomicSub(&li_[TRK_LI_N_CALLED], (int)(v & 0xffffu));
                if (v >> 16) atomicSub(&li_[TRK_LI_N_LOWPLOIDY], (int)(v >> 16));
            } else if (r == nal + V2_W1) {
                int32_t* li_ = a.out.delta_locus_int + (int64_t)l * TRK_LI_COLS;
                if (v & 0xffffu) atomicSub(&li_[TRK_LI_N_HOM_LEN], (int)(v & 0xffffu));
                if (v >> 16) atomicSub(&li_[TRK_LI_N_HOM_STR], (int)(v >> 16));
            }
        }
        __syncthreads();
    }
    }  // locus blocks of this workgroup
    if (s0 < S) {
        typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u16x4* p16 = reinterpret_cast<u16x4*>(a.part16 + ((size_t)biy * (2 + GS_NF)) * S + s0);
        const size_t rs = (size_t)S / 4;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        // (the pairs are already in the partial rows' layout: four 16-bit counters = two registers = one 8-byte store)
        reinterpret_cast<u32x2*>(p16)[0] = (u32x2){numcalls[0], numcalls[1]};
        reinterpret_cast<u32x2*>(p16 + rs)[0] = (u32x2){dpmiss[0], dpmiss[1]};
#pragma unroll
        for (int k = 0; k < GS_NF; ++k) reinterpret_cast<u32x2*>(p16 + (2 + k) * rs)[0] = (u32x2){fc[k][0], fc[k][1]};
        u64x2* p64 = reinterpret_cast<u64x2*>(a.part64 + (size_t)biy * S + s0);
        p64[0] = (u64x2){(unsigned long long)totaldp[0], (unsigned long long)totaldp[1]};
        p64[1] = (u64x2){(unsigned long long)totaldp[2], (unsigned long long)totaldp[3]};
    }
}

// ---------------------------------------------------------------------------
// k_cf_reduce : sums the per-workgroup partial sample counters of k_call_filter_v4 / _gs into the int64 outputs.
// Thread (quad of 4 samples, slice): walks the locus blocks by = slice, slice + NSL, ...; the NSL slices of a quad are
// added up through LDS and one thread per quad does the (non-atomic) += on the outputs.  12.5k loci x 10k samples:
// 1280 workgroups x 60k counters = 7.7 M device-scope atomics (58 us, all at the end of the single round of
// workgroups) become 18 MB of plain stores + this kernel; at 100k loci 75 M atomics (0.11 ms) become 125 MB.
// ---------------------------------------------------------------------------
constexpr int CFR_NSL = 16, CFR_QPB = 16;   // slices per quad, quads per workgroup (256 threads)
constexpr int CFR_MAX_FILTERS = 9;            // (the closed GangSTR list of k_call_filter_gs)
constexpr int CFR_ROWS = 2 + CFR_MAX_FILTERS, CFR_UNR = 4;
struct CfrBits { int32_t bit[CFR_MAX_FILTERS]; };   // counter row of filter k = 1 + bit[k]
__global__ __launch_bounds__(CFR_NSL* CFR_QPB) void k_cf_reduce(const uint16_t* __restrict__ part16,
                                                               const unsigned long long* __restrict__ part64,
                                                               int gy, int S, int nrow16, CfrBits fb, trk_call_out out) {
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    __shared__ uint32_t red32[CFR_NSL][CFR_ROWS][CFR_QPB * 4];
    __shared__ unsigned long long red64[CFR_NSL][CFR_QPB * 4];
    const int ql = threadIdx.x % CFR_QPB, slice = threadIdx.x / CFR_QPB;
    const int quad = blockIdx.x * CFR_QPB + ql;
    const bool live = quad * 4 < S;
    const size_t rs = (size_t)S / 4;
    uint32_t acc[CFR_ROWS][4];
    unsigned long long accd[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < CFR_ROWS; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0;
    if (live) {
        const u16x4* p16 = reinterpret_cast<const u16x4*>(part16) + quad;
        const u64x2* p64 = reinterpret_cast<const u64x2*>(part64) + (size_t)quad * 2;
        // every row of CFR_UNR locus blocks requested before the first is added: a handful of round trips in all
        for (int by0 = slice; by0 < gy; by0 += CFR_NSL * CFR_UNR) {
            u16x4 v[CFR_UNR][CFR_ROWS];
            u64x2 da[CFR_UNR], db[CFR_UNR];
#pragma unroll
            for (int u = 0; u < CFR_UNR; ++u) {
                const int by = by0 + u * CFR_NSL;
                const bool ok = by < gy;
                const size_t byc = ok ? by : 0;
#pragma unroll
                for (int r = 0; r < CFR_ROWS; ++r)
                    v[u][r] = (ok && r < nrow16) ? p16[(byc * nrow16 + r) * rs] : (u16x4){0, 0, 0, 0};
                da[u] = ok ? p64[byc * rs * 2] : (u64x2){0, 0};
                db[u] = ok ? p64[byc * rs * 2 + 1] : (u64x2){0, 0};
            }
#pragma unroll
            for (int u = 0; u < CFR_UNR; ++u) {
#pragma unroll
                for (int r = 0; r < CFR_ROWS; ++r) {
                    acc[r][0] += v[u][r].x; acc[r][1] += v[u][r].y; acc[r][2] += v[u][r].z; acc[r][3] += v[u][r].w;
                }
                accd[0] += da[u].x; accd[1] += da[u].y; accd[2] += db[u].x; accd[3] += db[u].y;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CFR_ROWS; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) red32[slice][r][ql * 4 + j] = acc[r][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) red64[slice][ql * 4 + j] = accd[j];
    __syncthreads();
    const int* bits = fb.bit;
    // outputs of this workgroup: (nrow16 + 1) rows x 64 samples, one thread each (non-atomic +=: nobody else writes
    // these counters while the call-filter pass runs)
    for (int o = threadIdx.x; o < (nrow16 + 1) * CFR_QPB * 4; o += CFR_NSL * CFR_QPB) {
        const int r = o / (CFR_QPB * 4), c = o % (CFR_QPB * 4);
        const int64_t smp = (int64_t)blockIdx.x * CFR_QPB * 4 + c;
        if (smp >= S) continue;
        unsigned long long t = 0;
        if (r < nrow16) {
#pragma unroll
            for (int k = 0; k < CFR_NSL; ++k) t += red32[k][r][c];
        } else {
#pragma unroll
            for (int k = 0; k < CFR_NSL; ++k) t += red64[k][c];
        }
        int64_t* dst = r == 0       ? out.sample_counters + smp
                       : r == 1     ? out.sample_dp_missing + smp
                       : r < nrow16 ? out.sample_counters + (int64_t)(1 + bits[r - 2]) * S + smp
                                    : out.sample_totaldp + smp;
        if (t) *dst += (int64_t)t;
    }
}

// ---------------------------------------------------------------------------
// k_locus_filter : one thread per locus
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_locus_filter(int L, const int32_t* __restrict__ locus_int,
                                                     const double* __restrict__ locus_f64,
                                                     trk_locus_filter_spec spec, uint32_t* __restrict__ bits_out,
                                                     unsigned long long* __restrict__ counters) {
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int32_t* li = locus_int + (int64_t)l * TRK_LI_COLS;
    const double* lf = locus_f64 + (int64_t)l * TRK_LF_COLS;
    uint32_t bits = 0;
    // nan thresholds disable a filter; nan statistics never fire (x < nan == false)
    if (spec.min_callrate == spec.min_callrate && lf[TRK_LF_CALLRATE] < spec.min_callrate)
        bits |= 1u << TRK_LOCF_CALLRATE;  // filters.py:60
    if (spec.min_hwep == spec.min_hwep) {
        int st = li[spec.use_length ? TRK_LI_HWE_STATUS_LEN : TRK_LI_HWE_STATUS_STR];
        if (st == TRK_HWE_VALUE_ERROR || st == TRK_HWE_INDEX_ERROR) atomicAdd(&counters[TRK_LC_HWE_ERRORS], 1ull);
        double hw = lf[spec.use_length ? TRK_LF_HWEP_LEN : TRK_LF_HWEP_STR];
        if (hw < spec.min_hwep) bits |= 1u << TRK_LOCF_HWE;  // filters.py:102
    }
    double het = lf[spec.use_length ? TRK_LF_HET_LEN : TRK_LF_HET_STR];
    if (spec.min_het == spec.min_het && het < spec.min_het) bits |= 1u << TRK_LOCF_HETLOW;   // filters.py:142
    if (spec.max_het == spec.max_het && het > spec.max_het) bits |= 1u << TRK_LOCF_HETHIGH;  // filters.py:183
    if (spec.extern_bits) bits |= (spec.extern_bits[l] & ((1u << spec.n_extern) - 1u)) << TRK_LOCF_EXTERN0;
    for (int k = 0; k < 28; ++k)
        if ((bits >> k) & 1u) atomicAdd(&counters[TRK_LC_FILTER0 + k], 1ull);  // dumpSTR.py:949
    const int n_called = li[TRK_LI_N_CALLED];
    if (n_called == 0) {  // dumpSTR.py:957-965
        bits |= 1u << TRK_LOCF_NO_CALLS;
        atomicAdd(&counters[TRK_LC_NO_CALLS], 1ull);
    }
    if (bits == 0) {  // dumpSTR.py:967-971
        atomicAdd(&counters[TRK_LC_PASS], 1ull);
        atomicAdd(&counters[TRK_LC_TOTALCALLS], (unsigned long long)n_called);
    }
    bits_out[l] = bits;
}

// ---------------------------------------------------------------------------
// k_synth : synthetic diploid genotypes + HipSTR-shaped FORMAT planes
// (bit-for-bit twin: trtools_amd/synth.py::cells_numpy)
// ---------------------------------------------------------------------------
// k_class_combine : group totals from the per-class counts of the class passes (trk_batch.class_runs): a group's
// allele counts and its additive locus_int columns are the sums over the classes whose bits contain the group.
// ws: per used class [sumA allele counts | L x TRK_LI_COLS rows], back to back.  Thread = one output element of
// group 0's arrays, looping over the groups.
// ---------------------------------------------------------------------------
struct ClassRuns {
    int n;
    uint8_t bits[256];
};
__global__ __launch_bounds__(256) void k_class_combine(ClassRuns cr, const int32_t* __restrict__ ws, int64_t n_ac,
                                                      int64_t n_li, int G, int32_t* __restrict__ allele_count,
                                                      int32_t* __restrict__ locus_int, int64_t twin_ac, int64_t twin_li) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ac + n_li) return;
    const int64_t per = n_ac + n_li;
    int32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool additive = true;
    if (i >= n_ac) {
        const int col = (int)((i - n_ac) % TRK_LI_COLS);
        additive = col == TRK_LI_N_CALLED || col == TRK_LI_N_LOWPLOIDY || col == TRK_LI_N_HOM_LEN ||
                   col == TRK_LI_N_HOM_STR || col == TRK_LI_N_BAD || col == TRK_LI_N_SAMPLES;
    }
    if (additive)
        for (int r = 0; r < cr.n; ++r) {
            const int32_t v = ws[(int64_t)r * per + i];
            const uint32_t bits = cr.bits[r];
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g] += ((bits >> g) & 1u) ? v : 0;
        }
    for (int g = 0; g < G; ++g) {
        if (i < n_ac) {
            allele_count[(int64_t)g * n_ac + i] = acc[g];
            if (twin_ac) allele_count[twin_ac + (int64_t)g * n_ac + i] = acc[g];
        } else {
            locus_int[(int64_t)g * n_li + (i - n_ac)] = acc[g];
            if (twin_li) locus_int[twin_li + (int64_t)g * n_li + (i - n_ac)] = acc[g];
        }
    }
}

// dst[l, j, :] = src[l, col[j], :] (col[j] < 0: a no-call column).  Diploid, n_dst % 4 == 0: one workgroup per row at a
// time (the row's 4 n_src bytes are fetched by ONE XCD and then served from its L2 however scattered col[] is; with
// the rows' chunks spread over workgroups every XCD pulled most of every row: 3.4 ms instead of ~1.6 for 100k x
// 10k), a thread stores 16 bytes = four gathered calls.  Any other shape: element-wise.
__global__ __launch_bounds__(256) void k_permute_columns(const int16_t* __restrict__ src, int16_t* __restrict__ dst,
                                                        const int32_t* __restrict__ col, int64_t n_loci, int n_src,
                                                        int n_dst, int ploidy, int use_lds) {
    extern __shared__ uint32_t perm_lds[];
    if (ploidy == 2 && (n_dst & 3) == 0) {
        const int q4 = n_dst >> 2;
        for (int64_t l = blockIdx.x; l < n_loci; l += gridDim.x) {
            const uint32_t* srow = reinterpret_cast<const uint32_t*>(src) + l * n_src;
            u32x4* drow = reinterpret_cast<u32x4*>(dst) + l * q4;
            if (use_lds) {
                // the row into LDS with coalesced loads (16 bytes per lane where the row start is aligned), the gather
                // out of LDS: scattered 4-byte reads cost LDS bank conflicts, not a cache line per lane
                if ((n_src & 3) == 0) {
                    const u32x4* s4 = reinterpret_cast<const u32x4*>(srow);
                    for (int i = threadIdx.x; i < (n_src >> 2); i += 256)
                        reinterpret_cast<u32x4*>(perm_lds)[i] = __builtin_nontemporal_load(s4 + i);
                } else {
                    for (int i = threadIdx.x; i < n_src; i += 256) perm_lds[i] = srow[i];
                }
                __syncthreads();
                for (int j4 = threadIdx.x; j4 < q4; j4 += 256) {
                    const int4 c = reinterpret_cast<const int4*>(col)[j4];
                    u32x4 o;
                    o[0] = c.x >= 0 ? perm_lds[c.x] : 0xffffffffu;
                    o[1] = c.y >= 0 ? perm_lds[c.y] : 0xffffffffu;
                    o[2] = c.z >= 0 ? perm_lds[c.z] : 0xffffffffu;
                    o[3] = c.w >= 0 ? perm_lds[c.w] : 0xffffffffu;
                    __builtin_nontemporal_store(o, drow + j4);
                }
                __syncthreads();
                continue;
            }
            for (int j4 = threadIdx.x; j4 < q4; j4 += 256) {
                const int4 c = reinterpret_cast<const int4*>(col)[j4];
                u32x4 o;
                o[0] = c.x >= 0 ? srow[c.x] : 0xffffffffu;
                o[1] = c.y >= 0 ? srow[c.y] : 0xffffffffu;
                o[2] = c.z >= 0 ? srow[c.z] : 0xffffffffu;
                o[3] = c.w >= 0 ? srow[c.w] : 0xffffffffu;
                __builtin_nontemporal_store(o, drow + j4);
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_loci * n_dst; i += (int64_t)gridDim.x * 256) {
        const int64_t l = i / n_dst;
        const int c = col[(int)(i - l * n_dst)];
        for (int p = 0; p < ploidy; ++p) dst[i * ploidy + p] = c >= 0 ? src[(l * n_src + c) * ploidy + p] : (int16_t)-1;
    }
}

// ---------------------------------------------------------------------------
// k_stream_probe : the call-filter pass's stream shape with no arithmetic -- column-owner tiling (thread = one
// 16-byte chunk column, workgroup = 1024 samples x a block of loci), n_in nontemporal 16-byte input streams OR-ed
// together, n_out output streams.  What the memory system of THIS box gives a 12 B-in / 8 B-out stream is the
// yardstick the bench prints next to k_call_filter (boxes differ by 15 %; profiles/r03_notes.md section 1).
// ---------------------------------------------------------------------------
struct ProbeArgs {
    const u32x4* in[4];
    u32x4* out[4];
};
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_stream_probe(ProbeArgs s, int L, int S4, int lpb, CfGeom geom) {
    extern __shared__ uint32_t probe_lds[];   // sized by the launcher to cap the resident workgroups per CU
    int bix, biy;
    if (!cf_place(geom, bix, biy)) return;     // the call-filter kernels' own workgroup -> (tile, range) map
    const int c = bix * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = biy * geom.walk * lpb, l1 = min(L, l0 + geom.walk * lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < NIN; ++k) r |= __builtin_nontemporal_load(s.in[k] + o);
#pragma unroll
        for (int k = 0; k < NOUT; ++k) __builtin_nontemporal_store(r + (uint32_t)k, s.out[k] + o);
    }
    if (lpb < 0) probe_lds[threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__global__ __launch_bounds__(256) void k_synth(trk_synth_spec sp, int16_t* __restrict__ gt, int32_t* __restrict__ dp,
                                              float* __restrict__ q, int32_t* __restrict__ dstutter,
                                              int32_t* __restrict__ dflank) {
    const int64_t n = (int64_t)sp.n_loci * sp.n_samples;
    for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < n;
         cell += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(cell / sp.n_samples);
        const int s = (int)(cell - (int64_t)l * sp.n_samples);
        const uint64_t gidx = (uint64_t)(sp.locus_base + l) * (uint64_t)sp.n_samples + (uint64_t)s;
        const uint64_t x = sp.seed + 0x9E3779B97F4A7C15ull * (gidx + 1ull);
        const uint64_t h1 = mix64(x);
        const uint64_t h2 = mix64(x + 0x632BE59BD9B4E019ull);
        const uint64_t h3 = mix64(x + 0xD1B54A32D192ED03ull);
        const uint32_t u0 = (uint32_t)(h1 & 0xffffffu);
        const uint32_t u1 = (uint32_t)((h1 >> 24) & 0xffffffu);
        const uint32_t um = (uint32_t)((h1 >> 48) & 0xffffu);
        const uint32_t ui = (uint32_t)(h2 & 0xffffu);
        const int off = sp.allele_off[l];
        const int A = sp.allele_off[l + 1] - off;
        const uint32_t* cdf = sp.allele_cdf24 + off;
        int a0 = 0, a1 = 0;
        while (a0 < A - 1 && u0 >= cdf[a0]) ++a0;
        while (a1 < A - 1 && u1 >= cdf[a1]) ++a1;
        if (ui < sp.inbreed_thr16[l]) a1 = a0;
        const uint32_t mt = sp.miss_thr16[l];
        const bool nocall = um < mt;
        const bool partial = !nocall && um < mt + 328u;  // ~0.5% '0/.' calls
        int16_t g0 = (int16_t)a0, g1 = (int16_t)a1;
        if (nocall) {
            g0 = -1;
            g1 = -1;
        } else if (partial) {
            g1 = -1;
        }
        gt[cell * 2 + 0] = g0;
        gt[cell * 2 + 1] = g1;
        // DP: Irwin-Hall(4 bytes) scaled to mean 30, sd 12, floor at 0
        uint32_t bsum = (uint32_t)((h2 >> 16) & 0xff) + (uint32_t)((h2 >> 24) & 0xff) +
                        (uint32_t)((h2 >> 32) & 0xff) + (uint32_t)((h2 >> 40) & 0xff);
        uint32_t v = bsum * 12u;
        int32_t d = v >= 1680u ? (int32_t)((v - 1680u) / 148u) : 0;
        // Q: 1 - (2*clz32 + bit)/100, float32
        uint32_t r = (uint32_t)(h3 & 0xffffffffu);
        int lz = r ? __clz(r) : 32;
        int qi = 100 - (2 * lz + (int)((h3 >> 32) & 1u));
        if (qi < 0) qi = 0;
        int32_t st = __popc((uint32_t)((h3 >> 33) & 0xffu)) >> 1;
        int32_t fl = __popc((uint32_t)((h3 >> 41) & 0xfu)) >> 1;
        if (st > d) st = d;
        if (fl > d) fl = d;
        if (nocall) {
            if (dp) dp[cell] = INT32_MIN;
            if (q) q[cell] = __builtin_nanf("");
            if (dstutter) dstutter[cell] = INT32_MIN;
            if (dflank) dflank[cell] = INT32_MIN;
        } else {
            if (dp) dp[cell] = d;
            if (q) q[cell] = (float)qi / 100.0f;
            if (dstutter) dstutter[cell] = st;
            if (dflank) dflank[cell] = fl;
        }
    }
}

__global__ __launch_bounds__(256) void k_synth_gangstr(trk_synth_spec sp, const int16_t* __restrict__ gt,
                                                      const int32_t* __restrict__ dp,
                                                      const int32_t* __restrict__ allele_repcn,
                                                      float* __restrict__ qexp, int32_t* __restrict__ repcn,
                                                      int32_t* __restrict__ rc, int32_t* __restrict__ repci) {
    const int64_t n = (int64_t)sp.n_loci * sp.n_samples;
    for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < n;
         cell += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(cell / sp.n_samples);
        const int s = (int)(cell - (int64_t)l * sp.n_samples);
        const uint64_t gidx = (uint64_t)(sp.locus_base + l) * (uint64_t)sp.n_samples + (uint64_t)s;
        const uint64_t x = sp.seed + 0x9E3779B97F4A7C15ull * (gidx + 1ull);
        const uint64_t h4 = mix64(x + 0xA0761D6478BD642Full);
        const uint64_t h5 = mix64(x + 0xE7037ED1A0B428DBull);
        const int g0 = gt[cell * 2], g1 = gt[cell * 2 + 1];
        const int off = sp.allele_off[l];
        const bool nocall = g0 < 0 && g1 < 0;
        if (nocall) {
            for (int j = 0; j < 3; ++j) qexp[cell * 3 + j] = __builtin_nanf("");
            for (int j = 0; j < 2; ++j) repcn[cell * 2 + j] = INT32_MIN;
            for (int j = 0; j < 4; ++j) {
                rc[cell * 4 + j] = INT32_MIN;
                repci[cell * 4 + j] = INT32_MIN;
            }
            continue;
        }
        const int32_t d = dp[cell] < 0 ? 0 : dp[cell];
        // QEXP: thousandths summing to 1000, or the -1 sentinel (1 in 16)
        const uint32_t b0 = (uint32_t)(h4 & 0xff), b1 = (uint32_t)((h4 >> 8) & 0xff);
        const int p0 = (int)(b0 * 1000u / 255u);
        const int p1 = (int)(((uint32_t)(1000 - p0) * b1) >> 8);
        const int p2 = 1000 - p0 - p1;
        if (((h4 >> 16) & 0xf) == 0) {
            for (int j = 0; j < 3; ++j) qexp[cell * 3 + j] = -1.0f;
        } else {
            qexp[cell * 3 + 0] = (float)p0 / 1000.0f;
            qexp[cell * 3 + 1] = (float)p1 / 1000.0f;
            qexp[cell * 3 + 2] = (float)p2 / 1000.0f;
        }
        // REPCN / REPCI
        const uint32_t e = (uint32_t)((h4 >> 20) & 0xffff);
        for (int j = 0; j < 2; ++j) {
            const int g = j == 0 ? g0 : g1;
            const int32_t r = g >= 0 ? allele_repcn[off + g] : INT32_MIN;
            repcn[cell * 2 + j] = r;
            if (g < 0) {
                repci[cell * 4 + 2 * j] = INT32_MIN;
                repci[cell * 4 + 2 * j + 1] = INT32_MIN;
                continue;
            }
            const int lo_w = (int)((e >> (4 * j)) & 3u), hi_w = (int)((e >> (4 * j + 2)) & 3u);
            int lo = r - lo_w, hi = r + hi_w;
            if (lo < 0) lo = 0;  // 'lo-hi' text cannot carry a negative bound
            if (((e >> (8 + j)) & 0x1f) == 0) lo = r + 1, hi = r + 2;  // ML estimate outside the CI (~3%)
            repci[cell * 4 + 2 * j] = lo;
            repci[cell * 4 + 2 * j + 1] = hi;
        }
        // RC: enclosing, spanning, FRR, bounding; sums to DP
        const uint32_t c0 = (uint32_t)(h5 & 0xff), c1 = (uint32_t)((h5 >> 8) & 0xff), c2 = (uint32_t)((h5 >> 16) & 0xff);
        int32_t encl = (int32_t)((c0 * (uint32_t)(d + 1)) >> 8);
        int32_t rem = d - encl;
        int32_t span = (int32_t)((c1 * (uint32_t)(rem + 1)) >> 8);
        rem -= span;
        int32_t frr = (int32_t)((c2 * (uint32_t)(rem + 1)) >> 8);
        int32_t bound = rem - frr;
        const uint32_t mode = (uint32_t)((h5 >> 24) & 0x3f);
        if (mode == 0) { encl = 0; span = d; frr = 0; bound = 0; }            // spanning reads only
        else if (mode == 1) { encl = 0; frr = 0; span = d / 2; bound = d - span; }  // spanning + bounding only
        rc[cell * 4 + 0] = encl;
        rc[cell * 4 + 1] = span;
        rc[cell * 4 + 2] = frr;
        rc[cell * 4 + 3] = bound;
    }
}

// [n_cells, ncol] -> [ncol, n_cells], 4-byte elements
__global__ __launch_bounds__(256) void k_planarize(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                  int64_t n_cells, int ncol) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells; c += stride)
        for (int k = 0; k < ncol; ++k) dst[(int64_t)k * n_cells + c] = src[c * ncol + k];
}

// rows of row_words 4-byte words -> rows of row_words + pad_words, the pad filled with `fill` (trk_pad_rows)
__global__ __launch_bounds__(256) void k_pad_rows(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                 int64_t n_rows, int row_words, int pad_words, uint32_t fill) {
    const int drow = row_words + pad_words;
    const int64_t n = n_rows * drow, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / drow;
        const int c = (int)(i - r * drow);
        dst[i] = c < row_words ? src[r * row_words + c] : fill;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers (called from trk_api.hip)
// ---------------------------------------------------------------------------
namespace trk {

// Geometry of a column-owner call-filter launch (CfGeom): L loci, gx column tiles, at most lpb_cap loci per LDS block,
// `occ` resident workgroups per CU.
//   * PERSISTENT when the batch is long enough (every workgroup gets at least two LDS blocks): W workgroups per CU
//     (TRK_CF_WGCU; default min(occ, 5)), each walking ONE contiguous range of loci block by block with its per-sample
//     counters in registers -- n_ranges partial-counter rows for k_cf_reduce instead of one per block.
//   * otherwise (a strong-scaling shard, a command-line batch): one block per workgroup, whole rounds of resident
//     workgroups and at least TRK_CF_MIN_ROUNDS (2) of them.  All workgroups of ONE round run their prologue (class
//     LUT), their stream and their flush at the same time, so the memory system idles twice; from two rounds on the
//     phases of different workgroups overlap (12.5k loci x 10k samples: 1070 workgroups of 117 loci = 0.84 rounds
//     0.77 ms; 2560 of 49 loci 0.63 ms).
//   * map (TRK_CF_MAP, default 2): the XCD-aware 1-D launch -- see CfGeom.
struct CfLaunch {
    CfGeom geom;
    int lpb;
    dim3 grid;
};
static CfLaunch cf_geometry(int L, int gx, int lpb_cap, int n_cu, int occ, int mult = 1) {
    CfLaunch r;
    if (lpb_cap < 1) lpb_cap = 1;
    if (mult > 1) lpb_cap = lpb_cap >= mult ? lpb_cap / mult * mult : mult;   // blocks of whole multiples of `mult` loci
    if (const char* e = trk_opt("TRK_CF_LPB")) {
        const int q = atoi(e);
        if (q > 0 && q < lpb_cap) lpb_cap = q;
    }
    int map = 2;
    if (const char* e = trk_opt("TRK_CF_MAP")) map = atoi(e);
    if (map < 0 || map > 2) map = 2;
    if (occ < 1) occ = 4;
    // one workgroup slot per CU stays free when five fit: same-process A/B at 100k x 10k (r04_notes section 2) -- four
    // per CU 3.53 + 0.014 ms (kernel + k_cf_reduce), five 3.46 + 0.13, and the step's small kernels on the other
    // queues (finalisers, HWE tests) find room: 4.29 against 4.36 ms per step
    int wgcu = occ >= 5 ? 4 : occ;
    if (const char* e = trk_opt("TRK_CF_WGCU")) wgcu = atoi(e) > 0 ? atoi(e) : wgcu;
    long pr = (long)wgcu * n_cu / gx;             // ranges of a persistent launch
    if (map == 2) pr = pr / 8 * 8;
    if (pr < 1) pr = 1;
    const long lpr = (L + pr - 1) / pr;             // loci per range
    int lpb, walk;
    // persistent ranges from 24 loci per range on (round 4, second half: was two full blocks).  A strong-scaling shard of
    // 12.5k loci used to launch 2560 workgroups of one block each and filled every slot of the chip; 960 persistent ones
    // run the pass in the same time and leave the step's finalisers and HWE tests on the other queues a slot per CU:
    // 0.66 -> 0.58 ms per step at 12.5k x 10k (tools/shard_probe.py, r04_notes section 14).  TRK_CF_PERSIST_MIN moves it.
    long persist_min = 24;
    if (const char* e = trk_opt("TRK_CF_PERSIST_MIN")) persist_min = atol(e) > 0 ? atol(e) : persist_min;
    if (lpr >= persist_min && !trk_opt("TRK_CF_NO_PERSIST")) {
        walk = (int)((lpr + lpb_cap - 1) / lpb_cap);
        lpb = (int)((lpr + walk - 1) / walk);
    } else {
        int min_rounds = 2;
        if (const char* e = trk_opt("TRK_CF_MIN_ROUNDS")) min_rounds = atoi(e) > 0 ? atoi(e) : 2;
        const long slots = (long)n_cu * occ;
        lpb = lpb_cap;
        long k = ((long)gx * ((L + lpb - 1) / lpb) + slots - 1) / slots;
        if (k < min_rounds) k = min_rounds;
        long gyr = (k * slots) / gx;
        if (gyr > L) gyr = L;
        if (gyr >= 1) {
            int lpb2 = (int)((L + gyr - 1) / gyr);
            if (lpb2 < 8) lpb2 = L < 8 ? (L > 0 ? L : 1) : 8;   // a block's fixed cost needs some loci to spread over
            if (lpb2 < lpb) lpb = lpb2;
        }
        walk = 1;
    }
    if (mult > 1) {
        const int up = (lpb + mult - 1) / mult * mult;
        lpb = up <= lpb_cap ? up : lpb_cap;
    }
    const int n_blocks = (L + lpb - 1) / lpb;
    r.lpb = lpb;
    r.geom.gx = gx;
    r.geom.walk = walk;
    r.geom.n_ranges = (n_blocks + walk - 1) / walk;
    r.geom.map = map;
    if (map == 2) r.grid = dim3((unsigned)((r.geom.n_ranges + 7) / 8 * 8) * (unsigned)gx);
    else if (map == 1) r.grid = dim3((unsigned)r.geom.n_ranges, (unsigned)gx);
    else r.grid = dim3((unsigned)gx, (unsigned)r.geom.n_ranges);
    return r;
}

static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// twin (TRK_STATS_TWIN): allele_count is [2][G, sumA] and locus_int [2][G, L, COLS]; both copies receive the counts.
// The streaming kernels write every output element themselves (and the twin copy in the same pass); the general
// paths accumulate with atomics into zeroed arrays and are followed by two device-to-device copies.
static void sync_gt_temporal() {
    static int last = -1;
    const int want = trk_opt("TRK_GT_TEMPORAL") ? atoi(trk_opt("TRK_GT_TEMPORAL")) : 0;
    if (want != last) {
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gt_temporal), &want, sizeof want);
        last = want;
    }
}

// The ungrouped diploid streaming kernels (k_locus_count_v3 / _v2 / _fast) on one batch -- or on a column-range VIEW of
// one (trk_batch.row_stride; the class passes of launch_locus_count).  false: the batch is outside what they cover.
static bool launch_count_streaming(const trk_batch& b, int max_alleles, int32_t* allele_count, int32_t* locus_int,
                                   hipStream_t stream, bool twin, size_t ac_elems, size_t li_elems, hipError_t* err) {
    const bool fast2 = (b.ploidy == 2) && !b.group_bits;
    const int64_t twin_ac = twin ? (int64_t)ac_elems : 0, twin_li = twin ? (int64_t)li_elems : 0;
    auto zero_outputs = [&]() -> hipError_t {
        hipError_t e = hipMemsetAsync(allele_count, 0, ac_elems * sizeof(int32_t), stream);
        if (e != hipSuccess) return e;
        return hipMemsetAsync(locus_int, 0, li_elems * sizeof(int32_t), stream);
    };
    auto copy_twin = [&]() -> hipError_t {
        if (!twin) return hipGetLastError();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(allele_count + ac_elems, allele_count, ac_elems * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(locus_int + li_elems, locus_int, li_elems * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    };
    if (fast2 && max_alleles > 0 && (b.n_samples % 4) == 0 && b.n_samples > 0) {
        const char* ver_env = trk_opt("TRK_CNT_VER");
        const bool use_v2 = !b.locus_ploidy && max_alleles + 2 < 65535 && !(ver_env && atoi(ver_env) == 1);
        if (!use_v2 && b.row_stride) return false;   // (the older kernel knows no views)
        const int extra = use_v2 ? 7 : 1;  // bins besides the alleles
        // words per wave: bins x K copies + one LUT word per bin, K = 32 while it fits in 16 KiB
        int kshift = 5;
        while (kshift > 0 && ((max_alleles + extra) << kshift) + max_alleles + extra > 4096) --kshift;
        int words = ((max_alleles + extra) << kshift) + max_alleles + extra;
        if (words <= 4096) {
            words = (words + 3) & ~3;
            size_t lds_fast = (size_t)COUNT_WAVES_PER_WG * words * sizeof(uint32_t);
            int wgs_fast = (b.n_loci + COUNT_WAVES_PER_WG - 1) / COUNT_WAVES_PER_WG;
            int cnt_u = 4;   // with the double-buffered streamer four chunks in flight beat two (tools/count_probe.py)
            if (const char* e = trk_opt("TRK_CNT_U")) cnt_u = atoi(e);
            dim3 grid(wgs_fast), block(WAVE * COUNT_WAVES_PER_WG);
            // short rows: R loci per wave (k_locus_count_v3) while the wider per-wave histogram still lets >= 2
            // workgroups share a CU.  TRK_CNT_R = 1 / 2 / 4 overrides the row-length rule (tools/perf_sweep.py).
            // four loci per wave for rows of up to 2048 samples (400k x 1k: R = 4 0.38 ms, R = 1 0.46; 200k x 2k: 0.37 / 0.37;
            // 100k x 4k: 0.37 / 0.29; 100k x 10k: 0.76 / 0.62)
            int rr = b.n_samples <= 2048 ? 4 : 1;
            if (const char* e = trk_opt("TRK_CNT_R")) rr = atoi(e);
            const int nbmax = max_alleles + 7;
            const int words3 = (2 * nbmax * 32 + 4 * nbmax + 3) & ~3;
            if (use_v2 && (rr == 2 || rr == 4) && (size_t)COUNT_WAVES_PER_WG * words3 * 4 <= 72 * 1024) {
                const int per_wg = COUNT_WAVES_PER_WG * rr;
                dim3 g3((b.n_loci + per_wg - 1) / per_wg);
                const size_t lds3 = (size_t)COUNT_WAVES_PER_WG * words3 * sizeof(uint32_t);
                if (rr == 4 && cnt_u == 4)
                    hipLaunchKernelGGL((k_locus_count_v3<4, 4>), g3, block, lds3, stream, b, allele_count, locus_int, nbmax, words3, twin_ac, twin_li, V3Fin{});
                else if (rr == 4)
                    hipLaunchKernelGGL((k_locus_count_v3<4, 2>), g3, block, lds3, stream, b, allele_count, locus_int, nbmax, words3, twin_ac, twin_li, V3Fin{});
                else if (cnt_u == 4)
                    hipLaunchKernelGGL((k_locus_count_v3<2, 4>), g3, block, lds3, stream, b, allele_count, locus_int, nbmax, words3, twin_ac, twin_li, V3Fin{});
                else
                    hipLaunchKernelGGL((k_locus_count_v3<2, 2>), g3, block, lds3, stream, b, allele_count, locus_int, nbmax, words3, twin_ac, twin_li, V3Fin{});
                // (only N_ALLELES / HWE status / NALLELES columns, written by the finaliser, are left untouched)
                { *err = hipGetLastError(); return true; }
            }
            if (use_v2) {
                // wave-per-locus kernel: two chunks per register set (83 VGPRs, five waves per SIMD) beat four (100
                // VGPRs, four waves) since the streamer swaps its two sets instead of copying: 0.70 vs 0.79 ms
                const int v2_u = trk_opt("TRK_CNT_U") ? cnt_u : 2;
                if (v2_u == 4)
                    hipLaunchKernelGGL(k_locus_count_v2<4>, grid, block, lds_fast, stream, b, allele_count, locus_int,
                                       kshift, words, twin_ac, twin_li);
                else if (v2_u == 1)
                    hipLaunchKernelGGL(k_locus_count_v2<1>, grid, block, lds_fast, stream, b, allele_count, locus_int,
                                       kshift, words, twin_ac, twin_li);
                else
                    hipLaunchKernelGGL(k_locus_count_v2<2>, grid, block, lds_fast, stream, b, allele_count, locus_int,
                                       kshift, words, twin_ac, twin_li);
                { *err = hipGetLastError(); return true; }
            } else {
                hipError_t ez = zero_outputs();
                if (ez != hipSuccess) { *err = ez; return true; }
                if (cnt_u == 4)
                    hipLaunchKernelGGL(k_locus_count_fast<4>, grid, block, lds_fast, stream, b, allele_count,
                                       locus_int, kshift, words);
                else
                    hipLaunchKernelGGL(k_locus_count_fast<2>, grid, block, lds_fast, stream, b, allele_count,
                                       locus_int, kshift, words);
            }
            { *err = copy_twin(); return true; }
        }
    }
    return false;
}

hipError_t launch_locus_count(const trk_batch& b, int max_alleles, int32_t* allele_count, int32_t* locus_int,
                              int n_cu, hipStream_t stream, bool twin, int32_t* class_ws) {
    sync_gt_temporal();
    const int G = b.group_bits ? b.n_groups : 1;
    const bool fast2 = (b.ploidy == 2) && !b.group_bits;
    const size_t ac_elems = (size_t)G * (size_t)b.n_alleles_total, li_elems = (size_t)G * b.n_loci * TRK_LI_COLS;
    auto zero_outputs = [&]() -> hipError_t {
        hipError_t e = hipMemsetAsync(allele_count, 0, ac_elems * sizeof(int32_t), stream);
        if (e != hipSuccess) return e;
        return hipMemsetAsync(locus_int, 0, li_elems * sizeof(int32_t), stream);
    };
    auto copy_twin = [&]() -> hipError_t {
        if (!twin) return hipGetLastError();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(allele_count + ac_elems, allele_count, ac_elems * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(locus_int + li_elems, locus_int, li_elems * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    };
    if (!b.n_class_runs) {
        hipError_t es = hipSuccess;
        if (launch_count_streaming(b, max_alleles, allele_count, locus_int, stream, twin, ac_elems, li_elems, &es)) return es;
    }
    // sample classes laid out as column ranges (trk_batch.class_runs): the ungrouped streaming kernel per range,
    // then the classes added into their groups (k_class_combine).  Any number of (overlapping) groups.
    if (b.n_class_runs > 0 && b.class_runs && class_ws && b.group_bits && b.ploidy == 2 && !b.locus_ploidy &&
        max_alleles > 0 && !b.row_stride) {
        const int NR = b.n_class_runs;
        const size_t ac1 = (size_t)b.n_alleles_total, li1 = (size_t)b.n_loci * TRK_LI_COLS;
        bool ok = NR <= 256;
        for (int r = 0; r < NR && ok; ++r) {
            const int32_t* q = b.class_runs + 4 * r;
            ok = q[0] >= 0 && q[1] > 0 && (q[0] & 3) == 0 && (q[1] & 3) == 0 && q[2] >= 0 && q[2] <= q[1] &&
                 q[0] + q[1] <= b.n_samples;
        }
        if (ok) {
            ClassRuns cr = {};
            int n_used = 0;
            for (int r = 0; r < NR; ++r) {
                const int32_t* q = b.class_runs + 4 * r;
                if ((q[3] & ((1 << G) - 1)) == 0 || q[2] == 0) continue;   // samples in no group are never read
                trk_batch v = b;
                v.gt = b.gt + (size_t)q[0] * 2;
                v.n_samples = q[1];
                v.n_pad_samples = q[1] - q[2];
                v.row_stride = b.n_samples;
                v.group_bits = nullptr;
                v.n_groups = 1;
                v.n_class_runs = 0;
                v.class_runs = nullptr;
                int32_t* ac = class_ws + (size_t)n_used * (ac1 + li1);
                hipError_t es = hipSuccess;
                if (!launch_count_streaming(v, max_alleles, ac, ac + ac1, stream, false, ac1, li1, &es)) { ok = false; break; }
                if (es != hipSuccess) return es;
                if (n_used < 256) cr.bits[n_used] = (uint8_t)q[3];
                ++n_used;
            }
            if (ok) {
                cr.n = n_used;
                const int64_t n_ac = (int64_t)ac1, n_li = (int64_t)li1;
                const int64_t total = n_ac + n_li;
                hipLaunchKernelGGL(k_class_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, cr, class_ws,
                                   n_ac, n_li, G, allele_count, locus_int, twin ? (int64_t)ac_elems : 0,
                                   twin ? (int64_t)li_elems : 0);
                return hipGetLastError();
            }
        }
    }
    if (b.row_stride) return hipErrorInvalidValue;   // a view outside the streaming kernels
    // sample groups (statSTR --samples): the streaming kernel with one histogram per class of group bits
    if (b.group_bits && G >= 1 && G <= 3 && b.ploidy == 2 && !b.locus_ploidy && max_alleles > 0 &&
        max_alleles + 2 < 65535 && b.n_samples > 0 && (b.n_samples % 4) == 0 &&
        ((uintptr_t)b.group_bits & 3u) == 0 && !trk_opt("TRK_CNT_NOGROUPFAST")) {
        const int ncls = 1 << G, nb = max_alleles + V2G_EXTRA;
        int kshift = 5;
        while (kshift > 2 && ncls * (nb << kshift) + nb > 3072) --kshift;   // <= 12 KiB per wave
        int words = ncls * (nb << kshift) + nb;
        if (words <= 3072) {
            words = (words + 3) & ~3;
            const size_t lds_g = (size_t)COUNT_WAVES_PER_WG * words * sizeof(uint32_t);
            const int wgs_g = (b.n_loci + COUNT_WAVES_PER_WG - 1) / COUNT_WAVES_PER_WG;
            hipError_t ez = zero_outputs();
            if (ez != hipSuccess) return ez;
            hipLaunchKernelGGL(k_locus_count_v2g, dim3(wgs_g), dim3(WAVE * COUNT_WAVES_PER_WG), lds_g, stream, b,
                               allele_count, locus_int, kshift, words);
            return copy_twin();
        }
    }
    int maxA = max_alleles > 0 ? max_alleles : 64;
    int bins = G * (maxA + (fast2 ? 0 : XB_N));
    int hist_entries = next_pow2(bins * 32);
    if (hist_entries < 256) hist_entries = 256;
    if (hist_entries > 4096) hist_entries = 4096;
    int lut_entries = next_pow2(maxA);
    if (lut_entries < 64) lut_entries = 64;
    if (lut_entries > 2048) lut_entries = 2048;
    size_t lds = (size_t)COUNT_WAVES_PER_WG * (hist_entries + lut_entries) * sizeof(uint32_t);
    int wgs = (b.n_loci + COUNT_WAVES_PER_WG - 1) / COUNT_WAVES_PER_WG;
    int max_wgs = n_cu * 8;
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs < 1) wgs = 1;
    hipError_t ez = zero_outputs();
    if (ez != hipSuccess) return ez;
    if (fast2)
        hipLaunchKernelGGL(k_locus_count<true>, dim3(wgs), dim3(WAVE * COUNT_WAVES_PER_WG), lds, stream, b,
                           allele_count, locus_int, hist_entries, lut_entries);
    else
        hipLaunchKernelGGL(k_locus_count<false>, dim3(wgs), dim3(WAVE * COUNT_WAVES_PER_WG), lds, stream, b,
                           allele_count, locus_int, hist_entries, lut_entries);
    return copy_twin();
}

// count + finaliser in one launch, the HWE tests in a second (small ungrouped diploid batches of short rows).
// false: outside what the fused kernel covers -- the caller takes launch_locus_count + launch_locus_finalize.
bool launch_locus_stats_fused(const trk_batch& b, int32_t* allele_count, int32_t* locus_int, double* locus_f64,
                              void* worklist, double nalleles_thresh, hipStream_t stream, hipError_t* err, int stage) {
    // stage 0: only answer whether the batch is covered; 1: count + finaliser; 2: the HWE tests
    // TRK_FUSED_STATS = 0: never; N: up to N loci.  Default: every batch of short rows (tools/fused_probe.py: 0.037 vs
    // 0.067 ms at 1k loci, 0.045 / 0.075 at 10k, 0.21 / 0.24 at 100k, 0.72 / 0.79 at 400k x 1k samples)
    const char* env = trk_opt("TRK_FUSED_STATS");
    const int64_t limit = env ? atoll(env) : (int64_t)1 << 30;
    const int max_alleles = b.max_alleles;
    if (limit <= 0 || b.n_loci > limit) return false;
    if (b.ploidy != 2 || b.group_bits || b.locus_ploidy || b.row_stride || b.n_class_runs > 0) return false;
    if (max_alleles <= 0 || max_alleles + 2 >= 65535 || (b.n_samples % 4) != 0 || b.n_samples <= 0 || b.n_samples > 2048)
        return false;
    if (trk_opt("TRK_CNT_VER") || trk_opt("TRK_CNT_R") || trk_opt("TRK_CNT_U")) return false;   // A/B knobs of the count kernels
    const int nbmax = max_alleles + 7;
    V3Fin fin;
    fin.locus_f64 = locus_f64;
    fin.items = reinterpret_cast<HweItem*>(reinterpret_cast<char*>(worklist) + 16);
    fin.nalleles_thresh = nalleles_thresh;
    fin.amax4 = (max_alleles + 3) & ~3;
    fin.fin_words = 12 * fin.amax4;
    const int words3 = (2 * nbmax * 32 + 8 * nbmax + 3) & ~3;   // histogram rows (later the scratch), LUTs, bin totals
    const size_t lds3 = (size_t)COUNT_WAVES_PER_WG * words3 * sizeof(uint32_t);
    if (lds3 > 80 * 1024) return false;
    if (stage == 0) return true;
    const unsigned int n_slots = 2u * (unsigned int)b.n_loci;
    if (stage == 2) {
        *err = launch_hwe_slots(fin.items, n_slots, locus_f64, stream);
        return true;
    }
    sync_gt_temporal();
    const int per_wg = COUNT_WAVES_PER_WG * 4;
    hipLaunchKernelGGL((k_locus_count_v3<4, 4, true>), dim3((b.n_loci + per_wg - 1) / per_wg), dim3(WAVE * COUNT_WAVES_PER_WG),
                       lds3, stream, b, allele_count, locus_int, nbmax, words3, (int64_t)0, (int64_t)0, fin);
    *err = hipGetLastError();
    return true;
}

hipError_t launch_locus_finalize(const trk_batch& b, const int32_t* allele_count, int32_t* locus_int,
                                 double* locus_f64, int32_t* scratch, void* worklist, double nalleles_thresh,
                                 hipStream_t stream) {
    const int G = b.group_bits ? b.n_groups : 1;
    int64_t n = (int64_t)G * b.n_loci;
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + FIN_THREADS - 1) / FIN_THREADS);
    const int maxA = b.max_alleles;
    unsigned int* count = reinterpret_cast<unsigned int*>(worklist);
    HweItem* items = reinterpret_cast<HweItem*>(reinterpret_cast<char*>(worklist) + 16);
    hipError_t e = hipMemsetAsync(count, 0, 16, stream);
    if (e != hipSuccess) return e;
    // Sixteen lanes per locus shorten the LATENCY of a small batch (800 loci: 78 -> 51 us incl. the tests, 1677: 78 ->
    // 58, 3355: 78 -> 67) and cost throughput on a large one (6000: 70 -> 75 us, 100k: 0.20 -> 0.70 ms -- the serial
    // sums idle fifteen lanes): up to 4096 rows.  TRK_FIN_COOP = 0 never, N: up to N rows (tools/fin_coop_probe.py).
    const char* coop_env = trk_opt("TRK_FIN_COOP");
    const int64_t coop_max = coop_env ? atoll(coop_env) : 4096;
    if (maxA > 0 && maxA <= 64 && n <= coop_max) {
        const int amax4 = (maxA + 3) & ~3;
        const size_t lds = (size_t)(FC_THREADS / FC_LPL) * 14 * amax4 * sizeof(uint32_t);
        const int64_t per_wg = FC_THREADS / FC_LPL;
        hipLaunchKernelGGL(k_locus_finalize_coop, dim3((unsigned)((n + per_wg - 1) / per_wg)), dim3(FC_THREADS), lds, stream, b,
                           allele_count, locus_int, locus_f64, nalleles_thresh, amax4, count, items);
    } else if (maxA > 0 && maxA <= 96) {
        size_t lds = (size_t)2 * maxA * FIN_THREADS * sizeof(int32_t);
        hipLaunchKernelGGL(k_locus_finalize<true>, dim3(blocks), dim3(FIN_THREADS), lds, stream, b, allele_count,
                           locus_int, locus_f64, scratch, nalleles_thresh, maxA, count, items);
    } else {
        hipLaunchKernelGGL(k_locus_finalize<false>, dim3(blocks), dim3(FIN_THREADS), 0, stream, b, allele_count,
                           locus_int, locus_f64, scratch, nalleles_thresh, maxA, count, items);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_hwe_tests(count, items, locus_f64, reinterpret_cast<unsigned int*>(items + 2 * n), n, stream);
}

size_t finalize_worklist_bytes(int64_t n_group_loci) {
    return 16 + (size_t)(2 * n_group_loci) * (sizeof(HweItem) + 4);   // header, items, overflow list
}

hipError_t launch_call_filter(const trk_batch& b, const trk_plane* planes, int n_planes,
                              const trk_call_filter* filters, int n_filters, int dp_plane, const trk_call_out& out,
                              int n_cu, hipStream_t stream, const Scratch& scratch) {
    CallArgs a;
    a.b = b;
    for (int i = 0; i < n_planes; ++i) a.planes[i] = planes[i];
    for (int i = 0; i < n_filters; ++i) a.filters[i] = filters[i];
    a.n_planes = n_planes;
    a.n_filters = n_filters;
    a.dp_plane = dp_plane;
    a.out = out;
    if (n_filters > 7) a.out.filter_mask8 = nullptr;
    const int S = b.n_samples, L = b.n_loci;
    if (S == 0 || L == 0) return hipSuccess;
    int gx = (S + CF_THREADS * CF_V - 1) / (CF_THREADS * CF_V);
    // enough blocks to fill the chip (>= 8 per CU) while keeping the per-block
    // counter flush (one atomic per sample per counter) small next to the stream
    int want_blocks = n_cu * 32;
    int gy = (want_blocks + gx - 1) / gx;
    if (gy > L) gy = L;
    if (gy < 1) gy = 1;
    int lpb = (L + gy - 1) / gy;
    if (lpb < 128) lpb = L < 128 ? L : 128;
    if (lpb > 4096) lpb = 4096;
    gy = (L + lpb - 1) / lpb;
    a.loci_per_block = lpb;
    size_t lds = (size_t)n_filters * CF_THREADS * CF_V * sizeof(uint32_t);
    // a Float depth plane (ExpansionHunter's LC) is summed in float64 by the per-call kernel only
    const bool float_dp = dp_plane >= 0 && (planes[dp_plane].dtype & 0xff) == TRK_DT_F32;
    const bool vec = (b.ploidy == 2) && (S % 4 == 0) && !float_dp;
    a.n_cells = (int64_t)L * S;
    // ---- vector sources: one [L*S] array per (plane, column) that can be fetched as 16-byte vectors --------
    a.n_src = 0;
    a.grp_n = 0;
    int grp_plane[CF_MAX_GROUPS];
    a.src_f32_mask = 0;
    a.reg_filter_mask = 0;
    a.int_thr_mask = 0;
    a.dp_src = -1;
    for (int k = 0; k < TRK_MAX_FILTERS; ++k) {
        a.f_src_a[k] = a.f_src_a2[k] = a.f_src_b[k] = -1;
        a.f_ci_n[k] = 0;
    }
    auto source_of = [&](int p, int col) -> int {
        if (!vec || p < 0 || p >= n_planes || trk_opt("TRK_CF_NOSRC")) return -1;
        const trk_plane& pl = planes[p];
        const bool planar = (pl.dtype & TRK_DT_PLANAR) != 0;
        if (col < 0 || col >= pl.ncol) return -1;
        if (pl.ncol > 1 && !planar) {   // interleaved: all k columns become sources together
            if (pl.ncol > 4 || ((uintptr_t)pl.data & 15u) || trk_opt("TRK_CF_NOGROUPS")) return -1;
            for (int g = 0; g < a.grp_n; ++g)
                if (grp_plane[g] == p) return a.grp_base[g] + col;
            if (a.grp_n >= CF_MAX_GROUPS || a.n_src + pl.ncol > CF_NSRC) return -1;
            const int g = a.grp_n++;
            grp_plane[g] = p;
            a.grp_ptr[g] = pl.data;
            a.grp_base[g] = (int8_t)a.n_src;
            a.grp_k[g] = (int8_t)pl.ncol;
            for (int c = 0; c < pl.ncol; ++c) {
                a.src_ptr[a.n_src] = nullptr;
                if ((pl.dtype & 0xff) == TRK_DT_F32) a.src_f32_mask |= 1u << a.n_src;
                ++a.n_src;
            }
            return a.grp_base[g] + col;
        }
        const char* ptr = static_cast<const char*>(pl.data) + (planar ? (size_t)col * a.n_cells * 4 : 0);
        if ((uintptr_t)ptr & 15u) return -1;
        for (int q = 0; q < a.n_src; ++q)
            if (a.src_ptr[q] == ptr) return q;
        if (a.n_src >= CF_NSRC) return -1;
        a.src_ptr[a.n_src] = ptr;
        if ((pl.dtype & 0xff) == TRK_DT_F32) a.src_f32_mask |= 1u << a.n_src;
        return a.n_src++;
    };
    if (vec) {
        if (dp_plane >= 0) a.dp_src = (int8_t)source_of(dp_plane, 0);
        for (int k = 0; k < n_filters; ++k) {
            const trk_call_filter& f = filters[k];
            bool ok = true;
            switch (f.op) {
                case TRK_F_LT: case TRK_F_GT: case TRK_F_CALLED_LT:
                    ok = (a.f_src_a[k] = (int8_t)source_of(f.plane_a, f.col_a)) >= 0;
                    if (ok && !((a.src_f32_mask >> a.f_src_a[k]) & 1u) && f.thr == f.thr) {
                        // (double)v < thr  <=>  v < ceil(thr);   (double)v > thr  <=>  v > floor(thr)
                        const double t = f.op == TRK_F_GT ? floor(f.thr) : ceil(f.thr);
                        if (t > -2147483647.0 && t < 2147483647.0) {
                            a.f_ithr[k] = (int32_t)t;
                            a.int_thr_mask |= 1u << k;
                        }
                    }
                    break;
                case TRK_F_RATIO_GT: case TRK_F_CALLED_EQ:
                    ok = (a.f_src_a[k] = (int8_t)source_of(f.plane_a, f.col_a)) >= 0 &&
                         (a.f_src_b[k] = (int8_t)source_of(f.plane_b, f.col_b)) >= 0;
                    break;
                case TRK_F_CALLED_SUM_LT:
                    ok = (a.f_src_a[k] = (int8_t)source_of(f.plane_a, f.col_a)) >= 0 &&
                         (a.f_src_a2[k] = (int8_t)source_of(f.plane_a, f.col_a2)) >= 0;
                    break;
                case TRK_F_CALLED_SUM_EQ:
                    ok = (a.f_src_a[k] = (int8_t)source_of(f.plane_a, f.col_a)) >= 0 &&
                         (a.f_src_a2[k] = (int8_t)source_of(f.plane_a, f.col_a2)) >= 0 &&
                         (a.f_src_b[k] = (int8_t)source_of(f.plane_b, f.col_b)) >= 0;
                    break;
                case TRK_F_CALLED_OUTSIDE_CI: {
                    const int nc = planes[f.plane_a].ncol;
                    ok = nc >= 1 && nc <= 2;
                    for (int c = 0; c < nc && ok; ++c) {
                        const int ml = source_of(f.plane_a, c), lo = source_of(f.plane_b, 2 * c),
                                  hi = source_of(f.plane_b, 2 * c + 1);
                        ok = ml >= 0 && lo >= 0 && hi >= 0;
                        a.f_ci[k][3 * c] = (int8_t)ml;
                        a.f_ci[k][3 * c + 1] = (int8_t)lo;
                        a.f_ci[k][3 * c + 2] = (int8_t)hi;
                    }
                    a.f_ci_n[k] = ok ? (int8_t)nc : 0;
                    break;
                }
                default:
                    ok = false;
            }
            if (ok) a.reg_filter_mask |= 1u << k;
        }
    }
    a.delta_stride = 0;
    bool lds_delta = false;
    if (out.delta_allele_count && vec && b.max_alleles > 0) {
        // the delta table must fit next to the filter counters: shrink the locus block if needed
        const int stride = b.max_alleles + DX_N;
        const size_t per_locus = ((size_t)stride + b.max_alleles + CF_LINFO) * sizeof(int32_t);  // table + LUT + info
        const size_t budget = 24 * 1024;
        if (per_locus * 8 <= budget) {
            int max_lpb = (int)(budget / per_locus);
            if (lpb > max_lpb) {
                lpb = max_lpb;
                gy = (L + lpb - 1) / lpb;
                a.loci_per_block = lpb;
            }
            lds_delta = true;
            a.delta_stride = stride;
        }
    }
    // ---- k_call_filter_v4: every filter a plain threshold on a single-column plane (or a HipSTR ratio over the
    // depth plane); static compare types, queued delta updates ----
    if (vec && !b.locus_ploidy && n_filters >= 1 && n_filters <= V2_MAX_FILTERS && !trk_opt("TRK_CF_GENERIC")) {
        V4Args v = {};
        V4Filter raw[V2_MAX_FILTERS];
        bool is_f32[V2_MAX_FILTERS];
        bool ok = true, ratio = false;
        for (int k = 0; k < n_filters && ok; ++k) {
            const trk_call_filter& f = filters[k];
            const trk_plane& pl = planes[f.plane_a];
            V4Filter& o = raw[k];
            o = V4Filter{};
            o.need_called = f.op == TRK_F_CALLED_LT;
            o.bit = k;
            o.dthr = f.thr;
            is_f32[k] = false;
            if (f.op == TRK_F_RATIO_GT) {
                // numerator an int32 source, denominator the depth vector the kernel loads anyway
                ok = a.f_src_a[k] >= 0 && a.src_ptr[a.f_src_a[k]] && a.dp_src >= 0 && a.f_src_b[k] == a.dp_src &&
                     !((a.src_f32_mask >> a.f_src_a[k]) & 1u) && !((a.src_f32_mask >> a.dp_src) & 1u);
                if (!ok) break;
                o.plane = a.src_ptr[a.f_src_a[k]];
                o.ratio = 1;
                o.dthr_up = nextafter(f.thr, INFINITY);
                ratio = true;
                continue;
            }
            ok = (f.op == TRK_F_LT || f.op == TRK_F_GT || f.op == TRK_F_CALLED_LT) && a.f_src_a[k] >= 0 &&
                 a.src_ptr[a.f_src_a[k]] && f.thr == f.thr;   // (a column of an interleaved plane has no array of its own)
            if (!ok) break;
            o.plane = a.src_ptr[a.f_src_a[k]];
            const bool gt_op = f.op == TRK_F_GT;
            if ((pl.dtype & 0xff) == TRK_DT_F32) {
                // x > t  <=>  -x < -t  (exact; a NaN on either side compares false both ways)
                is_f32[k] = true;
                o.is_float = 1;
                const float t = (float)f.thr;
                uint32_t tb;
                memcpy(&tb, &t, 4);
                o.flip = gt_op ? 0x80000000u : 0u;
                o.thr = (int32_t)(tb ^ o.flip);
            } else {
                // (double)v < thr  <=>  v < ceil(thr);   (double)v > thr  <=>  v > floor(thr)  <=>  !(v < floor(thr) + 1)
                const double t = gt_op ? floor(f.thr) : ceil(f.thr);
                if (!(t > -2147483647.0 && t < 2147483647.0)) ok = false;
                o.thr = (int32_t)t + (gt_op ? 1 : 0);
                o.invert = gt_op ? 1 : 0;
            }
        }
        if (ok && dp_plane >= 0 && a.dp_src < 0) ok = false;
        const bool delta = out.delta_allele_count != nullptr;
        if (ok && delta && !(b.max_alleles > 0 && b.max_alleles <= 120)) ok = false;
        if (ok) {
            v.b = b;
            v.dp = dp_plane >= 0 ? reinterpret_cast<const int32_t*>(a.src_ptr[a.dp_src]) : nullptr;
            v.out = a.out;
            v.delta_nal = delta ? b.max_alleles : 0;
            // order: the filters of one plane neighbours (one load per plane where the instantiation shares it), the
            // depth plane's first, integer planes before float planes
            int n = 0, nflt = 0;
            bool used[V2_MAX_FILTERS] = {false};
            for (int pass = 0; pass < 3; ++pass)
                for (int k = 0; k < n_filters; ++k) {
                    if (used[k]) continue;
                    if (pass == 0 && raw[k].plane != (const void*)v.dp) continue;
                    if (pass <= 1 && is_f32[k]) continue;
                    const void* pl = raw[k].plane;
                    for (int q = k; q < n_filters; ++q)
                        if (!used[q] && raw[q].plane == pl && is_f32[q] == is_f32[k]) {
                            v.f[n++] = raw[q];
                            used[q] = true;
                            nflt += is_f32[q] ? 1 : 0;
                        }
                }
            int alias = (v.dp && v.f[0].plane == (const void*)v.dp) ? 1 : 0;
            if (alias && n_filters >= 2 && v.f[1].plane == v.f[0].plane) alias = 3;
            if (!delta) alias = 0;                 // (only the dumpSTR shape -- delta outputs -- has the sharing builds)
            int tflt = nflt <= 1 ? nflt : -1;      // -1: the compare type is a run-time flag per filter
            if (!delta) tflt = -1;
            if (tflt < 0 && alias) alias = 0;
            void (*k4)(V4Args) = nullptr;
            // the output set as a template argument where it pays: filter_mask8 alone with delta outputs and static
            // compare types -- dumpSTR's command line (compute.dumpstr_batch)
            const size_t qbytes = delta ? (size_t)(CF_THREADS / WAVE) * V4_QCAP * sizeof(uint2) : 0;
            const size_t per_locus4 = delta ? ((size_t)b.max_alleles + V2_EXTRA + CF_LINFO) * sizeof(uint32_t) : 0;
            size_t lds_budget = 26 * 1024;
            if (const char* e = trk_opt("TRK_CF_LDS_KB")) lds_budget = (size_t)(atoi(e) > 12 ? atoi(e) : 12) * 1024;
            const int max_lpb4 = delta ? (int)((lds_budget - qbytes - 8) / per_locus4) : 255;
            const bool compact = delta && tflt >= 0 && out.filter_mask8 && !out.gt_out && !out.filter_mask && max_lpb4 >= V4_PD &&
                                 lpb >= V4_PD;
#define TRK_V4_PICK(NFV)                                                                                              \
    if (compact) {                                                                                                    \
        if (tflt == 0)                                                                                                \
            k4 = alias == 3   ? (ratio ? k_call_filter_v4<NFV, 0, true, true, (NFV >= 2 ? 3 : 1), 2> : k_call_filter_v4<NFV, 0, true, false, (NFV >= 2 ? 3 : 1), 2>) \
                 : alias == 1 ? (ratio ? k_call_filter_v4<NFV, 0, true, true, 1, 2> : k_call_filter_v4<NFV, 0, true, false, 1, 2>) \
                              : (ratio ? k_call_filter_v4<NFV, 0, true, true, 0, 2> : k_call_filter_v4<NFV, 0, true, false, 0, 2>); \
        else                                                                                                          \
            k4 = alias == 3   ? (ratio ? k_call_filter_v4<NFV, 1, true, true, (NFV >= 2 ? 3 : 1), 2> : k_call_filter_v4<NFV, 1, true, false, (NFV >= 2 ? 3 : 1), 2>) \
                 : alias == 1 ? (ratio ? k_call_filter_v4<NFV, 1, true, true, 1, 2> : k_call_filter_v4<NFV, 1, true, false, 1, 2>) \
                              : (ratio ? k_call_filter_v4<NFV, 1, true, true, 0, 2> : k_call_filter_v4<NFV, 1, true, false, 0, 2>); \
    } else if (!delta) k4 = ratio ? k_call_filter_v4<NFV, -1, false, true, 0> : k_call_filter_v4<NFV, -1, false, false, 0>;  \
    else if (tflt < 0) k4 = ratio ? k_call_filter_v4<NFV, -1, true, true, 0> : k_call_filter_v4<NFV, -1, true, false, 0>; \
    else if (tflt == 0)                                                                                               \
        k4 = alias == 3   ? (ratio ? k_call_filter_v4<NFV, 0, true, true, (NFV >= 2 ? 3 : 1)> : k_call_filter_v4<NFV, 0, true, false, (NFV >= 2 ? 3 : 1)>) \
             : alias == 1 ? (ratio ? k_call_filter_v4<NFV, 0, true, true, 1> : k_call_filter_v4<NFV, 0, true, false, 1>) \
                          : (ratio ? k_call_filter_v4<NFV, 0, true, true, 0> : k_call_filter_v4<NFV, 0, true, false, 0>); \
    else                                                                                                              \
        k4 = alias == 3   ? (ratio ? k_call_filter_v4<NFV, 1, true, true, (NFV >= 2 ? 3 : 1)> : k_call_filter_v4<NFV, 1, true, false, (NFV >= 2 ? 3 : 1)>) \
             : alias == 1 ? (ratio ? k_call_filter_v4<NFV, 1, true, true, 1> : k_call_filter_v4<NFV, 1, true, false, 1>) \
                          : (ratio ? k_call_filter_v4<NFV, 1, true, true, 0> : k_call_filter_v4<NFV, 1, true, false, 0>)
            bool cv1 = true;       // one chunk per thread (TRK_CF_CV2: the two-chunk tile, for A/B runs of the headline's set)
            switch (n_filters) {
                case 1: TRK_V4_PICK(1); break;
                case 2: TRK_V4_PICK(2); break;
                case 3: TRK_V4_PICK(3); break;
                case 4: TRK_V4_PICK(4); break;
                case 5: TRK_V4_PICK(5); break;
                default: TRK_V4_PICK(6); break;
            }
#undef TRK_V4_PICK
            if (compact && n_filters == 3 && tflt == 1 && alias == 3 && !ratio && trk_opt("TRK_CF_CV2")) {
                k4 = k_call_filter_v4<3, 1, true, false, 3, 2, 2>;     // (A/B: the two-chunk tile, the headline's filter set)
                cv1 = false;
            }
            // LDS: the block's delta table + class LUT + locus info, and one queue per wave -- within 32 KiB, so that
            // five workgroups fit a CU
            if (delta) {
                // (a workgroup's LDS stays well below 160 KiB / 5: at 32 360 bytes the occupancy query still says five
                // workgroups per CU and four are resident -- a persistent launch then runs a second round)
                const int max_lpb = max_lpb4;
                if (lpb > max_lpb) lpb = max_lpb;
                if (lpb > 255) lpb = 255;
                if (lpb < 1) lpb = 1;
            }
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k4, CF_THREADS, (size_t)lpb * per_locus4 + qbytes + 8) != hipSuccess || occ < 1)
                occ = 4;
            // (the compact build: two chunks per thread, i.e. column tiles of 2048 samples; blocks of whole ring turns)
            const int gxl = (compact && !cv1) ? (S + 2 * CF_THREADS * CF_V - 1) / (2 * CF_THREADS * CF_V) : gx;
            const CfLaunch cl = cf_geometry(L, gxl, lpb, n_cu, occ, compact ? V4_PD : 1);
            lpb = cl.lpb;
            const size_t lds4 = delta ? (((size_t)lpb * per_locus4 + 7) & ~(size_t)7) + qbytes : 0;
            v.loci_per_block = lpb;
            v.geom = cl.geom;
            gy = cl.geom.n_ranges;
            v.part16 = nullptr;
            v.part64 = nullptr;
            if ((long)cl.geom.walk * lpb < 65536 && !trk_opt("TRK_CF_ATOMICS")) {
                const size_t b16 = (((size_t)gy * (2 + n_filters) * S * sizeof(uint16_t)) + 255) & ~(size_t)255;
                const size_t b64 = (size_t)gy * S * sizeof(unsigned long long);
                if (void* ws = scratch.get(scratch.user, b16 + b64)) {
                    v.part16 = static_cast<uint16_t*>(ws);
                    v.part64 = reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + b16);
                }
            }
            if (trk_opt("TRK_CF_VERBOSE"))
                fprintf(stderr, "k_call_filter_v4<%d,%d,%d,%d,%d,%d>: L %d gx %d lpb %d walk %d ranges %d map %d grid %u x %u, lds %zu B, "
                                "occupancy %d WG/CU\n", n_filters, tflt, (int)delta, (int)ratio, alias, compact ? 2 : 0, L, gxl, lpb, cl.geom.walk,
                        cl.geom.n_ranges, cl.geom.map, cl.grid.x, cl.grid.y, lds4, occ);
            hipLaunchKernelGGL(k4, cl.grid, dim3(CF_THREADS), lds4, stream, v);
            if (v.part16) {
                hipError_t e1 = hipGetLastError();
                if (e1 != hipSuccess) return e1;
                if (scratch.next_kernel) scratch.next_kernel(scratch.user);
                CfrBits fb = {};
                for (int k = 0; k < n_filters; ++k) fb.bit[k] = v.f[k].bit;
                hipLaunchKernelGGL(k_cf_reduce, dim3((S / 4 + CFR_QPB - 1) / CFR_QPB), dim3(CFR_NSL * CFR_QPB), 0, stream,
                                   v.part16, v.part64, gy, S, 2 + n_filters, fb, out);
            }
            return hipGetLastError();
        }
    }
    // ---- dumpSTR's closed GangSTR list (k_call_filter_gs): every filter one of the nine roles, the depth plane the
    // DP operand, multi-column planes all planar or all interleaved.  TRK_CF_NOGS=1: the interpreter kernel. ----
    if (vec && !b.locus_ploidy && n_filters >= 1 && n_filters <= GS_NF && !trk_opt("TRK_CF_NOGS") &&
        !trk_opt("TRK_CF_GENERIC") && (!out.delta_allele_count || (b.max_alleles > 0 && b.max_alleles <= 120))) {
        GsArgs g = {};
        bool ok = true;
        int qexp_pl = -1, rc_pl = -1, cn_pl = -1, ci_pl = -1, q_pl = -1;
        auto is_i32 = [&](int p) { return (planes[p].dtype & 0xff) == TRK_DT_I32; };
        auto is_f32 = [&](int p) { return (planes[p].dtype & 0xff) == TRK_DT_F32; };
        auto same = [&](int& slot, int p) { if (slot < 0) slot = p; return slot == p; };
        auto int_thr = [&](double thr, bool gt_op, int32_t& o) {
            const double t = gt_op ? floor(thr) : ceil(thr);   // (double)v < thr <=> v < ceil(thr); > <=> v > floor(thr)
            if (!(thr == thr) || !(t > -2147483647.0 && t < 2147483647.0)) return false;
            o = (int32_t)t;
            return true;
        };
        for (int k = 0; k < n_filters && ok; ++k) {
            const trk_call_filter& f = filters[k];
            int role = -1;
            switch (f.op) {
                case TRK_F_LT:
                    if (f.plane_a == dp_plane && is_i32(f.plane_a) && planes[f.plane_a].ncol == 1) {
                        role = 0;
                        ok = int_thr(f.thr, false, g.f[0].ithr);
                    } else if (is_f32(f.plane_a) && planes[f.plane_a].ncol == 1 && f.thr == f.thr) {
                        role = 2;
                        ok = same(q_pl, f.plane_a);
                        g.f[2].fthr = (float)f.thr;
                    }
                    break;
                case TRK_F_GT:
                    if (f.plane_a == dp_plane && is_i32(f.plane_a) && planes[f.plane_a].ncol == 1) {
                        role = 1;
                        ok = int_thr(f.thr, true, g.f[1].ithr);
                    }
                    break;
                case TRK_F_CALLED_LT:
                    if (is_f32(f.plane_a) && planes[f.plane_a].ncol == 3 && (f.col_a == 1 || f.col_a == 2) && f.thr == f.thr) {
                        role = f.col_a == 1 ? 3 : 4;
                        ok = same(qexp_pl, f.plane_a);
                        g.f[role].fthr = (float)f.thr;
                    }
                    break;
                case TRK_F_CALLED_SUM_LT:
                    if (is_f32(f.plane_a) && planes[f.plane_a].ncol == 3 && f.thr == f.thr &&
                        ((f.col_a == 1 && f.col_a2 == 2) || (f.col_a == 2 && f.col_a2 == 1))) {
                        role = 5;
                        ok = same(qexp_pl, f.plane_a);
                        g.f[5].fthr = (float)f.thr;
                    }
                    break;
                case TRK_F_CALLED_EQ:
                    if (is_i32(f.plane_a) && planes[f.plane_a].ncol == 4 && f.col_a == 1 && f.plane_b == dp_plane &&
                        dp_plane >= 0 && is_i32(dp_plane) && planes[dp_plane].ncol == 1 && f.col_b == 0) {
                        role = 6;
                        ok = same(rc_pl, f.plane_a);
                    }
                    break;
                case TRK_F_CALLED_SUM_EQ:
                    if (is_i32(f.plane_a) && planes[f.plane_a].ncol == 4 && f.plane_b == dp_plane && dp_plane >= 0 &&
                        is_i32(dp_plane) && planes[dp_plane].ncol == 1 && f.col_b == 0 &&
                        ((f.col_a == 1 && f.col_a2 == 3) || (f.col_a == 3 && f.col_a2 == 1))) {
                        role = 7;
                        ok = same(rc_pl, f.plane_a);
                    }
                    break;
                case TRK_F_CALLED_OUTSIDE_CI:
                    if (is_i32(f.plane_a) && planes[f.plane_a].ncol == 2 && is_i32(f.plane_b) && planes[f.plane_b].ncol == 4) {
                        role = 8;
                        ok = same(cn_pl, f.plane_a) && same(ci_pl, f.plane_b);
                    }
                    break;
                default:
                    break;
            }
            if (role < 0 || g.f[role].on) ok = false;    // not one of the nine, or a role twice
            if (ok) {
                g.f[role].on = 1;
                g.f[role].bit = k;
            }
        }
        for (int r = 0; r < GS_NF; ++r)
            if (!g.f[r].on) g.f[r].bit = -1;             // (its counter row is all zero and lands nowhere)
        if (ok && dp_plane >= 0 && !(is_i32(dp_plane) && planes[dp_plane].ncol == 1)) ok = false;
        // multi-column planes: all planar or all interleaved
        int n_planar = 0, n_inter = 0;
        for (int p : {qexp_pl, rc_pl, cn_pl, ci_pl})
            if (p >= 0) ((planes[p].dtype & TRK_DT_PLANAR) ? n_planar : n_inter)++;
        if (ok && n_planar && n_inter) ok = false;
        if (ok) {
            const bool planar = n_planar > 0 || n_inter == 0;
            const size_t colb = (size_t)L * S * 4;        // bytes of one planar column
            auto colp = [&](int p, int c) { return static_cast<const void*>(static_cast<const char*>(planes[p].data) + colb * c); };
            g.b = b;
            g.has_dp = dp_plane >= 0;
            g.use_qexp = qexp_pl >= 0;
            g.use_rc = rc_pl >= 0;
            g.use_ci = cn_pl >= 0;
            g.p[0] = dp_plane >= 0 ? planes[dp_plane].data : nullptr;
            g.p[1] = q_pl >= 0 ? planes[q_pl].data : nullptr;
            if (planar) {
                if (qexp_pl >= 0) { g.p[2] = colp(qexp_pl, 1); g.p[3] = colp(qexp_pl, 2); }
                if (rc_pl >= 0) { g.p[4] = colp(rc_pl, 1); g.p[5] = colp(rc_pl, 3); }
                if (cn_pl >= 0) {
                    g.p[6] = colp(cn_pl, 0); g.p[7] = colp(cn_pl, 1);
                    for (int c = 0; c < 4; ++c) g.p[8 + c] = colp(ci_pl, c);
                }
            } else {
                g.p[2] = qexp_pl >= 0 ? planes[qexp_pl].data : nullptr;
                g.p[3] = rc_pl >= 0 ? planes[rc_pl].data : nullptr;
                g.p[4] = cn_pl >= 0 ? planes[cn_pl].data : nullptr;
                g.p[5] = ci_pl >= 0 ? planes[ci_pl].data : nullptr;
            }
            const bool delta = out.delta_allele_count != nullptr;
            g.out = a.out;
            g.delta_nal = delta ? b.max_alleles : 0;
            // the whole list in dumpSTR's own order: the instantiation with the compile-time role set
            bool canonical = dp_plane >= 0;
            for (int r = 0; r < GS_NF; ++r) canonical = canonical && g.f[r].on && g.f[r].bit == r;
            // (interleaved planes: the build with run-time flags is the faster one -- 3.44 against 3.6-3.8 ms at 50k x
            // 5k on one box, profiles/r03_notes.md -- so the compile-time set serves the planar layout only)
            canonical = canonical && planar && !trk_opt("TRK_GS_RUNTIME_FLAGS");
            void (*kg)(GsArgs) =
                canonical ? (delta ? k_call_filter_gs<true, true, 0x1ff> : k_call_filter_gs<true, false, 0x1ff>)
                          : (planar ? (delta ? k_call_filter_gs<true, true, -1> : k_call_filter_gs<true, false, -1>)
                                    : (delta ? k_call_filter_gs<false, true, -1> : k_call_filter_gs<false, false, -1>));
            // geometry as for k_call_filter_v4 (cf_geometry), the delta table within 32 KiB
            const size_t per_locus = delta ? ((size_t)b.max_alleles + V2_EXTRA + CF_LINFO) * sizeof(uint32_t) : 0;
            const size_t gq = delta ? (size_t)(CF_THREADS / WAVE) * V4_QCAP * sizeof(uint2) + 8 : 0;   // the waves' queues
            int lpb2 = lpb;
            if (delta) lpb2 = std::max(1, std::min<int>({lpb2, 255, (int)((30 * 1024 - gq) / per_locus)}));
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kg, CF_THREADS, (size_t)lpb2 * per_locus + gq) != hipSuccess || occ < 1)
                occ = 3;
            const CfLaunch cl = cf_geometry(L, gx, lpb2, n_cu, occ);
            lpb2 = cl.lpb;
            const int gy2 = cl.geom.n_ranges;
            g.loci_per_block = lpb2;
            g.geom = cl.geom;
            const size_t b16 = (((size_t)gy2 * (2 + GS_NF) * S * sizeof(uint16_t)) + 255) & ~(size_t)255;
            const size_t b64 = (size_t)gy2 * S * sizeof(unsigned long long);
            void* ws = (long)cl.geom.walk * lpb2 < 65536 ? scratch.get(scratch.user, b16 + b64) : nullptr;
            if (ws) {
                g.part16 = static_cast<uint16_t*>(ws);
                g.part64 = reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + b16);
                hipLaunchKernelGGL(kg, cl.grid, dim3(CF_THREADS), delta ? ((((size_t)lpb2 * per_locus) + 7) & ~(size_t)7) + gq : 0, stream, g);
                hipError_t e1 = hipGetLastError();
                if (e1 != hipSuccess) return e1;
                if (scratch.next_kernel) scratch.next_kernel(scratch.user);
                CfrBits fb = {};
                for (int r = 0; r < GS_NF; ++r) fb.bit[r] = g.f[r].bit;
                hipLaunchKernelGGL(k_cf_reduce, dim3((S / 4 + CFR_QPB - 1) / CFR_QPB), dim3(CFR_NSL * CFR_QPB), 0, stream,
                                   g.part16, g.part64, gy2, S, 2 + GS_NF, fb, out);
                return hipGetLastError();
            }
        }
    }
    if (vec) {
        if (out.delta_allele_count && !lds_delta) {
            // no room for an LDS table (huge allele sets): per-call evaluation with global atomics
            hipLaunchKernelGGL(k_call_filter<false>, dim3(gx, gy), dim3(CF_THREADS), lds, stream, a);
            return hipGetLastError();
        }
        // knobs for tools/perf_sweep.py: TRK_CF_ROUNDS / TRK_CF_WPC workgroups per CU, TRK_CF_DELTA_KB table budget
        int rounds = 1, wpc = 0, delta_kb = 8;
        if (const char* e = trk_opt("TRK_CF_ROUNDS")) rounds = atoi(e) > 0 ? atoi(e) : rounds;
        if (const char* e = trk_opt("TRK_CF_WPC")) wpc = atoi(e);
        if (const char* e = trk_opt("TRK_CF_DELTA_KB")) delta_kb = atoi(e) > 0 ? atoi(e) : delta_kb;
        // load slots / element offsets of the register-vector path (CallArgs: slot_*, src_off, src_stride)
        for (int q = 0; q < 16; ++q) {
            a.slot_ptr[q] = b.gt;
            a.slot_mult[q] = 1;
            a.slot_chunk[q] = 0;
            a.src_off[q] = (int8_t)(4 * q);
            a.src_stride[q] = 1;
        }
        for (int q = 0; q < a.n_src && q < 16; ++q)
            if (a.src_ptr[q]) a.slot_ptr[q] = a.src_ptr[q];
        for (int g = 0; g < a.grp_n; ++g)
            for (int c = 0; c < a.grp_k[g]; ++c) {
                const int q = a.grp_base[g] + c;
                a.slot_ptr[q] = a.grp_ptr[g];
                a.slot_mult[q] = a.grp_k[g];
                a.slot_chunk[q] = (int8_t)c;
                a.src_off[q] = (int8_t)(4 * a.grp_base[g] + c);
                a.src_stride[q] = a.grp_k[g];
            }
        const bool allreg = a.reg_filter_mask == (n_filters >= 32 ? ~0u : (1u << n_filters) - 1u) &&
                            (dp_plane < 0 || a.dp_src >= 0) && !trk_opt("TRK_CF_NOALLREG");
        void (*kfn)(CallArgs) = nullptr;
        // TRK_CF_NOREGVEC=1: the select-chain form also for planar sources (A/B timing)
        const bool regvec = allreg && !trk_opt("TRK_CF_NOREGVEC");
#define TRK_FAST(NS)                                                                                          \
    kfn = regvec ? (a.grp_n == 0 ? k_call_filter_fast<NS, true, 1> : k_call_filter_fast<NS, true, 2>)         \
                 : allreg ? k_call_filter_fast<NS, true, 0> : k_call_filter_fast<NS, false, 0>
        if (a.n_src <= 4) { TRK_FAST(4); }
        else if (a.n_src <= 8) { TRK_FAST(8); }
        else if (a.n_src <= 12) { TRK_FAST(12); }
        else { TRK_FAST(16); }
#undef TRK_FAST
        // geometry: every workgroup owns one contiguous locus range (<= 61440 loci: the 16-bit per-thread
        // counters) and all of them are resident at once -- the grid is the kernel's occupancy x CUs, so no
        // partial second round of workgroups trails the first -- and walks it in sub-blocks sized by the
        // LDS delta table.
        const size_t lds_counters = (size_t)n_filters * CF_THREADS * 2 * sizeof(uint32_t);
        const size_t per_locus = ((size_t)a.delta_stride + b.max_alleles + CF_LINFO) * sizeof(int32_t);
        int max_sub = lds_delta ? (int)((size_t)delta_kb * 1024 / per_locus) : 0;
        if (lds_delta && max_sub < 8) max_sub = 8;
        if (wpc <= 0) {
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, CF_THREADS,
                                                             lds_counters + (size_t)max_sub * per_locus) != hipSuccess ||
                occ < 1)
                occ = 3;
            wpc = occ * rounds;
        }
        int gyf = (n_cu * wpc) / gx;   // round down: never more workgroups than slots
        if (gyf > L) gyf = L;
        if (gyf < 1) gyf = 1;
        int lpw = (L + gyf - 1) / gyf;
        if (lpw > 61440) lpw = 61440;
        gyf = (L + lpw - 1) / lpw;
        int sub = lpw;
        lds = lds_counters;
        if (lds_delta) {
            const int nsub = (lpw + max_sub - 1) / max_sub;
            sub = (lpw + nsub - 1) / nsub;
            lds += (size_t)sub * per_locus;
        }
        a.loci_per_wg = lpw;
        a.loci_per_block = sub;
        hipLaunchKernelGGL(kfn, dim3(gx, gyf), dim3(CF_THREADS), lds, stream, a);
    } else {
        hipLaunchKernelGGL(k_call_filter<false>, dim3(gx, gy), dim3(CF_THREADS), lds, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_locus_filter(int L, const int32_t* locus_int, const double* locus_f64,
                               const trk_locus_filter_spec& spec, uint32_t* bits, int64_t* counters,
                               hipStream_t stream) {
    if (L == 0) return hipSuccess;
    hipLaunchKernelGGL(k_locus_filter, dim3((L + 255) / 256), dim3(256), 0, stream, L, locus_int, locus_f64, spec,
                       bits, reinterpret_cast<unsigned long long*>(counters));
    return hipGetLastError();
}

hipError_t launch_synth(const trk_synth_spec& sp, int16_t* gt, int32_t* dp, float* q, int32_t* dstutter,
                        int32_t* dflank, int n_cu, hipStream_t stream) {
    int64_t n = (int64_t)sp.n_loci * sp.n_samples;
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)n_cu * 32) blocks = (int64_t)n_cu * 32;
    hipLaunchKernelGGL(k_synth, dim3((int)blocks), dim3(256), 0, stream, sp, gt, dp, q, dstutter, dflank);
    return hipGetLastError();
}

hipError_t launch_synth_gangstr(const trk_synth_spec& sp, const int16_t* gt, const int32_t* dp,
                                const int32_t* allele_repcn, float* qexp, int32_t* repcn, int32_t* rc,
                                int32_t* repci, int n_cu, hipStream_t stream) {
    int64_t n = (int64_t)sp.n_loci * sp.n_samples;
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)n_cu * 32) blocks = (int64_t)n_cu * 32;
    hipLaunchKernelGGL(k_synth_gangstr, dim3((int)blocks), dim3(256), 0, stream, sp, gt, dp, allele_repcn, qexp,
                       repcn, rc, repci);
    return hipGetLastError();
}

hipError_t launch_permute_columns(const int16_t* src, int16_t* dst, const int32_t* col, int64_t n_loci, int n_src,
                                  int n_dst, int ploidy, int n_cu, hipStream_t stream) {
    if (n_loci <= 0 || n_dst <= 0) return hipSuccess;
    int64_t wgs = (ploidy == 2 && (n_dst & 3) == 0) ? n_loci : (n_loci * n_dst + 255) / 256;
    const int64_t cap = (int64_t)n_cu * 32;
    if (wgs > cap) wgs = cap;
    // rows of up to 16k diploid samples are staged in LDS (64 KiB); longer ones are gathered from global memory
    const bool lds_ok = ploidy == 2 && (n_dst & 3) == 0 && (size_t)n_src * 4 <= 64 * 1024 && !trk_opt("TRK_PERMUTE_NOLDS");
    hipLaunchKernelGGL(k_permute_columns, dim3((unsigned)wgs), dim3(256), lds_ok ? (size_t)n_src * 4 : 0, stream, src, dst, col,
                       n_loci, n_src, n_dst, ploidy, lds_ok ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_stream_probe(const void* const* in, int n_in, void* const* out, int n_out, int64_t n_loci,
                               int64_t n_samples, int n_cu, hipStream_t stream) {
    if (!((n_in == 3 && n_out == 2) || (n_in == 0 && n_out == 2)) || n_samples % 4) return hipErrorInvalidValue;
    ProbeArgs a = {};
    for (int k = 0; k < n_in; ++k) a.in[k] = static_cast<const u32x4*>(in[k]);
    for (int k = 0; k < n_out; ++k) a.out[k] = static_cast<u32x4*>(out[k]);
    const int S4 = (int)(n_samples / 4), L = (int)n_loci;
    const int gx = (S4 + 255) / 256;
    // the call-filter pass's own geometry (cf_geometry: persistent ranges, XCD-aware map; the same TRK_CF_* knobs),
    // five workgroups per CU admitted (capped here by 30 KiB of dynamic LDS), blocks of 80 loci as the kernel's
    const CfLaunch cl = cf_geometry(L, gx, 80, n_cu, 5);
    const int lpb = cl.lpb;
    if (n_in == 0) hipLaunchKernelGGL((k_stream_probe<0, 2>), cl.grid, dim3(256), 30 * 1024, stream, a, L, S4, lpb, cl.geom);
    else hipLaunchKernelGGL((k_stream_probe<3, 2>), cl.grid, dim3(256), 30 * 1024, stream, a, L, S4, lpb, cl.geom);
    return hipGetLastError();
}

hipError_t launch_pad_rows(const void* src, void* dst, int64_t n_rows, int row_words, int pad_words, uint32_t fill,
                           int n_cu, hipStream_t stream) {
    const int64_t n = n_rows * ((int64_t)row_words + pad_words);
    const int64_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < (int64_t)n_cu * 32 ? want : (int64_t)n_cu * 32);
    hipLaunchKernelGGL(k_pad_rows, dim3(grid ? grid : 1), dim3(256), 0, stream, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(dst), n_rows, row_words, pad_words, fill);
    return hipGetLastError();
}

hipError_t launch_planarize(const void* src, void* dst, int64_t n_cells, int ncol, hipStream_t stream) {
    int64_t blocks = (n_cells + 255) / 256;
    if (blocks > 65535 * 16) blocks = 65535 * 16;
    hipLaunchKernelGGL(k_planarize, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(dst), n_cells, ncol);
    return hipGetLastError();
}

}  // namespace trk
