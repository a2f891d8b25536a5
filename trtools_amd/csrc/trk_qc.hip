// trk_qc.hip -- qcSTR's reductions over a batch for gfx950 (second half).
//
// Reference loop (one Python iteration per record, trtools/qcSTR/qcSTR.py:529-561):
//     idx_gts = trrecord.GetGenotypeIndicies()[sample_index, :-1]
//     calls   = ~np.all(idx_gts == -1, axis=1)            # a call unless EVERY haplotype index is -1
//     sample_calls += calls ; chrom_calls[chrom] += np.sum(calls)
//     quality_scores = trrecord.GetQualityScores()[sample_index, :] ; quality_scores[~calls] = nan
//     not --quality-ignore-no-call:  nan -> 0 ; per_sample_total += q ; per_locus.append(np.mean(q))
//     --quality-ignore-no-call:      only the non-nan entries enter the sums and the mean
// Here: ONE pass over the genotype tensor and the quality plane (8 B per call) gives, per sample, the number of
// calls, the quality sum and the number of entries summed, and the same three per locus; the means are divisions the
// caller does (qcSTR.py:619-621, 554-556).
//
//   k_qc_scan<V>    column-owner tiling as in the call-filter kernel: a thread owns V consecutive samples (4 when the
//                   batch is diploid with S % 4 == 0: one 16-byte load per plane, else 1 and any ploidy) and walks the
//                   loci of its workgroup (blockIdx.y); the per-sample sums live in registers and are written once, as
//                   this workgroup's partial row.  Per locus a wave needs the sum over its lanes: the two counts are
//                   popcounts of lane masks (scalar), the float64 quality sum of eight loci at a time goes through LDS
//                   (column sums by eight lanes each + three shuffle steps) -- one partial row per wave.
//   k_qc_scan4      round 6: the diploid S % 4 == 0 form with its loads batched, no branches around them, one scalar load
//                   of a group's ploidies, an XCD-aware 1-D launch and whole-line stores of the per-locus partials.
//   k_qc_finish     partial rows added in a fixed order (four slices of rows, each in row order: deterministic float64
//                   sums) into the int64 / float64 outputs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/trk.h"
#include "../../include/trk_test.h"
#include "trk_internal.h"

namespace {

constexpr int WAVE = 64;
constexpr int QC_THREADS = 256;
constexpr int QC_WAVES = QC_THREADS / WAVE;
constexpr int QC_G = 8;  // loci per reduction group
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct QcArgs {
    trk_batch b;
    const uint8_t* sample_in;  // [S] or null
    const float* quality;      // [L*S] or null
    int ignore_no_call;
    int loci_per_wg;
    int s_pad;                 // samples rounded up to the workgroup tile
    int n_rows;                // gridDim.x * QC_WAVES: per-locus partial rows
    int gx, gy;                // column tiles, locus ranges (k_qc_scan4's 1-D launch derives its place from them)
    uint32_t* p_calls;         // [gridDim.y][s_pad]
    uint32_t* p_qn;            // [gridDim.y][s_pad]
    double* p_qsum;            // [gridDim.y][s_pad]
    uint32_t* l_calls;         // [n_rows][L]
    uint32_t* l_qn;            // [n_rows][L]
    double* l_qsum;            // [n_rows][L]
};

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// x += this lane's bit of the wave-wide mask (the mask as the carry of an add-with-carry)
__device__ __forceinline__ void add_mask(uint32_t& x, uint64_t mask) {
    asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(x) : "s"(mask) : "vcc");
}

template <int V>
__global__ __launch_bounds__(QC_THREADS) void k_qc_scan(const QcArgs a) {
    __shared__ double red_all[QC_WAVES][QC_G * WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid >> 6;
    double* red = red_all[wid];
    const int S = a.b.n_samples, L = a.b.n_loci, P = a.b.ploidy;
    const int64_t s0 = ((int64_t)blockIdx.x * QC_THREADS + tid) * V;
    const bool live = s0 < S;   // V == 4: S % 4 == 0, a thread's samples are all in range or all out
    const bool has_q = a.quality != nullptr;
    const bool ignore = a.ignore_no_call != 0;
    const int row = blockIdx.x * QC_WAVES + wid;
    uint64_t inm[V];            // lanes whose sample j is in the sample set
#pragma unroll
    for (int j = 0; j < V; ++j) inm[j] = __ballot(live && (!a.sample_in || a.sample_in[s0 + j] != 0));
    uint32_t calls[V], qn[V];
    double qsum[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        calls[j] = 0;
        qn[j] = 0;
        qsum[j] = 0.0;
    }
    const int l_begin = blockIdx.y * a.loci_per_wg;
    const int l_end = min(L, l_begin + a.loci_per_wg);
    for (int lb = l_begin; lb < l_end; lb += QC_G) {
        double lq[QC_G];
        uint32_t lc[QC_G], ln[QC_G];
#pragma unroll
        for (int v = 0; v < QC_G; ++v) {
            lq[v] = 0.0;
            lc[v] = ln[v] = 0;
            const int l = lb + v;
            if (l >= l_end) continue;   // uniform
            const int pl = a.b.locus_ploidy ? min((int)a.b.locus_ploidy[l], P) : P;
            uint64_t callm[V];
            float q[V];
            if (V == 4) {
                u32x4 w = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                f32x4 qv = {0.f, 0.f, 0.f, 0.f};
                if (live) {
                    const int64_t c4 = ((int64_t)l * S + s0) >> 2;
                    w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.b.gt) + c4);
                    if (has_q) qv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.quality) + c4);
                }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    // a call unless every haplotype index of the record is -1 (qcSTR.py:533-535)
                    const uint64_t m = pl > 1 ? __ballot(w[j] != 0xffffffffu) : __ballot((w[j] & 0xffffu) != 0xffffu);
                    callm[j] = m & inm[j];
                    q[j] = qv[j];
                }
            } else {
                bool call = false;
                float qv = 0.f;
                if (live) {
                    const int16_t* g = a.b.gt + ((int64_t)l * S + s0) * P;
                    for (int p = 0; p < pl; ++p) call |= g[p] != -1;
                    if (has_q) qv = a.quality[(int64_t)l * S + s0];
                }
                callm[0] = __ballot(call) & inm[0];
                q[0] = qv;
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                add_mask(calls[j], callm[j]);
                lc[v] += (uint32_t)__popcll(callm[j]);
                if (has_q) {
                    // no-calls become nan (qcSTR.py:540); then either nan -> 0 and every selected sample counts
                    // (543), or only the non-nan entries do (545, 551, 556)
                    const uint64_t valid = callm[j] & __ballot(q[j] == q[j]);
                    const uint64_t counted = ignore ? valid : inm[j];
                    const double qz = __builtin_amdgcn_inverse_ballot_w64(valid) ? (double)q[j] : 0.0;
                    qsum[j] += qz;
                    add_mask(qn[j], counted);
                    lq[v] += qz;
                    ln[v] += (uint32_t)__popcll(counted);
                }
            }
        }
        // per-locus sums of this wave: counts are uniform already, the quality sums go through LDS
        uint32_t my_lc = 0, my_ln = 0;
        const int vi = lane >> 3, p8 = lane & 7;
#pragma unroll
        for (int v = 0; v < QC_G; ++v) {
            my_lc = vi == v ? lc[v] : my_lc;
            my_ln = vi == v ? ln[v] : my_ln;
        }
        double sum = 0.0;
        if (has_q) {
#pragma unroll
            for (int v = 0; v < QC_G; ++v) red[v * WAVE + lane] = lq[v];
            wave_fence();
            const double2* rp = reinterpret_cast<const double2*>(red + vi * WAVE + p8 * 8);
            const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
            sum = ((r0.x + r0.y) + (r1.x + r1.y)) + ((r2.x + r2.y) + (r3.x + r3.y));
            sum += __shfl_xor(sum, 1, WAVE);
            sum += __shfl_xor(sum, 2, WAVE);
            sum += __shfl_xor(sum, 4, WAVE);
            wave_fence();
        }
        if (p8 == 0 && lb + vi < l_end) {
            const size_t at = (size_t)row * L + lb + vi;
            a.l_calls[at] = my_lc;
            if (has_q) {
                a.l_qn[at] = my_ln;
                a.l_qsum[at] = sum;
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const size_t at = (size_t)blockIdx.y * a.s_pad + s0 + j;
            a.p_calls[at] = calls[j];
            if (has_q) {
                a.p_qn[at] = qn[j];
                a.p_qsum[at] = qsum[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_qc_scan4<HAS_Q, IGNORE> (round 6): the diploid, S % 4 == 0 form of k_qc_scan with the memory system in mind.
// k_qc_scan<4> asked for one locus at a time -- two 16-byte loads, `s_waitcnt vmcnt(0)`, and before them a VECTOR byte
// load of locus_ploidy[l] with a wait of its own (gfx9 has no scalar byte load) -- inside run-time branches on
// quality / ignore / ploidy: two kilobytes in flight per wave, 0.56-0.61 of the HBM peak at 8 B per call.  Here:
//   * the flags are template arguments, the tail lanes read chunk 0 and are masked out (no branch around a load);
//   * the ploidies of a group of eight loci are ONE 8-byte scalar load (groups start on multiples of eight);
//   * the sixteen loads of a group are issued back to back before the first is used: 16 KB in flight per wave.
// Same sums in the same order as k_qc_scan<4> (per sample: locus order; per locus: eight lanes' column sums, then
// three shuffle steps), so the outputs are bit for bit the old kernel's.
template <bool HAS_Q, bool IGNORE>
__global__ __launch_bounds__(QC_THREADS) void k_qc_scan4(const QcArgs a) {
    __shared__ double red_all[QC_WAVES][QC_G * WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid >> 6;
    double* red = red_all[wid];
    const int S = a.b.n_samples, L = a.b.n_loci;
    // 1-D launch, XCD-aware as the call-filter kernels' (CfGeom map 2, trk_kernels.hip): block b runs on XCD b % 8, so
    // range = b % 8 + 8 * ((b / 8) / gx), tile = (b / 8) % gx puts the gx column tiles of a range -- whole rows, every
    // 4 KB page of them -- on ONE XCD (one L2, one set of translations) and gives each XCD every eighth range
    const int slot = (int)blockIdx.x >> 3;
    const int bx = slot % a.gx, by = ((int)blockIdx.x & 7) + 8 * (slot / a.gx);
    if (by >= a.gy) return;
    const int64_t s0 = ((int64_t)bx * QC_THREADS + tid) * 4;
    const bool live = s0 < S;
    const int row = bx * QC_WAVES + wid;
    uint64_t inm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inm[j] = __ballot(live && (!a.sample_in || a.sample_in[s0 + j] != 0));
    uint32_t calls[4] = {0, 0, 0, 0}, qn[4] = {0, 0, 0, 0};
    double qsum[4] = {0.0, 0.0, 0.0, 0.0};
    const int l_begin = by * a.loci_per_wg;                  // (a multiple of 4 QC_G: qc_geometry)
    const int l_end = min(L, l_begin + a.loci_per_wg);
    const int64_t col4 = live ? (s0 >> 2) : 0;               // this lane's chunk of a row (a dead lane: chunk 0, masked out)
    const int64_t row4 = (int64_t)(S >> 2);
    const u32x4* gt4 = reinterpret_cast<const u32x4*>(a.b.gt);
    const f32x4* q4 = reinterpret_cast<const f32x4*>(a.quality);
    const int vi = lane >> 3, p8 = lane & 7;
    // the ploidies of a group: eight bytes in one load where the array has them, asked for one group AHEAD, behind the
    // sixteen plane loads of the group before (the memory counter is in order: when those have arrived, so has this)
    uint32_t keep_lc = 0, keep_ln = 0;
    double keep_sum = 0.0;
    auto wide_at = [&](int lb) {
        return a.b.locus_ploidy && lb + QC_G <= L && (reinterpret_cast<uintptr_t>(a.b.locus_ploidy + lb) & 7) == 0;
    };
    uint64_t pp_next = 0x0202020202020202ull;
    if (l_begin < l_end && wide_at(l_begin)) pp_next = *reinterpret_cast<const uint64_t*>(a.b.locus_ploidy + l_begin);
    for (int lb = l_begin; lb < l_end; lb += QC_G) {
        const int ng = min(QC_G, l_end - lb);                // uniform
        const bool wide = wide_at(lb);
        uint64_t pp = pp_next;
        // a call unless every haplotype index of the record is -1 (qcSTR.py:533-535): the haplotypes that count are the low
        // half of a haploid record's word, the whole word otherwise -- one mask, no branch.  The per-locus counts are wave-
        // uniform: each goes straight into the lane that will store it (lane >> 3 == v) -- sixteen live scalars spilled
        // the kernel's scalar registers into vector lanes.  The group's loads go out in two halves of four loci (eight
        // loads back to back: 8 KB per wave in flight; all sixteen at once cost 178 registers or scratch).
        double lq[QC_G];
        uint32_t my_lc = 0, my_ln = 0;
        constexpr int QC_H = QC_G / 2;
        if (__builtin_expect(a.b.locus_ploidy && !wide, 0)) {     // (a range's last group, a view that starts off an 8-byte boundary)
            pp = 0;
            for (int v = 0; v < ng; ++v) pp |= (uint64_t)a.b.locus_ploidy[lb + v] << (8 * v);
        }
        auto group = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;      // all eight loci of the group are in the range
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 w[QC_H];
                f32x4 qv[QC_H];
#pragma unroll
                for (int x = 0; x < QC_H; ++x) {
                    const int lx = lb + h * QC_H + x;
                    const int64_t c4 = (int64_t)(FULL ? lx : min(lx, l_end - 1)) * row4 + col4;   // (beyond the range: the last row again, unused)
                    w[x] = __builtin_nontemporal_load(gt4 + c4);
                    if (HAS_Q) qv[x] = __builtin_nontemporal_load(q4 + c4);
                }
                if (h == 0) {
                    pp_next = 0x0202020202020202ull;
                    if (lb + QC_G < l_end && wide_at(lb + QC_G)) pp_next = *reinterpret_cast<const uint64_t*>(a.b.locus_ploidy + lb + QC_G);
                }
#pragma unroll
                for (int x = 0; x < QC_H; ++x) {
                    const int v = h * QC_H + x;
                    lq[v] = 0.0;
                    if (!FULL && v >= ng) continue;                                                  // uniform
                    const uint32_t sel = ((pp >> (8 * v)) & 0xffu) <= 1u ? 0xffffu : 0xffffffffu;    // uniform
                    uint32_t lcv = 0, lnv = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint64_t callm = __ballot((w[x][j] & sel) != sel) & inm[j];
                        add_mask(calls[j], callm);
                        lcv += (uint32_t)__popcll(callm);
                        if (HAS_Q) {
                            // no-calls become nan (qcSTR.py:540); then either nan -> 0 and every selected sample counts (543),
                            // or only the non-nan entries do (545, 551, 556)
                            const float qf = qv[x][j];
                            const uint64_t valid = callm & __ballot(qf == qf);
                            const uint64_t counted = IGNORE ? valid : inm[j];
                            const double qz = (double)(__builtin_amdgcn_inverse_ballot_w64(valid) ? qf : 0.0f);    // (selected as a float: one move)
                            qsum[j] += qz;
                            add_mask(qn[j], counted);
                            lq[v] += qz;
                            lnv += (uint32_t)__popcll(counted);
                        }
                    }
                    my_lc = vi == v ? lcv : my_lc;
                    my_ln = vi == v ? lnv : my_ln;
                }
            }
        };
        if (__builtin_expect(ng == QC_G, 1)) group(std::true_type{});
        else group(std::false_type{});
        double sum = 0.0;
        if (HAS_Q) {
#pragma unroll
            for (int v = 0; v < QC_G; ++v) red[v * WAVE + lane] = lq[v];
            wave_fence();
            const double2* rp = reinterpret_cast<const double2*>(red + vi * WAVE + p8 * 8);
            const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
            sum = ((r0.x + r0.y) + (r1.x + r1.y)) + ((r2.x + r2.y) + (r3.x + r3.y));
            sum += __shfl_xor(sum, 1, WAVE);
            sum += __shfl_xor(sum, 2, WAVE);
            sum += __shfl_xor(sum, 4, WAVE);
            wave_fence();
        }
        // the results of FOUR groups leave together: group k of the four is kept by the lanes with lane & 7 == k (the sum is
        // on all eight lanes of its locus), and thirty-two lanes then store thirty-two consecutive loci -- whole 128-byte
        // lines (ranges start on multiples of 32 loci) instead of eight values per group: a store of a few bytes costs this
        // memory system what a kilobyte costs (profiles/r05_notes.md section 2)
        const int gslot = ((lb - l_begin) / QC_G) & 3;
        if (p8 == gslot) {
            keep_lc = my_lc;
            keep_ln = my_ln;
            keep_sum = sum;
        }
        if (gslot == 3 || lb + QC_G >= l_end) {
            const int lx = lb - gslot * QC_G + p8 * QC_G + vi;
            if (p8 <= gslot && lx < l_end) {
                const size_t at = (size_t)row * L + lx;
                a.l_calls[at] = keep_lc;
                if (HAS_Q) {
                    a.l_qn[at] = keep_ln;
                    a.l_qsum[at] = keep_sum;
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t at = (size_t)by * a.s_pad + s0 + j;
            a.p_calls[at] = calls[j];
            if (HAS_Q) {
                a.p_qn[at] = qn[j];
                a.p_qsum[at] = qsum[j];
            }
        }
    }
}

// out[i] = sum over rows r of part[r][i]: a workgroup = 64 columns x 4 slices of the rows, every slice added in row order
// and the four slice sums in slice order (float64 sums are reproducible; round 6: one thread per column walked 200 rows
// of the per-sample partials alone -- forty workgroups on the whole chip, 77 us)
__global__ __launch_bounds__(256) void k_qc_finish(int n, int n_rows, size_t row_stride, const uint32_t* __restrict__ pc,
                                                   const uint32_t* __restrict__ pn, const double* __restrict__ pq,
                                                   int64_t* __restrict__ oc, int64_t* __restrict__ on,
                                                   double* __restrict__ oq) {
    __shared__ int64_t sc[4][64], sm[4][64];
    __shared__ double sq[4][64];
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const int per = (n_rows + 3) / 4, r0 = slice * per, r1 = min(n_rows, r0 + per);
    int64_t c = 0, m = 0;
    double q = 0.0;
    if (i < n) {
        for (int r = r0; r < r1; ++r) {
            c += pc[(size_t)r * row_stride + i];
            if (pq) {
                m += pn[(size_t)r * row_stride + i];
                q += pq[(size_t)r * row_stride + i];
            }
        }
    }
    sc[slice][col] = c;
    sm[slice][col] = m;
    sq[slice][col] = q;
    __syncthreads();
    if (slice || i >= n) return;
    oc[i] = (sc[0][col] + sc[1][col]) + (sc[2][col] + sc[3][col]);
    if (pq) {
        if (on) on[i] = (sm[0][col] + sm[1][col]) + (sm[2][col] + sm[3][col]);
        if (oq) oq[i] = ((sq[0][col] + sq[1][col]) + sq[2][col]) + sq[3][col];
    }
}

}  // namespace

namespace trk {

static void qc_geometry(const trk_batch& b, const float* quality, int n_cu, int& V, int& gx, int& gy, int& lpw) {
    V = (b.ploidy == 2 && (b.n_samples % 4) == 0 && (reinterpret_cast<uintptr_t>(b.gt) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(quality) & 15) == 0)
            ? 4
            : 1;
    gx = (b.n_samples + QC_THREADS * V - 1) / (QC_THREADS * V);
    // ~8 workgroups per CU over the whole grid; a workgroup walks at least 64 loci (its partial rows are L wide)
    int wgcu = 8;
    if (const char* o = trk_opt("TRK_QC_WGCU")) wgcu = atoi(o) > 0 ? atoi(o) : wgcu;
    gy = n_cu * wgcu / gx;                // (rounded DOWN: 2050 workgroups for 2048 slots left two of them a round of their own)
    if (gy < 1) gy = 1;
    const int max_gy = (b.n_loci + 63) / 64;
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    lpw = (b.n_loci + gy - 1) / gy;
    lpw = (lpw + 4 * QC_G - 1) / (4 * QC_G) * (4 * QC_G);   // (32 loci: k_qc_scan4 stores the per-locus partials of four groups as whole lines)
    gy = (b.n_loci + lpw - 1) / lpw;
}

size_t qc_workspace_bytes(const trk_batch& b, const float* quality, int n_cu) {
    int V, gx, gy, lpw;
    qc_geometry(b, quality, n_cu, V, gx, gy, lpw);
    const size_t s_pad = (size_t)gx * QC_THREADS * V;
    return (size_t)gy * s_pad * 16 + (size_t)gx * QC_WAVES * b.n_loci * 16 + 256;
}

hipError_t launch_qc_reduce(const trk_batch& b, const trk_qc_params& prm, const trk_qc_out& out, void* workspace,
                            int n_cu, hipStream_t stream) {
    int V, gx, gy, lpw;
    qc_geometry(b, prm.quality, n_cu, V, gx, gy, lpw);
    QcArgs a;
    a.b = b;
    a.sample_in = prm.sample_in;
    a.quality = prm.quality;
    a.ignore_no_call = prm.ignore_no_call;
    a.loci_per_wg = lpw;
    a.s_pad = gx * QC_THREADS * V;
    a.n_rows = gx * QC_WAVES;
    a.gx = gx;
    a.gy = gy;
    unsigned char* w = static_cast<unsigned char*>(workspace);
    const size_t ns = (size_t)gy * a.s_pad, nl = (size_t)a.n_rows * b.n_loci;
    a.p_qsum = reinterpret_cast<double*>(w);
    a.l_qsum = a.p_qsum + ns;
    a.p_calls = reinterpret_cast<uint32_t*>(a.l_qsum + nl);
    a.p_qn = a.p_calls + ns;
    a.l_calls = a.p_qn + ns;
    a.l_qn = a.l_calls + nl;
    if (V == 4 && trk_opt("TRK_QC_OLD") == nullptr) {
        const bool hq = prm.quality != nullptr, ig = prm.ignore_no_call != 0;
        auto k = hq ? (ig ? k_qc_scan4<true, true> : k_qc_scan4<true, false>) : k_qc_scan4<false, false>;
        hipLaunchKernelGGL(k, dim3(gx * ((gy + 7) / 8 * 8)), dim3(QC_THREADS), 0, stream, a);
    } else if (V == 4)
        hipLaunchKernelGGL(k_qc_scan<4>, dim3(gx, gy), dim3(QC_THREADS), 0, stream, a);
    else
        hipLaunchKernelGGL(k_qc_scan<1>, dim3(gx, gy), dim3(QC_THREADS), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const bool q = prm.quality != nullptr;
    hipLaunchKernelGGL(k_qc_finish, dim3((b.n_samples + 63) / 64), dim3(256), 0, stream, b.n_samples, gy,
                       (size_t)a.s_pad, a.p_calls, a.p_qn, q ? a.p_qsum : nullptr, out.sample_calls, out.sample_qual_n,
                       out.sample_qual_sum);
    hipLaunchKernelGGL(k_qc_finish, dim3((b.n_loci + 63) / 64), dim3(256), 0, stream, b.n_loci, a.n_rows,
                       (size_t)b.n_loci, a.l_calls, a.l_qn, q ? a.l_qsum : nullptr, out.locus_calls, out.locus_qual_n,
                       out.locus_qual_sum);
    return hipGetLastError();
}

}  // namespace trk
