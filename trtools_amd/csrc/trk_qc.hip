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
//   k_qc_finish     partial rows added in row order (deterministic float64 sums) into the int64 / float64 outputs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

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

// out[i] = sum over rows r of part[r][i], rows in order (float64 sums are reproducible)
__global__ __launch_bounds__(256) void k_qc_finish(int n, int n_rows, size_t row_stride, const uint32_t* __restrict__ pc,
                                                   const uint32_t* __restrict__ pn, const double* __restrict__ pq,
                                                   int64_t* __restrict__ oc, int64_t* __restrict__ on,
                                                   double* __restrict__ oq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t c = 0, m = 0;
    double q = 0.0;
    for (int r = 0; r < n_rows; ++r) {
        c += pc[(size_t)r * row_stride + i];
        if (pq) {
            m += pn[(size_t)r * row_stride + i];
            q += pq[(size_t)r * row_stride + i];
        }
    }
    oc[i] = c;
    if (pq) {
        if (on) on[i] = m;
        if (oq) oq[i] = q;
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
    gy = (n_cu * 8 + gx - 1) / gx;
    const int max_gy = (b.n_loci + 63) / 64;
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    lpw = (b.n_loci + gy - 1) / gy;
    lpw = (lpw + QC_G - 1) / QC_G * QC_G;
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
    unsigned char* w = static_cast<unsigned char*>(workspace);
    const size_t ns = (size_t)gy * a.s_pad, nl = (size_t)a.n_rows * b.n_loci;
    a.p_qsum = reinterpret_cast<double*>(w);
    a.l_qsum = a.p_qsum + ns;
    a.p_calls = reinterpret_cast<uint32_t*>(a.l_qsum + nl);
    a.p_qn = a.p_calls + ns;
    a.l_calls = a.p_qn + ns;
    a.l_qn = a.l_calls + nl;
    if (V == 4)
        hipLaunchKernelGGL(k_qc_scan<4>, dim3(gx, gy), dim3(QC_THREADS), 0, stream, a);
    else
        hipLaunchKernelGGL(k_qc_scan<1>, dim3(gx, gy), dim3(QC_THREADS), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const bool q = prm.quality != nullptr;
    hipLaunchKernelGGL(k_qc_finish, dim3((b.n_samples + 255) / 256), dim3(256), 0, stream, b.n_samples, gy,
                       (size_t)a.s_pad, a.p_calls, a.p_qn, q ? a.p_qsum : nullptr, out.sample_calls, out.sample_qual_n,
                       out.sample_qual_sum);
    hipLaunchKernelGGL(k_qc_finish, dim3((b.n_loci + 255) / 256), dim3(256), 0, stream, b.n_loci, a.n_rows,
                       (size_t)b.n_loci, a.l_calls, a.l_qn, q ? a.l_qsum : nullptr, out.locus_calls, out.locus_qual_n,
                       out.locus_qual_sum);
    return hipGetLastError();
}

}  // namespace trk
