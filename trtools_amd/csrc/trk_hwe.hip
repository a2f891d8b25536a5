// trk_hwe.hip -- the exact binomial tests behind statSTR's / dumpSTR's HWE p-values (trtools/utils/utils.py:334-338:
// scipy.stats.binomtest(num_hom, n, exp_hom_frac).pvalue; restated in trk_binom.h), as kernels of their own
// translation unit.
//
// Why a file of its own: this file is compiled with `-mllvm -disable-machine-licm` (csrc/Makefile).  A test is a handful
// of pmf evaluations inside short loops, a pmf evaluation is log / log1p / exp polynomials, and the machine-level
// loop-invariant code motion lifts every one of their ~40 double constants out of those loops into vector registers
// that then stay live for the whole kernel: 155 VGPRs (or 128 + 108 B of scratch, or 72 + 332 B: rounds 2-4).  With
// the constants materialised where they are used the same source needs 89 registers and no scratch -- a kernel that
// fits beside the four call-filter waves of a SIMD (91 registers each) without spilling.  The streaming kernels of
// trk_kernels.hip keep the default (their loops WANT their invariants hoisted).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/trk.h"
#include "trk_binom.h"
#include "trk_internal.h"

namespace {
using trk::HweItem;
constexpr int HWE_THREADS = 64;

// Two neighbouring lanes per test: the kernel lasts as long as its slowest test, and a test is a chain of pmf
// evaluations that go two at a time this way (trk_binom.h).  On a large batch the kernel runs beside the call-filter
// kernel of the step, in the registers that one leaves free (128 per lane): it holds the pair routine alone -- the few
// tests the pair hands back (worklist header word 1 counts them) are done by k_hwe_test_serial right after.
__global__ __launch_bounds__(HWE_THREADS, 4) void k_hwe_test(unsigned int* __restrict__ hwe_count,
                                                             const HweItem* __restrict__ items,
                                                             double* __restrict__ locus_f64,
                                                             unsigned int* __restrict__ overflow) {
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int t = tid >> 1;
    if (t >= hwe_count[0]) return;
    const HweItem it = items[t];
    bool ok;
    const double pv = trkmath::binomtest_two_sided_pair(it.k, it.n, it.p, (int)(tid & 1), &ok);  // utils.py:334-338
    if (tid & 1) return;
    if (!ok) {
        overflow[atomicAdd(&hwe_count[1], 1u)] = t;
        return;
    }
    double* lf = locus_f64 + (int64_t)it.slot * TRK_LF_COLS;
    if (it.modes & 1) lf[TRK_LF_HWEP_LEN] = pv;
    if (it.modes & 2) lf[TRK_LF_HWEP_STR] = pv;
}

// one lane per test: the items of `list` (header word 1 entries), or every item (list == nullptr, TRK_HWE_SERIAL=1)
// (compiled for the registers the call-filter waves leave free, like the pair kernel: a launch that needs more waits
// for the whole call-filter kernel to retire, with nothing to do)
__global__ __launch_bounds__(HWE_THREADS, 4) void k_hwe_test_serial(const unsigned int* __restrict__ hwe_count,
                                                                    const HweItem* __restrict__ items,
                                                                    double* __restrict__ locus_f64,
                                                                    const unsigned int* __restrict__ list) {
    const unsigned int total = list ? hwe_count[1] : hwe_count[0];
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const HweItem it = items[list ? list[t] : t];
        double pv;
        [[clang::always_inline]] pv = trkmath::binomtest_two_sided(it.k, it.n, it.p);
        double* lf = locus_f64 + (int64_t)it.slot * TRK_LF_COLS;
        if (it.modes & 1) lf[TRK_LF_HWEP_LEN] = pv;
        if (it.modes & 2) lf[TRK_LF_HWEP_STR] = pv;
    }
}

// the fused small-batch pass's tests: fixed slots (k_locus_count_v3<.., FIN>), two lanes per test, the few tests the
// pair routine hands back finished in place by the serial routine
__global__ __launch_bounds__(HWE_THREADS) void k_hwe_test_slots(const HweItem* __restrict__ items, unsigned int n_slots,
                                                                double* __restrict__ locus_f64) {
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int t = tid >> 1;
    if (t >= n_slots) return;
    const HweItem it = items[t];
    if (it.modes == 0) return;
    bool ok;
    double pv = trkmath::binomtest_two_sided_pair(it.k, it.n, it.p, (int)(tid & 1), &ok);  // utils.py:334-338
    if (tid & 1) return;
    if (!ok) [[clang::always_inline]] pv = trkmath::binomtest_two_sided(it.k, it.n, it.p);
    double* lf = locus_f64 + (int64_t)it.slot * TRK_LF_COLS;
    if (it.modes & 1) lf[TRK_LF_HWEP_LEN] = pv;
    if (it.modes & 2) lf[TRK_LF_HWEP_STR] = pv;
}

// the lane-pair test on caller-supplied triples (trk_binomtest_batch: parity tests of the routine itself)
__global__ __launch_bounds__(HWE_THREADS) void k_binomtest_batch(const int64_t* __restrict__ k, const int64_t* __restrict__ n,
                                                                 const double* __restrict__ p, int64_t count,
                                                                 double* __restrict__ out, int lanes) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = lanes == 2 ? tid >> 1 : tid;
    if (t >= count) return;
    const int64_t kk = k[t], nn = n[t];
    const double pp = p[t];
    const bool valid = nn >= 1 && kk >= 0 && kk <= nn && pp >= 0.0 && pp <= 1.0;
    double pv = __builtin_nan("");
    if (valid) pv = lanes == 2 ? trkmath::binomtest_two_sided_pair_or_serial(kk, nn, pp, (int)(tid & 1)) : trkmath::binomtest_two_sided(kk, nn, pp);
    if (lanes != 2 || !(tid & 1)) out[t] = pv;
}
}  // namespace

namespace trk {
hipError_t launch_hwe_tests(unsigned int* count, const HweItem* items, double* locus_f64, unsigned int* overflow,
                            int64_t n, hipStream_t stream) {
    const int hwe_serial = trk_opt("TRK_HWE_SERIAL") ? 1 : 0;   // one lane per test, for A/B timing
    if (hwe_serial) {
        hipLaunchKernelGGL(k_hwe_test_serial, dim3((unsigned)((2 * n + HWE_THREADS - 1) / HWE_THREADS)), dim3(HWE_THREADS), 0,
                           stream, count, items, locus_f64, (const unsigned int*)nullptr);
        return hipGetLastError();
    }
    const int tblocks = (int)((4 * n + HWE_THREADS - 1) / HWE_THREADS);   // up to two tests per locus, two lanes per test
    hipLaunchKernelGGL(k_hwe_test, dim3(tblocks), dim3(HWE_THREADS), 0, stream, count, items, locus_f64, overflow);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hwe_test_serial, dim3(256), dim3(HWE_THREADS), 0, stream, count, items, locus_f64,
                       (const unsigned int*)overflow);
    return hipGetLastError();
}

hipError_t launch_hwe_slots(const HweItem* items, unsigned int n_slots, double* locus_f64, hipStream_t stream) {
    hipLaunchKernelGGL(k_hwe_test_slots, dim3((2 * n_slots + HWE_THREADS - 1) / HWE_THREADS), dim3(HWE_THREADS), 0, stream,
                       items, n_slots, locus_f64);
    return hipGetLastError();
}

hipError_t launch_binomtest_batch(const int64_t* k, const int64_t* n, const double* p, int64_t count, double* out,
                                  int lanes, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_binomtest_batch, dim3((unsigned)((lanes * count + HWE_THREADS - 1) / HWE_THREADS)), dim3(HWE_THREADS), 0,
                       stream, k, n, p, count, out, lanes);
    return hipGetLastError();
}
}  // namespace trk
