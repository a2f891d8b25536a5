// Sustained rate of v_mfma_f64_16x16x4_f64 on this part: NACC independent accumulator tiles per wave, WAVES waves per
// SIMD, a long loop.  Prints cycles per instruction per SIMD and TFLOP/s.   hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(int wgs_per_cu, int ncu, int sclk_khz) {
    double* out;
    const int grid = ncu * wgs_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_inst_per_simd = (double)iters * NACC * wgs_per_cu;      // 4 waves per WG = one per SIMD
    const double flops = (double)grid * 4 * iters * NACC * 2048.0;
    printf("NACC %d, %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.1f cycles per instruction per SIMD at %d MHz\n", NACC,
           wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * sclk_khz * 1e3 / n_inst_per_simd, sclk_khz / 1000);
    hipFree(out);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int sclk = p.clockRate;
    run<1>(1, p.multiProcessorCount, sclk); run<4>(1, p.multiProcessorCount, sclk); run<8>(1, p.multiProcessorCount, sclk);
    run<4>(2, p.multiProcessorCount, sclk); run<4>(4, p.multiProcessorCount, sclk); run<2>(8, p.multiProcessorCount, sclk);
    return 0;
}
