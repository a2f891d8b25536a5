// Lane layout of v_mfma_f64_16x16x4_f64 on gfx950, discovered empirically (ad-hoc probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, int* rowmap, int* colmap) {
    // A[16][4], B[4][16]; guess: lane i supplies A[i%16][i/16], B[i/16][i%16]
    int i = threadIdx.x;
    double a = A[(i % 16) * 4 + i / 16];
    double b = B[(i / 16) * 16 + i % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[i * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int r = 0; r < 16; ++r) for (int k = 0; k < 4; ++k) hA[r * 4 + k] = 1 + r * 4 + k;          // distinct
    for (int k = 0; k < 4; ++k) for (int c = 0; c < 16; ++c) hB[k * 16 + c] = 1000 + 37 * k + 101 * c + (k * c % 7);
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[r * 4 + k] * hB[k * 16 + c]; ref[r * 16 + c] = s; }
    double *dA, *dB, *dD; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, nullptr, nullptr);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) {
        double v = hD[i * 4 + r]; int fr = -1, fc = -1;
        for (int x = 0; x < 256; ++x) if (ref[x] == v) { fr = x / 16; fc = x % 16; }
        if (i < 20 || i % 16 == 0) printf("lane %2d reg %d -> D[%d][%d]\n", i, r, fr, fc);
        int er = 4 * (i / 16) + r, ec = i % 16;
        if (fr != er || fc != ec) ok = 0;
    }
    printf("guess D[4*(lane/16)+reg][lane%%16]: %s\n", ok ? "CONFIRMED" : "WRONG");
    return 0;
}
