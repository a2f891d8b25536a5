// class_probe.hip -- what a FRESH process's first allocations look like to the call-filter pass's two write streams
// ("reserve the pair in trk_init ... and say whether that holds over 16 fresh processes").
// Allocates NP planes of [L, S] 4-byte cells as the very first thing the process does, times the write-only two-plane
// stream (the pass's tiling: a workgroup owns 1024 samples and walks `lpb` loci) for every pair (i, j), i < j, prints the
// matrix and the rule outcomes: is (0, 1) fast?  which is the first k with (0, k) fast?
// Build: hipcc --offload-arch=gfx950 -O3 -o class_probe class_probe.hip ; run: ./class_probe [NP] [L] [S] [pre_gb]
//   pre_gb: gigabytes allocated (and kept) BEFORE the planes, in 4 GB pieces -- the inputs of a real run
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_w2(u32x4* a, u32x4* b, int L, int S4, int lpb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        __builtin_nontemporal_store(r, a + o);
        __builtin_nontemporal_store(r + 1u, b + o);
    }
}
int main(int argc, char** argv) {
    const int NP = argc > 1 ? atoi(argv[1]) : 8;
    const int L = argc > 2 ? atoi(argv[2]) : 100000, S = argc > 3 ? atoi(argv[3]) : 10016;
    const int pre_gb = argc > 4 ? atoi(argv[4]) : 0;
    const int S4 = S / 4;
    const size_t plane = (size_t)L * S4 * 16;
    std::vector<void*> pre;
    for (int g = 0; g < pre_gb; g += 4) { void* p; CK(hipMalloc(&p, (size_t)4 << 30)); pre.push_back(p); }
    std::vector<u32x4*> pl(NP);
    for (int k = 0; k < NP; ++k) { CK(hipMalloc((void**)&pl[k], plane + 256)); CK(hipMemset(pl[k], 1, plane)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int gx = (S4 + 255) / 256, lpb = 80, gy = (L + lpb - 1) / lpb;
    std::vector<std::vector<float>> m(NP, std::vector<float>(NP, 0.f));
    float best = 1e30f, worst = 0;
    for (int i = 0; i < NP; ++i)
        for (int j = i + 1; j < NP; ++j) {
            hipLaunchKernelGGL(k_w2, dim3(gx, gy), dim3(256), 0, 0, pl[i], pl[j], L, S4, lpb);
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_w2, dim3(gx, gy), dim3(256), 0, 0, pl[i], pl[j], L, S4, lpb);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            m[i][j] = m[j][i] = ms / 3;
            best = ms / 3 < best ? ms / 3 : best;
            worst = ms / 3 > worst ? ms / 3 : worst;
        }
    const float cut = 0.5f * (best + worst);
    const bool two = worst > 1.08f * best;
    int first_fast = -1;
    for (int k = 1; k < NP && first_fast < 0; ++k) if (!two || m[0][k] < cut) first_fast = k;
    printf("planes %d x %.2f GB after %d GB; levels %.3f .. %.3f ms (%s); (0,1) %s; first k with (0,k) fast: %d; row 0:", NP,
           plane * 1e-9, pre_gb, best, worst, two ? "two levels" : "ONE level", (!two || m[0][1] < cut) ? "FAST" : "slow", first_fast);
    for (int k = 1; k < NP; ++k) printf(" %.2f", m[0][k]);
    printf("\n");
    if (getenv("MATRIX"))
        for (int i = 0; i < NP; ++i) { printf("#"); for (int j = 0; j < NP; ++j) printf(" %5.2f", m[i][j]); printf("\n"); }
    // classes: greedy grouping by "slow with each other"
    printf("classes:");
    std::vector<int> cls(NP, -1);
    int nc = 0;
    for (int i = 0; i < NP; ++i) {
        if (cls[i] >= 0) continue;
        cls[i] = nc;
        for (int j = i + 1; j < NP; ++j) if (cls[j] < 0 && two && m[i][j] >= cut) cls[j] = nc;
        ++nc;
    }
    for (int i = 0; i < NP; ++i) printf(" %d", cls[i]);
    printf("\n");
    return 0;
}
