// rank_probe.hip -- are the placement classes of the two-output write stream thirds of the device's memory (HBM ranks)?
// Fresh process: plane P0, its neighbour P0b, a spacer of G GB, P1, a spacer of G GB, P2; hipMalloc / hipMemCreate
// times of the spacers; the write-only two-stream probe over every pair.
// Build: hipcc --offload-arch=gfx950 -O3 -o rank_probe rank_probe.hip ; run: ./rank_probe [G = 100]
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_w2(u32x4* a, u32x4* b, int L, int S4, int lpr, int gx) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int by = xcd + 8 * (slot / gx), bx = slot % gx;
    const int c = bx * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = by * lpr, l1 = min(L, l0 + lpr);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        __builtin_nontemporal_store(r, a + o);
        __builtin_nontemporal_store(r + 1u, b + o);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const double G = argc > 1 ? atof(argv[1]) : 100.0;
    const int L = 100000, S4 = 2504, gx = 10, nr = 96, lpr = (L + nr - 1) / nr;
    const size_t plane = (size_t)L * S4 * 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto probe = [&](void* a, void* b) {
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_w2, dim3(nr * gx), dim3(256), 30 * 1024, 0, (u32x4*)a, (u32x4*)b, L, S4, lpr, gx);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return best;
    };
    std::vector<void*> P;
    std::vector<const char*> name;
    auto plane_alloc = [&](const char* n) { void* p; CK(hipMalloc(&p, plane)); P.push_back(p); name.push_back(n); };
    plane_alloc("P0"); plane_alloc("P0b");
    void* sp[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) {
        const size_t bytes = (size_t)(G * 1073741824.0);
        double t = now();
        hipError_t e = hipMalloc(&sp[k], bytes);
        printf("hipMalloc(%.0f GB): %s, %.3f s\n", G, hipGetErrorString(e), now() - t);
        if (e != hipSuccess) { (void)hipGetLastError(); sp[k] = nullptr; }
        plane_alloc(k == 0 ? "P1" : "P2");
        if (k == 0) plane_alloc("P1b");
    }
    {   // how long does the same reservation take without a mapping?
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        hipMemGenericAllocationHandle_t h;
        for (int k = 0; k < 2; ++k) if (sp[k]) { double t = now(); CK(hipFree(sp[k])); printf("hipFree: %.3f s\n", now() - t); sp[k] = nullptr; }
        double t = now();
        hipError_t e = hipMemCreate(&h, (size_t)(G * 1073741824.0), &prop, 0);
        printf("hipMemCreate(%.0f GB): %s, %.3f s\n", G, hipGetErrorString(e), now() - t);
        if (e == hipSuccess) { t = now(); (void)hipMemRelease(h); printf("hipMemRelease: %.3f s\n", now() - t); } else (void)hipGetLastError();
    }
    printf("write-only pair probe (ms):      ");
    for (size_t j = 0; j < P.size(); ++j) printf("%6s ", name[j]);
    printf("\n");
    for (size_t i = 0; i < P.size(); ++i) {
        printf("%-32s ", name[i]);
        for (size_t j = 0; j < P.size(); ++j) {
            if (i == j) printf("   -   ");
            else printf("%6.3f ", probe(P[i], P[j]));
        }
        printf("\n");
    }
    return 0;
}
