// stream_probe.hip -- what does the memory system of an MI355X give a kernel with the call-filter pass's exact
// stream shape and NO arithmetic?  ("next round" item 1.i)
//
// The dumpSTR call-filter pass (k_call_filter_v2<3,true,false>) reads three [L,S] 4-byte planes (GT, DP, Q) and
// writes two (GT', mask): 12 B in + 8 B out per call, every access a 16-byte nontemporal vector, column-owner
// tiling (a thread owns 4 consecutive samples and walks a block of loci; grid = (S/1024, L/lpb)).
// This program times that shape, bare, next to the variations that could explain a gap:
//   cf<U>     column-owner tiling, U loci in flight per thread (the product kernel: U = 1)
//   cf+occ    the same with resident workgroups capped to 5 per CU by dynamic LDS (the product kernel's occupancy)
//   row       one wave per locus row (the count kernel's mapping), same five streams
//   flat      flat index space: thread i moves chunk i of every stream (a classic "triad")
//   read3 / write2 / copy   the read-only, write-only and 1:1 mixes at the same tiling
//   plain     ordinary loads / stores instead of nontemporal ones
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip ; run: ./stream_probe [L] [S] [reps]
// Output: one line per variant (min / avg ms over the repetitions, TB/s of the bytes the variant moves).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Streams {
    const u32x4* in[3];
    u32x4* out[2];
};

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// column-owner tiling: thread = one 16-byte chunk column (4 samples), walks the loci of its block
// persistent + strided: workgroup y takes loci y*U .. y*U+U-1, then + gridDim.y*U, ...: all resident workgroups march
// through the tensor together (a window of gridDim.y*U rows per stream)
template <int U, int NIN, int NOUT, bool NT>
__global__ __launch_bounds__(256) void k_cfs(Streams s, int L, int S4, uint32_t* sink) {
    extern __shared__ uint32_t dummy[];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    u32x4 acc = {0, 0, 0, 0};
    for (int l = blockIdx.y * U; l < L; l += gridDim.y * U) {
        u32x4 v[U][NIN > 0 ? NIN : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (l + u < L) v[u][k] = ld<NT>(s.in[k] + (size_t)(l + u) * S4 + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (l + u >= L) break;
            u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
#pragma unroll
            for (int k = 0; k < NIN; ++k) r |= v[u][k];
            if (NOUT == 0) acc ^= r;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) st<NT>(s.out[k] + (size_t)(l + u) * S4 + c, r + (uint32_t)k);
        }
    }
    if (NOUT == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
    if (dummy && L < 0) dummy[threadIdx.x] = 0;
}

template <int U, int NIN, int NOUT, bool NT>
__global__ __launch_bounds__(256) void k_cf(Streams s, int L, int S4, int lpb, uint32_t* sink) {
    extern __shared__ uint32_t dummy[];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    u32x4 acc = {0, 0, 0, 0};
    int l = l0;
    for (; l + U <= l1; l += U) {
        u32x4 v[U][NIN > 0 ? NIN : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < NIN; ++k) v[u][k] = ld<NT>(s.in[k] + (size_t)(l + u) * S4 + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
#pragma unroll
            for (int k = 0; k < NIN; ++k) r |= v[u][k];
            if (NOUT == 0) acc ^= r;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) st<NT>(s.out[k] + (size_t)(l + u) * S4 + c, r + (uint32_t)k);
        }
    }
    for (; l < l1; ++l) {
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < NIN; ++k) r |= ld<NT>(s.in[k] + (size_t)l * S4 + c);
        if (NOUT == 0) acc ^= r;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) st<NT>(s.out[k] + (size_t)l * S4 + c, r + (uint32_t)k);
    }
    if (NOUT == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// the product shape with data of the product's kind.  MODE 0: outputs = constants derived from the loop counter;
// 1: out0 = in0 (the masked genotypes are the genotypes), out1 = sparse word (a few % non-zero, like the filter mask);
// 2: out0 = in0, out1 = in1 ^ in2 (dense random words); 3: both outputs dense random
template <int MODE>
__global__ __launch_bounds__(256) void k_cf_data(Streams s, int L, int S4, int lpb, uint32_t thr) {
    extern __shared__ uint32_t dummy[];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        const u32x4 a = __builtin_nontemporal_load(s.in[0] + o), b = __builtin_nontemporal_load(s.in[1] + o),
                    d = __builtin_nontemporal_load(s.in[2] + o);
        u32x4 r0, r1;
        if (MODE == 0) {
            r0 = (u32x4){(uint32_t)l, 1u, 2u, 3u};
            r1 = r0 + 1u;
            if ((a.x ^ b.x ^ d.x) == 0x12345u && (a.y ^ b.y ^ d.y) == 0x54321u) r1.x = 7u;   // keep the loads
        } else if (MODE == 1) {
            r0 = a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = (b[j] & 0xffffu) < thr;
                r1[j] = hit ? 5u : ((d[j] & 0xffu) == 0u ? 0x80000000u : 0u);
                if (hit) r0[j] = 0xffffffffu;
            }
        } else if (MODE == 2) {
            r0 = a;
            r1 = b ^ d;
        } else {
            r0 = a ^ b;
            r1 = b ^ d;
        }
        __builtin_nontemporal_store(r0, s.out[0] + o);
        __builtin_nontemporal_store(r1, s.out[1] + o);
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed, int kind) {
    // kind 0: uniform random words; 1: small-integer pairs in 16-bit halves (genotype-like); 2: small ints (depth-like)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        uint32_t v = (uint32_t)z;
        if (kind == 1) v = ((v & 7u) | (((v >> 8) & 7u) << 16));
        else if (kind == 2) v = 10u + (v & 63u);
        p[i] = v;
    }
}

// column-owner tiling, generalised: TPB threads per workgroup, V adjacent 16-byte chunks per thread (a wave covers
// V KB contiguous per stream and locus), XY = 1: blockIdx.x walks the locus blocks and blockIdx.y the column tiles
// (workgroups launched together are neighbours in LOCI instead of in columns)
template <int TPB, int V, int XY>
__global__ __launch_bounds__(TPB) void k_cfg(Streams s, int L, int S4, int lpb) {
    extern __shared__ uint32_t dummy[];
    const int bx = XY ? blockIdx.y : blockIdx.x, by = XY ? blockIdx.x : blockIdx.y;
    const int c0 = (bx * TPB + threadIdx.x) * V;
    if (c0 >= S4) return;
    const int l0 = by * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        u32x4 a[V][3];
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (c0 + v < S4) a[v][k] = __builtin_nontemporal_load(s.in[k] + (size_t)l * S4 + c0 + v);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (c0 + v >= S4) break;
            u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
            r |= a[v][0] | a[v][1] | a[v][2];
            __builtin_nontemporal_store(r, s.out[0] + (size_t)l * S4 + c0 + v);
            __builtin_nontemporal_store(r + 1u, s.out[1] + (size_t)l * S4 + c0 + v);
        }
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// the column-owner walk over rows that are RS4 chunks apart while only C4 chunks of each are used (a padded row
// stride: the pad is never touched)
template <int XY>
__global__ __launch_bounds__(256) void k_cfr(Streams s, int L, int C4, int RS4, int lpb) {
    extern __shared__ uint32_t dummy[];
    const int bx = XY ? blockIdx.y : blockIdx.x, by = XY ? blockIdx.x : blockIdx.y;
    const int c = bx * 256 + threadIdx.x;
    if (c >= C4) return;
    const int l0 = by * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * RS4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
        __builtin_nontemporal_store(r, s.out[0] + o);
        __builtin_nontemporal_store(r + 1u, s.out[1] + o);
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// the column-owner walk with the STORES' cache policy spelled out (loads: nontemporal as in the product).
// POL 0: nt (the product's), 1: plain, 2: sc1, 3: sc0 sc1, 4: nt sc1, 5: nt sc0 sc1
template <int POL>
__device__ __forceinline__ void st_pol(u32x4* p, u32x4 v) {
    if (POL == 0) __builtin_nontemporal_store(v, p);
    else if (POL == 1) *p = v;
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int POL, bool LNT = true>
__global__ __launch_bounds__(256) void k_cfp(Streams s, int L, int S4, int lpb) {
    extern __shared__ uint32_t dummy[];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        r |= ld<LNT>(s.in[0] + o) | ld<LNT>(s.in[1] + o) | ld<LNT>(s.in[2] + o);
        st_pol<POL>(s.out[0] + o, r);
        st_pol<POL>(s.out[1] + o, r + 1u);
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// one wave per locus row, 4 waves per workgroup, U chunks in flight per lane
template <int U, int NIN, int NOUT, bool NT>
__global__ __launch_bounds__(256) void k_row(Streams s, int L, int S4, uint32_t* sink) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= L) return;
    const int lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    const size_t base = (size_t)l * S4;
    for (int c0 = lane; c0 < S4; c0 += 64 * U) {
        u32x4 v[U][NIN > 0 ? NIN : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 64 * u;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (c < S4) v[u][k] = ld<NT>(s.in[k] + base + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 64 * u;
            if (c >= S4) break;
            u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
#pragma unroll
            for (int k = 0; k < NIN; ++k) r |= v[u][k];
            if (NOUT == 0) acc ^= r;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) st<NT>(s.out[k] + base + c, r + (uint32_t)k);
        }
    }
    if (NOUT == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// flat: grid-stride over all chunks
template <int NIN, int NOUT, bool NT>
__global__ __launch_bounds__(256) void k_flat(Streams s, size_t n, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 r = {(uint32_t)i, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < NIN; ++k) r |= ld<NT>(s.in[k] + i);
        if (NOUT == 0) acc ^= r;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) st<NT>(s.out[k] + i, r + (uint32_t)k);
    }
    if (NOUT == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// flat, PER consecutive 4 KB blocks per workgroup (a thread's chunks are 4 KB apart); ALL: every load first, then
// every store (one wait), else load/store block by block
template <int PER, bool ALL>
__global__ __launch_bounds__(256) void k_flat_adj(Streams s, size_t n) {
    const size_t i0 = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
    if (ALL) {
        u32x4 r[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const size_t i = i0 + (size_t)p * 256;
            r[p] = (u32x4){(uint32_t)i, 1u, 2u, 3u};
            if (i < n) r[p] |= ld<true>(s.in[0] + i) | ld<true>(s.in[1] + i) | ld<true>(s.in[2] + i);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const size_t i = i0 + (size_t)p * 256;
            if (i < n) { st<true>(s.out[0] + i, r[p]); st<true>(s.out[1] + i, r[p] + 1u); }
        }
    } else {
#pragma unroll 1
        for (int p = 0; p < PER; ++p) {
            const size_t i = i0 + (size_t)p * 256;
            if (i >= n) break;
            u32x4 r = {(uint32_t)i, 1u, 2u, 3u};
            r |= ld<true>(s.in[0] + i) | ld<true>(s.in[1] + i) | ld<true>(s.in[2] + i);
            st<true>(s.out[0] + i, r);
            st<true>(s.out[1] + i, r + 1u);
        }
    }
}

struct Result { std::string name; double mn, avg, bytes; };
static std::vector<Result> results;
static hipEvent_t e0, e1;
static int reps = 7;

template <typename F>
static void run(const char* name, double bytes, F launch) {
    launch();   // warm-up
    CK(hipDeviceSynchronize());
    double mn = 1e30, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        mn = std::min(mn, (double)ms);
        sum += ms;
    }
    CK(hipGetLastError());
    results.push_back({name, mn, sum / reps, bytes});
    printf("%-44s min %7.3f ms  avg %7.3f ms  %6.2f TB/s (avg)  %5.3f of 8 TB/s\n", name, mn, sum / reps,
           bytes / (sum / reps) * 1e-9, bytes / (sum / reps) * 1e-9 / 8.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int L = argc > 1 ? atoi(argv[1]) : 100000, S = argc > 2 ? atoi(argv[2]) : 10000;
    if (argc > 3) reps = atoi(argv[3]);
    const int S4 = S / 4;
    const size_t n = (size_t)L * S4, plane = n * 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int mclk = 0, sclk = 0;
    hipDeviceGetAttribute(&mclk, hipDeviceAttributeMemoryClockRate, 0);
    hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
    printf("# device %s, %d CUs, sclk %d kHz, mclk %d kHz; L = %d, S = %d, plane = %.2f GB, reps = %d\n", prop.name,
           prop.multiProcessorCount, sclk, mclk, L, S, plane * 1e-9, reps);
    Streams s;
    void* buf[5];
    for (int k = 0; k < 5; ++k) {
        // SKEW=<bytes>: stream k starts k * SKEW bytes into its allocation (do the five streams, walked at the same
        // offset, meet on the same channels?)
        const size_t skew = getenv("SKEW") ? (size_t)atoll(getenv("SKEW")) * k : 0;
        CK(hipMalloc(&buf[k], plane + skew + 256));
        CK(hipMemset(buf[k], k + 1, plane + skew));
        buf[k] = (char*)buf[k] + skew;
        if (k == 0 && getenv("SKEW")) printf("# SKEW %s bytes per stream\n", getenv("SKEW"));
    }
    for (int k = 0; k < 3; ++k) s.in[k] = (const u32x4*)buf[k];
    s.out[0] = (u32x4*)buf[3];
    s.out[1] = (u32x4*)buf[4];
    uint32_t* sink;
    CK(hipMalloc(&sink, 64));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    const int ncu = prop.multiProcessorCount;
    const int gx = (S4 + 255) / 256;
    // the product kernel's grid rule: whole rounds of (occupancy x CUs) workgroups
    auto lpb_for = [&](int occ) {
        const long slots = (long)ncu * occ;
        int lpb = 117;
        const long min_wgs = (long)gx * ((L + lpb - 1) / lpb);
        long k = (min_wgs + slots - 1) / slots;
        if (k < 2) k = 2;
        long gyr = k * slots / gx;
        int lpb2 = (int)((L + gyr - 1) / gyr);
        return lpb2 < lpb ? lpb2 : lpb;
    };
    const double b5 = 5.0 * plane, b3 = 3.0 * plane, b2 = 2.0 * plane;
    if (argc > 4 && !strcmp(argv[4], "data")) {
        // does the DATA matter?  product shape, product grid; inputs constant bytes / realistic / random
        const int lpb = lpb_for(5), gy = (L + lpb - 1) / lpb;
        const size_t lds5 = 30 * 1024;
        const char* fills[3] = {"inputs memset bytes", "inputs genotype-/depth-like small integers", "inputs uniform random words"};
        for (int f = 0; f < 3; ++f) {
            if (f > 0)
                for (int k = 0; k < 3; ++k) {
                    hipLaunchKernelGGL(k_fill, dim3(ncu * 16), dim3(256), 0, 0, (uint32_t*)buf[k], n * 4, 77u + k,
                                       f == 2 ? 0 : (k == 0 ? 1 : 2));
                }
            CK(hipDeviceSynchronize());
            char nm[160];
            snprintf(nm, sizeof nm, "data: %s; outputs constants", fills[f]);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cf_data<0>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, 2000u); });
            snprintf(nm, sizeof nm, "data: %s; out0 = in0, out1 sparse (3 %% set)", fills[f]);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cf_data<1>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, f == 1 ? 12u : 2000u); });
            snprintf(nm, sizeof nm, "data: %s; out0 = in0, out1 = in1 ^ in2", fills[f]);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cf_data<2>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, 0u); });
            snprintf(nm, sizeof nm, "data: %s; both outputs xor of inputs", fills[f]);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cf_data<3>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, 0u); });
        }
        // the plain copy / read figures on random data
        run("random data: flat 1in/1out nt (copy), 16 x CUs", b2, [&] { hipLaunchKernelGGL((k_flat<1, 1, true>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("random data: flat 3in/0out nt (read), 16 x CUs", b3, [&] { hipLaunchKernelGGL((k_flat<3, 0, true>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("random data: row<2> 1in/0out nt (count kernel's stream)", 1.0 * plane, [&] { hipLaunchKernelGGL((k_row<2, 1, 0, true>), dim3((L + 3) / 4), dim3(256), 0, 0, s, L, S4, sink); });
        printf("JSON [");
        for (size_t i = 0; i < results.size(); ++i)
            printf("%s{\"name\": \"%s\", \"min_ms\": %.4f, \"avg_ms\": %.4f, \"tbps\": %.3f}", i ? ", " : "",
                   results[i].name.c_str(), results[i].mn, results[i].avg, results[i].bytes / results[i].avg * 1e-9);
        printf("]\n");
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "arena")) {
        // ONE allocation holding the five streams, stream k at k * (plane rounded up to 2 MB + skew): is the relative
        // placement of the streams (which channels the same offset of each stream falls on) worth anything, and is it
        // reproducible from run to run?  (Separately allocated planes: 3.22 ... 3.83 ms from process to process.)
        const size_t slot = (plane + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        const size_t max_skew = (size_t)48 << 20;
        char* arena;
        CK(hipMalloc(&arena, 5 * (slot + max_skew) + 4096));
        CK(hipMemset(arena, 1, 5 * (slot + max_skew)));
        const int lpb = lpb_for(5), gy = (L + lpb - 1) / lpb;
        const size_t skews[] = {0, 256, 4096, 65536, 131072, 262144, 524288, 1048576, 1064960, 1310720, 2097152, 3145728,
                                4194304, 5242880, 8388608, 12582912, 16777216, 25165824, 33554432, 0};
        for (size_t sk : skews) {
            Streams a;
            for (int k = 0; k < 3; ++k) a.in[k] = (const u32x4*)(arena + k * (slot + sk));
            a.out[0] = (u32x4*)(arena + 3 * (slot + sk));
            a.out[1] = (u32x4*)(arena + 4 * (slot + sk));
            char nm[128];
            snprintf(nm, sizeof nm, "arena: streams %zu MB + %zu bytes apart", slot >> 20, sk);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfp<0>), dim3(gx, gy), dim3(256), 30 * 1024, 0, a, L, S4, lpb); });
        }
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "policy")) {
        // cache policy of the two output streams (the guide: plain / sc0 / nt stores KEEP the line in the XCD's L2,
        // sc1 / sc0 sc1 DROP it): does writing through help a stream that never reads its output again?
        const int lpb = lpb_for(5), gy = (L + lpb - 1) / lpb;
        for (int round = 0; round < 2; ++round) {
#define PL(P, NAME) run("policy: stores " NAME, b5, [&] { hipLaunchKernelGGL((k_cfp<P>), dim3(gx, gy), dim3(256), 30 * 1024, 0, s, L, S4, lpb); });
            PL(0, "nt (product)") PL(1, "plain") PL(2, "sc1") PL(3, "sc0 sc1") PL(4, "sc1 nt") PL(5, "sc0 sc1 nt")
#undef PL
            run("policy: loads plain, stores nt", b5, [&] { hipLaunchKernelGGL((k_cfp<0, false>), dim3(gx, gy), dim3(256), 30 * 1024, 0, s, L, S4, lpb); });
        }
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "stride")) {
        // S (argv[2]) is the allocation's row length; COLS of each row are used, rows STRIDE samples apart for a list
        // of strides: which alignment of the row start does the column-owner walk need?
        const int cols = getenv("COLS") ? atoi(getenv("COLS")) : 10000;
        const int C4 = cols / 4, gxc = (C4 + 255) / 256;
        const double bytes = 20.0 * L * cols;
        const int lpb = lpb_for(5), gy = (L + lpb - 1) / lpb;
        for (int stride : {10000, 10016, 10048, 10112, 10240, 10496, 10752, 11264, 12288}) {
            if (stride > S || stride < cols) continue;
            char nm[160];
            snprintf(nm, sizeof nm, "stride %5d samples (%6d B, aligned to %4d B): %d loci/block column-major", stride, stride * 4,
                     (stride * 4) & -(stride * 4), lpb);
            run(nm, bytes, [&] { hipLaunchKernelGGL((k_cfr<0>), dim3(gxc, gy), dim3(256), 30 * 1024, 0, s, L, C4, stride / 4, lpb); });
            snprintf(nm, sizeof nm, "stride %5d samples (%6d B, aligned to %4d B): %d loci/block locus-major", stride, stride * 4,
                     (stride * 4) & -(stride * 4), lpb);
            run(nm, bytes, [&] { hipLaunchKernelGGL((k_cfr<1>), dim3(gy, gxc), dim3(256), 30 * 1024, 0, s, L, C4, stride / 4, lpb); });
            const int gy32 = (L + 31) / 32;
            snprintf(nm, sizeof nm, "stride %5d samples (%6d B, aligned to %4d B): 32 loci/block locus-major", stride, stride * 4,
                     (stride * 4) & -(stride * 4));
            run(nm, bytes, [&] { hipLaunchKernelGGL((k_cfr<1>), dim3(gy32, gxc), dim3(256), 30 * 1024, 0, s, L, C4, stride / 4, 32); });
        }
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "short")) {
        // how short-lived must a workgroup be to stream like the one-chunk-per-thread kernel?  loci per block 1..32 of
        // the column-owner walk (both launch orders), and flat grid-stride kernels with 2..8 chunks per thread
        for (int lpb : {1, 2, 3, 4, 8, 16, 32}) {
            const int gyy = (L + lpb - 1) / lpb;
            char nm[128];
            snprintf(nm, sizeof nm, "short: column-owner, %d loci/block, column-major launch", lpb);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfg<256, 1, 0>), dim3(gx, gyy), dim3(256), 0, 0, s, L, S4, lpb); });
            snprintf(nm, sizeof nm, "short: column-owner, %d loci/block, locus-major launch", lpb);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfg<256, 1, 1>), dim3(gyy, gx), dim3(256), 0, 0, s, L, S4, lpb); });
        }
        for (int per : {1, 2, 4, 8, 16}) {
            const unsigned g = (unsigned)((n + 256ull * per - 1) / (256ull * per));
            char nm[128];
            snprintf(nm, sizeof nm, "short: flat grid-stride, %d chunks per thread (grid %u)", per, g);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_flat<3, 2, true>), dim3(g), dim3(256), 0, 0, s, n, sink); });
        }
#define FA(PER, ALL)                                                                                              \
        {                                                                                                         \
            const unsigned g = (unsigned)((n + 256ull * PER - 1) / (256ull * PER));                               \
            char nm[128];                                                                                         \
            snprintf(nm, sizeof nm, "short: flat, %d adjacent 4 KB blocks per workgroup, %s", PER,                \
                     ALL ? "all loads then all stores" : "block by block");                                       \
            run(nm, b5, [&] { hipLaunchKernelGGL((k_flat_adj<PER, ALL>), dim3(g), dim3(256), 0, 0, s, n); });     \
        }
        FA(2, false) FA(2, true) FA(4, false) FA(4, true) FA(8, false) FA(16, false)
#undef FA
        for (int lpb : {49, 112, 400}) {
            const int gyy = (L + lpb - 1) / lpb;
            char nm[128];
            snprintf(nm, sizeof nm, "short: column-owner, %d loci/block, column-major launch", lpb);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfg<256, 1, 0>), dim3(gx, gyy), dim3(256), 0, 0, s, L, S4, lpb); });
        }
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "tiles")) {
        // workgroup width / chunks per thread / launch order of the column-owner tiling (5 workgroups' worth of
        // waves per CU kept by dynamic LDS where the workgroup is 256 threads)
        const int lpb = lpb_for(5);
        const size_t lds5 = 30 * 1024;
#define TL(TPB, V, XY, LDS, LPB)                                                                                  \
        {                                                                                                         \
            const int gxx = (S4 + TPB * V - 1) / (TPB * V), gyy = (L + (LPB) - 1) / (LPB);                          \
            char nm[128];                                                                                         \
            snprintf(nm, sizeof nm, "tiles: %d threads x %d chunks, %s-major launch, %d loci/block", TPB, V,       \
                     XY ? "locus" : "column", LPB);                                                                \
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfg<TPB, V, XY>), XY ? dim3(gyy, gxx) : dim3(gxx, gyy), dim3(TPB), LDS, 0, s, L, S4, LPB); }); \
        }
        TL(256, 1, 0, lds5, lpb) TL(256, 1, 1, lds5, lpb) TL(256, 2, 0, lds5, lpb) TL(256, 2, 1, lds5, lpb)
        TL(256, 4, 0, lds5, lpb) TL(512, 1, 0, 60 * 1024, lpb) TL(1024, 1, 0, 0, lpb) TL(1024, 1, 1, 0, lpb)
        TL(128, 1, 0, 15 * 1024, lpb) TL(64, 1, 0, 7 * 1024, lpb) TL(64, 1, 1, 7 * 1024, lpb)
        TL(256, 1, 0, lds5, 16) TL(256, 1, 1, lds5, 16) TL(256, 2, 1, lds5, 16) TL(64, 1, 1, 7 * 1024, 16)
        TL(256, 1, 1, lds5, 4) TL(64, 4, 1, 7 * 1024, 8)
#undef TL
        run("flat 3in/2out nt, one chunk per thread", b5, [&] { hipLaunchKernelGGL((k_flat<3, 2, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, s, n, sink); });
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "sweep")) {
        // resident workgroups per CU (capped by dynamic LDS) x loci in flight x locus assignment, grid = exactly the
        // resident set (persistent workgroups)
        for (int wgcu : {2, 3, 4, 5, 6, 8}) {
            const size_t lds = wgcu >= 8 ? 0 : std::min<size_t>(64 * 1024, (size_t)(160 * 1024 / wgcu - 2048));
            const int gy = std::max(1, wgcu * ncu / gx);
            const int lpb = (L + gy - 1) / gy;
            char nm[128];
#define SW(UU)                                                                                                      \
            snprintf(nm, sizeof nm, "sweep cf<%d> block   %d WG/CU persistent (gy %d, %d loci/WG)", UU, wgcu, gy, lpb); \
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cf<UU, 3, 2, true>), dim3(gx, gy), dim3(256), lds, 0, s, L, S4, lpb, sink); }); \
            snprintf(nm, sizeof nm, "sweep cf<%d> strided %d WG/CU persistent (gy %d)", UU, wgcu, gy);             \
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfs<UU, 3, 2, true>), dim3(gx, gy), dim3(256), lds, 0, s, L, S4, sink); });
            SW(1) SW(2) SW(4)
#undef SW
        }
        // the product grid rule (whole rounds) with strided-in-round assignment is the same as block for one round;
        // non-persistent strided: gy = L / lpb workgroups each taking lpb loci gy apart
        for (int lpb : {16, 112}) {
            const int gy = (L + lpb - 1) / lpb;
            char nm[128];
            snprintf(nm, sizeof nm, "sweep cf<1> strided, %d loci per WG (gy %d), 5 WG/CU", lpb, gy);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cfs<1, 3, 2, true>), dim3(gx, gy), dim3(256), 30 * 1024, 0, s, L, S4, sink); });
        }
        printf("JSON [");
        for (size_t i = 0; i < results.size(); ++i)
            printf("%s{\"name\": \"%s\", \"min_ms\": %.4f, \"avg_ms\": %.4f, \"tbps\": %.3f}", i ? ", " : "",
                   results[i].name.c_str(), results[i].mn, results[i].avg, results[i].bytes / results[i].avg * 1e-9);
        printf("]\n");
        return 0;
    }
    {
        const int lpb = lpb_for(5), gy = (L + lpb - 1) / lpb;
        printf("# column-owner grid (%d, %d), %d loci per block\n", gx, gy, lpb);
        const size_t lds5 = 30 * 1024;   // 160 KiB / 5 > 30 KiB > 160 KiB / 6: five workgroups per CU
        run("cf<1> 3in/2out nt, 5 WG/CU (product shape)", b5, [&] { hipLaunchKernelGGL((k_cf<1, 3, 2, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<1> 3in/2out nt, 8 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<1, 3, 2, true>), dim3(gx, gy), dim3(256), 0, 0, s, L, S4, lpb, sink); });
        run("cf<2> 3in/2out nt, 5 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<2, 3, 2, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<2> 3in/2out nt, 8 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<2, 3, 2, true>), dim3(gx, gy), dim3(256), 0, 0, s, L, S4, lpb, sink); });
        run("cf<4> 3in/2out nt, 5 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<4, 3, 2, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<4> 3in/2out nt, 8 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<4, 3, 2, true>), dim3(gx, gy), dim3(256), 0, 0, s, L, S4, lpb, sink); });
        const size_t lds3 = 50 * 1024, lds2 = 70 * 1024;
        run("cf<1> 3in/2out nt, 3 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<1, 3, 2, true>), dim3(gx, gy), dim3(256), lds3, 0, s, L, S4, lpb, sink); });
        run("cf<4> 3in/2out nt, 2 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<4, 3, 2, true>), dim3(gx, gy), dim3(256), lds2, 0, s, L, S4, lpb, sink); });
        run("cf<1> 3in/2out plain ld/st, 5 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<1, 3, 2, false>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<2> 3in/2out plain ld/st, 8 WG/CU", b5, [&] { hipLaunchKernelGGL((k_cf<2, 3, 2, false>), dim3(gx, gy), dim3(256), 0, 0, s, L, S4, lpb, sink); });
        run("cf<1> 3in/0out nt (read only), 5 WG/CU", b3, [&] { hipLaunchKernelGGL((k_cf<1, 3, 0, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<2> 3in/0out nt (read only), 8 WG/CU", b3, [&] { hipLaunchKernelGGL((k_cf<2, 3, 0, true>), dim3(gx, gy), dim3(256), 0, 0, s, L, S4, lpb, sink); });
        run("cf<1> 0in/2out nt (write only), 5 WG/CU", b2, [&] { hipLaunchKernelGGL((k_cf<1, 0, 2, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<1> 1in/1out nt (copy), 5 WG/CU", b2, [&] { hipLaunchKernelGGL((k_cf<1, 1, 1, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<2> 1in/1out nt (copy), 8 WG/CU", b2, [&] { hipLaunchKernelGGL((k_cf<2, 1, 1, true>), dim3(gx, gy), dim3(256), 0, 0, s, L, S4, lpb, sink); });
        run("cf<1> 3in/1out nt (16 B/call), 5 WG/CU", 4.0 * plane, [&] { hipLaunchKernelGGL((k_cf<1, 3, 1, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        run("cf<1> 2in/2out nt (fused-count analogue)", 4.0 * plane, [&] { hipLaunchKernelGGL((k_cf<1, 2, 2, true>), dim3(gx, gy), dim3(256), lds5, 0, s, L, S4, lpb, sink); });
        // other block sizes of the same tiling
        for (int q : {16, 49, 400, 2000}) {
            const int gy2 = (L + q - 1) / q;
            char nm[96];
            snprintf(nm, sizeof nm, "cf<1> 3in/2out nt, 5 WG/CU, %d loci/block", q);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_cf<1, 3, 2, true>), dim3(gx, gy2), dim3(256), lds5, 0, s, L, S4, q, sink); });
        }
    }
    {
        const int g = (L + 3) / 4;
        run("row<1> 3in/2out nt (wave per locus row)", b5, [&] { hipLaunchKernelGGL((k_row<1, 3, 2, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<2> 3in/2out nt", b5, [&] { hipLaunchKernelGGL((k_row<2, 3, 2, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<4> 3in/2out nt", b5, [&] { hipLaunchKernelGGL((k_row<4, 3, 2, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<2> 3in/2out plain", b5, [&] { hipLaunchKernelGGL((k_row<2, 3, 2, false>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<2> 1in/0out nt (count kernel's stream)", 1.0 * plane, [&] { hipLaunchKernelGGL((k_row<2, 1, 0, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<4> 1in/0out nt", 1.0 * plane, [&] { hipLaunchKernelGGL((k_row<4, 1, 0, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<2> 3in/0out nt", b3, [&] { hipLaunchKernelGGL((k_row<2, 3, 0, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
        run("row<2> 1in/1out nt (copy)", b2, [&] { hipLaunchKernelGGL((k_row<2, 1, 1, true>), dim3(g), dim3(256), 0, 0, s, L, S4, sink); });
    }
    {
        for (int wpc : {8, 16, 32}) {
            char nm[96];
            snprintf(nm, sizeof nm, "flat 3in/2out nt, grid %d x CUs", wpc);
            run(nm, b5, [&] { hipLaunchKernelGGL((k_flat<3, 2, true>), dim3(ncu * wpc), dim3(256), 0, 0, s, n, sink); });
        }
        run("flat 3in/2out nt, one chunk per thread", b5, [&] { hipLaunchKernelGGL((k_flat<3, 2, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, s, n, sink); });
        run("flat 3in/2out plain, grid 16 x CUs", b5, [&] { hipLaunchKernelGGL((k_flat<3, 2, false>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("flat 1in/1out plain (float4 copy), 16 x CUs", b2, [&] { hipLaunchKernelGGL((k_flat<1, 1, false>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("flat 1in/1out nt (copy), 16 x CUs", b2, [&] { hipLaunchKernelGGL((k_flat<1, 1, true>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("flat 1in/0out nt (read), 16 x CUs", 1.0 * plane, [&] { hipLaunchKernelGGL((k_flat<1, 0, true>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("flat 3in/0out nt (read), 16 x CUs", b3, [&] { hipLaunchKernelGGL((k_flat<3, 0, true>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("flat 0in/2out nt (write), 16 x CUs", b2, [&] { hipLaunchKernelGGL((k_flat<0, 2, true>), dim3(ncu * 16), dim3(256), 0, 0, s, n, sink); });
        run("hipMemcpyAsync D2D (one plane)", b2, [&] { CK(hipMemcpyAsync(buf[3], buf[0], plane, hipMemcpyDeviceToDevice, 0)); });
    }
    // machine-readable tail
    printf("JSON [");
    for (size_t i = 0; i < results.size(); ++i)
        printf("%s{\"name\": \"%s\", \"min_ms\": %.4f, \"avg_ms\": %.4f, \"tbps\": %.3f}", i ? ", " : "", results[i].name.c_str(),
               results[i].mn, results[i].avg, results[i].bytes / results[i].avg * 1e-9);
    printf("]\n");
    return 0;
}
