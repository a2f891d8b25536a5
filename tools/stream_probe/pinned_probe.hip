// pinned_probe.hip -- the call-filter pass's stream shape (3 x 16 B/lane in, 2 out, no arithmetic) with the placement
// of the two OUTPUT planes pinned: every geometry below is timed in ONE process on a verified-fast AND on a
// verified-slow pair of output planes (profiles/r03_notes.md section 22: a pair of planes is on one of two levels, 18 %
// apart, for as long as the allocations live).  VERDICT r03 item 1: round 3's geometry sweeps predate that finding.
//
//   pairs      allocate candidate planes, probe each against plane 0 with the product shape, keep a fast and a slow pair
//   geometry   loci per block 16 ... 4000, resident workgroups per CU 1 ... 5 (persistent, contiguous range or strided),
//              locus-major launch, flat one-chunk-per-thread -- on both pairs
//   one-stream the two outputs as ONE allocation: row-interleaved, 1 KB-tile-interleaved, 16 B-interleaved, halves
//   policy     store policy per output stream (nt / plain mixes)
//   vmm        planes built from hipMemCreate chunks (one handle per plane, alternating chunks of a common pool)
// Build: hipcc --offload-arch=gfx950 -O3 -o pinned_probe pinned_probe.hip ; run: ./pinned_probe [L] [S] [reps] [sections]
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

// output addressing.  OUT 0: two planes [L][S4].  1: one buffer, rows interleaved [L][2][S4].  2: one buffer, a
// wave's kilobyte of each output side by side [L][S4/64][2][64].  3: 16-byte chunks interleaved [L][S4][2].
template <int OUT>
__device__ __forceinline__ void put(const Streams& s, int l, int c, int S4, u32x4 r0, u32x4 r1) {
    if (OUT == 0) {
        const size_t o = (size_t)l * S4 + c;
        __builtin_nontemporal_store(r0, s.out[0] + o);
        __builtin_nontemporal_store(r1, s.out[1] + o);
    } else if (OUT == 1) {
        const size_t o = (size_t)l * 2 * S4 + c;
        __builtin_nontemporal_store(r0, s.out[0] + o);
        __builtin_nontemporal_store(r1, s.out[0] + o + S4);
    } else if (OUT == 2) {
        const size_t o = ((size_t)l * S4 + (c & ~63)) * 2 + (c & 63);
        __builtin_nontemporal_store(r0, s.out[0] + o);
        __builtin_nontemporal_store(r1, s.out[0] + o + 64);
    } else {
        const size_t o = ((size_t)l * S4 + c) * 2;
        __builtin_nontemporal_store(r0, s.out[0] + o);
        __builtin_nontemporal_store(r1, s.out[0] + o + 1);
    }
}

// column-owner walk.  ASSIGN 0: workgroup row `by` takes the contiguous range [by * lpb, (by + 1) * lpb);
// 1: strided, loci by, by + gy, ...; 2: contiguous range, but the walk starts `rot` rows into it and wraps (the
// workgroups of one launch round do not touch the same row offset at the same time).  XY: locus-major launch order.
template <int OUT, int ASSIGN, int XY>
__global__ __launch_bounds__(256) void k_walk(Streams s, int L, int S4, int lpb) {
    extern __shared__ uint32_t dummy[];
    const int bx = XY ? blockIdx.y : blockIdx.x, by = XY ? blockIdx.x : blockIdx.y;
    const int gy = XY ? gridDim.x : gridDim.y;
    const int c = bx * 256 + threadIdx.x;
    if (c >= S4) return;
    if (ASSIGN == 1) {
        for (int l = by; l < L; l += gy) {
            const size_t o = (size_t)l * S4 + c;
            u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
            r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
            put<OUT>(s, l, c, S4, r, r + 1u);
        }
    } else {
        const int l0 = by * lpb, l1 = min(L, l0 + lpb), n = l1 - l0;
        int l = l0;
        if (ASSIGN == 2 && n > 0) l = l0 + (int)(((unsigned)by * 2654435761u + (unsigned)bx * 40503u) % (unsigned)n);
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)l * S4 + c;
            u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
            r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
            put<OUT>(s, l, c, S4, r, r + 1u);
            if (++l == l1) l = l0;
        }
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// XCD-aware 1-D launch (block b runs on XCD b % 8, observed): the gx column tiles of one locus range get consecutive
// slots of ONE XCD, so that an XCD streams whole rows of a few ranges instead of 4 KB pieces of every range.
// range = xcd + 8 * (slot / gx), column tile = slot % gx; n_ranges is rounded up to a multiple of 8 by the launcher.
template <int OUT>
__global__ __launch_bounds__(256) void k_walk_xcd(Streams s, int L, int S4, int lpb, int gx_) {
    extern __shared__ uint32_t dummy[];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int by = xcd + 8 * (slot / gx_), bx = slot % gx_;
    const int c = bx * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = by * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
        put<OUT>(s, l, c, S4, r, r + 1u);
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// XCD-aware, K walkers per range: the K workgroups of a (range, tile) take loci l0 + k, l0 + k + K, ... -- the range's
// window is K adjacent rows per stream instead of one, and there are K times fewer ranges (distinct places in memory
// that are streamed at the same time) for the same number of resident workgroups.
__global__ __launch_bounds__(256) void k_walk_xcdk(Streams s, int L, int S4, int lpr, int gx_, int K) {
    extern __shared__ uint32_t dummy[];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int per = gx_ * K;
    const int range = xcd + 8 * (slot / per), rem = slot % per, k = rem / gx_, bx = rem % gx_;
    const int c = bx * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = range * lpr, l1 = min(L, l0 + lpr);
    for (int l = l0 + k; l < l1; l += K) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
        __builtin_nontemporal_store(r, s.out[0] + o);
        __builtin_nontemporal_store(r + 1u, s.out[1] + o);
    }
    if (dummy && lpr < 0) dummy[threadIdx.x] = 0;
}

// store policy per output stream: P 0 nt, 1 plain, 2 sc1, 3 sc0 sc1
template <int P>
__device__ __forceinline__ void st_pol(u32x4* p, u32x4 v) {
    if (P == 0) __builtin_nontemporal_store(v, p);
    else if (P == 1) *p = v;
    else if (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int P0, int P1>
__global__ __launch_bounds__(256) void k_pol(Streams s, int L, int S4, int lpb) {
    extern __shared__ uint32_t dummy[];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
        st_pol<P0>(s.out[0] + o, r);
        st_pol<P1>(s.out[1] + o, r + 1u);
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// NIN inputs, NOUT outputs of the same walk (read-only / write-only / partial mixes)
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_mix(Streams s, int L, int S4, int lpb, uint32_t* sink) {
    extern __shared__ uint32_t dummy[];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    u32x4 acc = {0, 0, 0, 0};
    for (int l = l0; l < l1; ++l) {
        const size_t o = (size_t)l * S4 + c;
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < NIN; ++k) r |= __builtin_nontemporal_load(s.in[k] + o);
        if (NOUT == 0) acc ^= r;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) __builtin_nontemporal_store(r + (uint32_t)k, s.out[k] + o);
    }
    if (NOUT == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

template <int OUT>
__global__ __launch_bounds__(256) void k_flat(Streams s, int L, int S4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)L * S4;
    if (i >= n) return;
    u32x4 r = {(uint32_t)i, 1u, 2u, 3u};
    r |= __builtin_nontemporal_load(s.in[0] + i) | __builtin_nontemporal_load(s.in[1] + i) | __builtin_nontemporal_load(s.in[2] + i);
    if (OUT == 0) {
        __builtin_nontemporal_store(r, s.out[0] + i);
        __builtin_nontemporal_store(r + 1u, s.out[1] + i);
    } else {
        const int l = (int)(i / S4), c = (int)(i - (size_t)l * S4);
        put<OUT>(s, l, c, S4, r, r + 1u);
    }
}


// ---- round 5: temporal decoupling of the two write streams ----
// XCD-aware persistent column-owner walk (the product's map 2); the outputs of K loci are held in registers and leave as
// bursts.  MODE 0: A at once, B held K loci (A A A A | B B B B interleaved only at burst boundaries); 1: both held, K
// stores to A then K stores to B; 2: as 1 with an s_waitcnt vmcnt(0) between the two bursts (A's stores have left the
// wave before B's start).
template <int K, int MODE>
__global__ __launch_bounds__(256) void k_walk_burst(Streams s, int L, int S4, int lpb, int gx_) {
    extern __shared__ uint32_t dummy[];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int by = xcd + 8 * (slot / gx_), bx = slot % gx_;
    const int c = bx * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = by * lpb, l1 = min(L, l0 + lpb);
    for (int lb = l0; lb < l1; lb += K) {
        u32x4 ha[K], hb[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int l = lb + i;
            if (l < l1) {
                const size_t o = (size_t)l * S4 + c;
                u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
                r |= __builtin_nontemporal_load(s.in[0] + o) | __builtin_nontemporal_load(s.in[1] + o) | __builtin_nontemporal_load(s.in[2] + o);
                if (MODE == 0) __builtin_nontemporal_store(r, s.out[0] + o);
                else ha[i] = r;
                hb[i] = r + 1u;
            }
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < K; ++i)
                if (lb + i < l1) __builtin_nontemporal_store(ha[i], s.out[0] + (size_t)(lb + i) * S4 + c);
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (lb + i < l1) __builtin_nontemporal_store(hb[i], s.out[1] + (size_t)(lb + i) * S4 + c);
    }
    if (dummy && lpb < 0) dummy[threadIdx.x] = 0;
}

// row-owner walk: a workgroup of T threads covers WHOLE rows (CH chunks per thread, chunk = tid + i * T) and walks a
// contiguous range of loci: 40 KB contiguous per locus and plane leave one workgroup.  ORDER 0: A, B per chunk;
// 1: the row's A stores, then its B stores; 2: K rows of A, then K rows of B (K = 2).
template <int T, int CH, int ORDER>
__global__ __launch_bounds__(T) void k_row(Streams s, int L, int S4, int lpw) {
    const int l0 = blockIdx.x * lpw, l1 = min(L, l0 + lpw);
    for (int l = l0; l < l1; ++l) {
        u32x4 r[CH];
        const size_t ro = (size_t)l * S4;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * T;
            r[i] = (u32x4){(uint32_t)l, 1u, 2u, 3u};
            if (c < S4) r[i] |= __builtin_nontemporal_load(s.in[0] + ro + c) | __builtin_nontemporal_load(s.in[1] + ro + c) | __builtin_nontemporal_load(s.in[2] + ro + c);
        }
        if (ORDER == 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = threadIdx.x + i * T;
                if (c < S4) { __builtin_nontemporal_store(r[i], s.out[0] + ro + c); __builtin_nontemporal_store(r[i] + 1u, s.out[1] + ro + c); }
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = threadIdx.x + i * T;
                if (c < S4) __builtin_nontemporal_store(r[i], s.out[0] + ro + c);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = threadIdx.x + i * T;
                if (c < S4) __builtin_nontemporal_store(r[i] + 1u, s.out[1] + ro + c);
            }
        }
    }
}

// write-only, ONE plane per kernel (two of them run on two queues at the same time: is the pair effect a matter of two
// streams of one WAVE, or of any two streams into the same class at the same time?)
__global__ __launch_bounds__(256) void k_fill_walk(u32x4* out, int L, int S4, int lpb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= S4) return;
    const int l0 = blockIdx.y * lpb, l1 = min(L, l0 + lpb);
    for (int l = l0; l < l1; ++l) {
        u32x4 r = {(uint32_t)l, 1u, 2u, 3u};
        __builtin_nontemporal_store(r, out + (size_t)l * S4 + c);
    }
}

// ---- round 5: the compact pass's stream (3 x 16 B/lane in, ONE byte per call out: 4 B per lane and locus) ----
// MODE 0: the 4-byte store as it is (nt); 1: plain store; 2: the bytes of 4 loci exchanged inside the workgroup through
// LDS so that a wave stores 16 B per lane (one kilobyte row segment per wave instruction) every fourth locus;
// 3: no output at all (the read-only walk, same loop)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k_compact(Streams s, uint32_t* out8, int L, int S4, int lpb, int gx_, uint32_t* sink) {
    __shared__ uint32_t stage[4][256];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int by = xcd + 8 * (slot / gx_), bx = slot % gx_;
    const int c = bx * 256 + threadIdx.x;
    const bool live = c < S4;
    const int l0 = by * lpb, l1 = min(L, l0 + lpb);
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[DEPTH][3];
    auto fetch = [&](int l, u32x4 (&r)[3]) {
        const size_t o = (size_t)min(l, l1 - 1) * S4 + (live ? c : 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = __builtin_nontemporal_load(s.in[k] + o);
    };
    if (l0 >= l1) return;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(l0 + d, ring[d]);
    for (int l = l0; l < l1; l += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (l + d >= l1) break;
            const u32x4 r = ring[d][0] | ring[d][1] | ring[d][2];
            fetch(l + d + DEPTH, ring[d]);
            const uint32_t m8 = (r.x & 0xff) | ((r.y & 0xff) << 8) | ((r.z & 0xff) << 16) | (r.w << 24);
            const size_t o = (size_t)(l + d) * S4 + c;
            if (MODE == 0) { if (live) __builtin_nontemporal_store(m8, out8 + o); }
            else if (MODE == 1) { if (live) out8[o] = m8; }
            else if (MODE == 2) {
                const int q = (l + d - l0) & 3;
                stage[q][threadIdx.x] = m8;
                if (q == 3 || l + d == l1 - 1) {
                    __syncthreads();
                    // wave w stores locus (l + d - q + w)'s kilobyte: lane i the 16 bytes of threads 4i .. 4i + 3
                    const int w = threadIdx.x >> 6, i = threadIdx.x & 63;
                    if (w <= q) {
                        const u32x4 v = {stage[w][4 * i], stage[w][4 * i + 1], stage[w][4 * i + 2], stage[w][4 * i + 3]};
                        const int cc = bx * 256 + 4 * i;
                        if (cc + 3 < S4) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out8 + (size_t)(l + d - q + w) * S4 + cc));
                        else for (int k = 0; k < 4; ++k) if (cc + k < S4) out8[(size_t)(l + d - q + w) * S4 + cc + k] = v[k];
                    }
                    __syncthreads();
                }
            } else if (MODE == 4) {
                // round 6: TILED mask layout [column tile][locus][256 lanes]: a workgroup's kilobytes of consecutive
                // loci lie back to back (one contiguous ~1 MB stream per workgroup instead of a kilobyte per 10 KB row)
                if (live) __builtin_nontemporal_store(m8, out8 + ((size_t)bx * L + (l + d)) * 256 + threadIdx.x);
            } else if (MODE == 5) {
                // tiled per WAVE: [column tile][wave][locus][64 lanes] -- each wave's 256 B of consecutive loci back to back
                if (live) __builtin_nontemporal_store(m8, out8 + (((size_t)bx * 4 + (threadIdx.x >> 6)) * L + (l + d)) * 64 + (threadIdx.x & 63));
            } else if (MODE == 6) {
                // tiled per wave + the bytes of 4 loci exchanged inside the WAVE (no barrier): lane i stores 16 B = the dwords
                // of lanes 4(i & 15) .. + 3 of locus (i >> 4): one contiguous kilobyte per wave every fourth locus
                const int q = (l + d - l0) & 3;
                const int w = threadIdx.x >> 6, i = threadIdx.x & 63;
                stage[q][threadIdx.x] = m8;
                if (q == 3 || l + d == l1 - 1) {
                    const int qq = i >> 4, r4 = 4 * (i & 15);
                    const u32x4 v = {stage[qq][64 * w + r4], stage[qq][64 * w + r4 + 1], stage[qq][64 * w + r4 + 2], stage[qq][64 * w + r4 + 3]};
                    if (qq <= q)
                        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out8 + (((size_t)bx * 4 + w) * L + (l + d - q + qq)) * 64 + r4));
                }
            } else acc ^= r;
        }
    }
    if (MODE == 3 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

struct Result { std::string name; double mn, avg; };
static std::vector<Result> results;
static hipEvent_t e0, e1;
static int reps = 5;

template <typename F>
static double run(const char* name, F launch, bool quiet = false) {
    launch();
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
    if (!quiet) {
        results.push_back({name, mn, sum / reps});
        printf("%-86s min %7.3f  avg %7.3f ms\n", name, mn, sum / reps);
        fflush(stdout);
    }
    return sum / reps;
}

static int ncu, gx, L, S4;
static size_t plane;
static uint32_t* sink;

static int lpb_product() {   // the product's grid rule: whole rounds of 5 workgroups per CU, <= 117 loci per block
    const long slots = (long)ncu * 5;
    int lpb = 117;
    const long min_wgs = (long)gx * ((L + lpb - 1) / lpb);
    long k = (min_wgs + slots - 1) / slots;
    if (k < 2) k = 2;
    long gyr = k * slots / gx;
    int lpb2 = (int)((L + gyr - 1) / gyr);
    return lpb2 < lpb ? lpb2 : lpb;
}
static size_t lds_for(int wgcu) {   // dynamic LDS that admits exactly wgcu workgroups per CU (160 KB per CU)
    if (wgcu >= 8) return 0;
    return std::min<size_t>(64 * 1024, (size_t)(160 * 1024 / wgcu - 2048));
}
static double product_shape(const Streams& s, const char* name, bool quiet = false) {
    const int lpb = lpb_product(), gy = (L + lpb - 1) / lpb;
    return run(name, [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb); }, quiet);
}

static void geometry(const Streams& s, const char* tag) {
    char nm[200];
    snprintf(nm, sizeof nm, "[%s] product shape (%d loci/block, 5 WG/CU)", tag, lpb_product());
    product_shape(s, nm);
    for (int lpb : {16, 32, 56, 112, 250, 500, 1000, 2000, 4000}) {
        const int gy = (L + lpb - 1) / lpb;
        snprintf(nm, sizeof nm, "[%s] contiguous %4d loci/block, <= 5 WG/CU (gy %d = %.2f rounds)", tag, lpb, gy, (double)gx * gy / (ncu * 5.0));
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb); });
    }
    for (int wgcu : {1, 2, 3, 4, 5, 8}) {
        const int gy = std::max(1, wgcu * ncu / gx), lpb = (L + gy - 1) / gy;
        snprintf(nm, sizeof nm, "[%s] persistent %d WG/CU contiguous range (gy %d, %d loci/WG)", tag, wgcu, gy, lpb);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb); });
        snprintf(nm, sizeof nm, "[%s] persistent %d WG/CU contiguous range, rotated start", tag, wgcu);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 2, 0>), dim3(gx, gy), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb); });
        snprintf(nm, sizeof nm, "[%s] persistent %d WG/CU strided (window of %d rows)", tag, wgcu, gy);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 1, 0>), dim3(gx, gy), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb); });
        snprintf(nm, sizeof nm, "[%s] persistent %d WG/CU contiguous range, locus-major launch", tag, wgcu);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 0, 1>), dim3(gy, gx), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb); });
    }
    // two whole rounds at low occupancy (a round's prologue / flush overlap with the other round's stream in the product)
    for (int wgcu : {2, 3}) {
        const int gy = std::max(1, 2 * wgcu * ncu / gx), lpb = (L + gy - 1) / gy;
        snprintf(nm, sizeof nm, "[%s] two rounds of %d WG/CU contiguous (gy %d, %d loci/WG)", tag, wgcu, gy, lpb);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb); });
    }
    for (int wgcu : {2, 3, 4, 5, 8}) {
        const int ny = std::max(8, wgcu * ncu / gx / 8 * 8), lpb = (L + ny - 1) / ny;
        snprintf(nm, sizeof nm, "[%s] XCD-aware persistent %d WG/CU (%d ranges of %d loci)", tag, wgcu, ny, lpb);
        run(nm, [&] { hipLaunchKernelGGL((k_walk_xcd<0>), dim3(ny * gx), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb, gx); });
    }
    for (int wgcu : {4, 5, 8})
        for (int K : {2, 4, 8, 16}) {
            int nr = wgcu * ncu / (gx * K) / 8 * 8;
            if (nr < 8) nr = 8;
            const int lpr = (L + nr - 1) / nr;
            snprintf(nm, sizeof nm, "[%s] XCD-aware persistent %d WG/CU, %d walkers per range (%d ranges of %d loci)", tag, wgcu, K, nr, lpr);
            run(nm, [&] { hipLaunchKernelGGL(k_walk_xcdk, dim3(nr * gx * K), dim3(256), lds_for(wgcu), 0, s, L, S4, lpr, gx, K); });
        }
    for (int lpb : {56, 112, 250, 500}) {
        const int ny = ((L + lpb - 1) / lpb + 7) / 8 * 8;
        snprintf(nm, sizeof nm, "[%s] XCD-aware %d loci/block, <= 5 WG/CU (%d ranges = %.2f rounds)", tag, lpb, ny, (double)gx * ny / (ncu * 5.0));
        run(nm, [&] { hipLaunchKernelGGL((k_walk_xcd<0>), dim3(ny * gx), dim3(256), lds_for(5), 0, s, L, S4, lpb, gx); });
    }
    for (int lpb : {32, 112}) {
        const int gy = (L + lpb - 1) / lpb;
        snprintf(nm, sizeof nm, "[%s] contiguous %4d loci/block, locus-major launch, 5 WG/CU", tag, lpb);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<0, 0, 1>), dim3(gy, gx), dim3(256), lds_for(5), 0, s, L, S4, lpb); });
    }
    snprintf(nm, sizeof nm, "[%s] flat, one chunk per thread", tag);
    const unsigned gflat = (unsigned)(((size_t)L * S4 + 255) / 256);
    run(nm, [&] { hipLaunchKernelGGL((k_flat<0>), dim3(gflat), dim3(256), 0, 0, s, L, S4); });
    const int lpb = lpb_product(), gy = (L + lpb - 1) / lpb;
    snprintf(nm, sizeof nm, "[%s] write only (2 out), product grid", tag);
    run(nm, [&] { hipLaunchKernelGGL((k_mix<0, 2>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb, sink); });
    snprintf(nm, sizeof nm, "[%s] 1 in / 2 out, product grid", tag);
    run(nm, [&] { hipLaunchKernelGGL((k_mix<1, 2>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb, sink); });
    snprintf(nm, sizeof nm, "[%s] 3 in / 1 out, product grid", tag);
    run(nm, [&] { hipLaunchKernelGGL((k_mix<3, 1>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb, sink); });
    snprintf(nm, sizeof nm, "[%s] read only (3 in), product grid", tag);
    run(nm, [&] { hipLaunchKernelGGL((k_mix<3, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb, sink); });
#define PL(P0, P1, NAME)                                                                                                   \
    snprintf(nm, sizeof nm, "[%s] store policy %s, product grid", tag, NAME);                                              \
    run(nm, [&] { hipLaunchKernelGGL((k_pol<P0, P1>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb); });
    PL(0, 0, "nt / nt") PL(0, 1, "nt / plain") PL(1, 0, "plain / nt") PL(0, 2, "nt / sc1") PL(0, 3, "nt / sc0 sc1") PL(1, 1, "plain / plain")
#undef PL
}


static void burst(const Streams& s, const char* tag) {
    char nm[200];
    const int wgcu = 4;
    const int ny = std::max(8, wgcu * ncu / gx / 8 * 8), lpb = (L + ny - 1) / ny;
    snprintf(nm, sizeof nm, "[%s] XCD-aware persistent 4 WG/CU, no burst (reference)", tag);
    run(nm, [&] { hipLaunchKernelGGL((k_walk_xcd<0>), dim3(ny * gx), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb, gx); });
#define BU(K, M, NAME)                                                                                                     \
    snprintf(nm, sizeof nm, "[%s] burst K = %d, %s", tag, K, NAME);                                                        \
    run(nm, [&] { hipLaunchKernelGGL((k_walk_burst<K, M>), dim3(ny * gx), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb, gx); });
    BU(2, 0, "A at once, B held") BU(4, 0, "A at once, B held") BU(8, 0, "A at once, B held") BU(16, 0, "A at once, B held")
    BU(2, 1, "K x A then K x B") BU(4, 1, "K x A then K x B") BU(8, 1, "K x A then K x B")
    BU(4, 2, "K x A, wait, K x B") BU(8, 2, "K x A, wait, K x B")
#undef BU
#define RO(T, CH, O, NAME)                                                                                                 \
    if ((T) * (CH) >= S4) for (int per : {1, 2, 4}) {                                                                      \
        const int nw = per * ncu, lpw = (L + nw - 1) / nw;                                                                 \
        snprintf(nm, sizeof nm, "[%s] row-owner %d threads x %d chunks, %d WG/CU, %s", tag, T, CH, per, NAME);             \
        run(nm, [&] { hipLaunchKernelGGL((k_row<T, CH, O>), dim3(nw), dim3(T), 0, 0, s, L, S4, lpw); });                   \
    }
    RO(1024, 3, 0, "A B per chunk") RO(1024, 3, 1, "row of A then row of B")
    RO(512, 5, 0, "A B per chunk") RO(512, 5, 1, "row of A then row of B")
    RO(256, 10, 0, "A B per chunk") RO(256, 10, 1, "row of A then row of B")
#undef RO
    // two write-only kernels, one plane each, at the same time on two queues; and one after the other
    static hipStream_t q1 = nullptr;
    static hipEvent_t ef = nullptr, eb = nullptr;
    if (!q1) { CK(hipStreamCreateWithFlags(&q1, hipStreamNonBlocking)); CK(hipEventCreate(&ef)); CK(hipEventCreate(&eb)); }
    const int lpbp = lpb_product(), gyp = (L + lpbp - 1) / lpbp;
    snprintf(nm, sizeof nm, "[%s] write only, plane A then plane B (two launches, one queue)", tag);
    run(nm, [&] {
        hipLaunchKernelGGL(k_fill_walk, dim3(gx, gyp), dim3(256), 0, 0, s.out[0], L, S4, lpbp);
        hipLaunchKernelGGL(k_fill_walk, dim3(gx, gyp), dim3(256), 0, 0, s.out[1], L, S4, lpbp);
    });
    snprintf(nm, sizeof nm, "[%s] write only, plane A and plane B at the same time (two queues)", tag);
    run(nm, [&] {
        CK(hipEventRecord(ef, 0));
        CK(hipStreamWaitEvent(q1, ef, 0));
        hipLaunchKernelGGL(k_fill_walk, dim3(gx, gyp), dim3(256), 0, 0, s.out[0], L, S4, lpbp);
        hipLaunchKernelGGL(k_fill_walk, dim3(gx, gyp), dim3(256), 0, q1, s.out[1], L, S4, lpbp);
        CK(hipEventRecord(eb, q1));
        CK(hipStreamWaitEvent(0, eb, 0));
    });
    snprintf(nm, sizeof nm, "[%s] write only, both planes from one kernel (product grid)", tag);
    run(nm, [&] { hipLaunchKernelGGL((k_mix<0, 2>), dim3(gx, gyp), dim3(256), lds_for(5), 0, s, L, S4, lpbp, sink); });
}

static void compact(const Streams& s, uint32_t* out8) {
    char nm[200];
    for (int wgcu : {4, 5, 8}) {
        const int ny = std::max(8, wgcu * ncu / gx / 8 * 8), lpb = (L + ny - 1) / ny;
#define CO(M, D, NAME)                                                                                                     \
        snprintf(nm, sizeof nm, "[compact] %d WG/CU, %d loci in flight, %s", wgcu, D, NAME);                               \
        run(nm, [&] { hipLaunchKernelGGL((k_compact<M, D>), dim3(ny * gx), dim3(256), 0, 0, s, out8, L, S4, lpb, gx, sink); });
        CO(3, 1, "read only") CO(3, 3, "read only") CO(0, 1, "4 B nt store") CO(0, 2, "4 B nt store") CO(0, 3, "4 B nt store")
        CO(1, 1, "4 B plain store") CO(1, 3, "4 B plain store") CO(2, 1, "16 B stores through LDS every 4th locus") CO(2, 3, "16 B stores through LDS every 4th locus")
        CO(4, 1, "TILED [tile][locus][256], 4 B nt store") CO(4, 2, "TILED [tile][locus][256], 4 B nt store") CO(4, 3, "TILED [tile][locus][256], 4 B nt store")
        CO(5, 1, "TILED per wave [tile][wave][locus][64], 4 B nt store") CO(5, 3, "TILED per wave [tile][wave][locus][64], 4 B nt store")
        CO(6, 1, "TILED per wave, 16 B stores (4 loci exchanged inside the wave)") CO(6, 3, "TILED per wave, 16 B stores (4 loci exchanged inside the wave)")
#undef CO
    }
}

template <int OUT>
static void one_stream_runs(const Streams& s, const char* tag, const char* layout) {
    char nm[200];
    const int lpb = lpb_product(), gy = (L + lpb - 1) / lpb;
    snprintf(nm, sizeof nm, "[%s] %s: product shape", tag, layout);
    run(nm, [&] { hipLaunchKernelGGL((k_walk<OUT, 0, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb); });
    for (int wgcu : {2, 3}) {
        const int gy2 = std::max(1, wgcu * ncu / gx), lpb2 = (L + gy2 - 1) / gy2;
        snprintf(nm, sizeof nm, "[%s] %s: persistent %d WG/CU contiguous range", tag, layout, wgcu);
        run(nm, [&] { hipLaunchKernelGGL((k_walk<OUT, 0, 0>), dim3(gx, gy2), dim3(256), lds_for(wgcu), 0, s, L, S4, lpb2); });
    }
    snprintf(nm, sizeof nm, "[%s] %s: flat, one chunk per thread", tag, layout);
    const unsigned gflat = (unsigned)(((size_t)L * S4 + 255) / 256);
    run(nm, [&] { hipLaunchKernelGGL((k_flat<OUT>), dim3(gflat), dim3(256), 0, 0, s, L, S4); });
}

// ---- VMM: a plane as a virtual range backed by physical chunks the program created itself ----
struct VmmPool {
    std::vector<hipMemGenericAllocationHandle_t> h;
    size_t chunk = 0;
};
static bool vmm_ok = true;
static bool vmm_create(VmmPool& p, size_t chunk, int n) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    p.chunk = chunk;
    for (int i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t hd;
        hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
        if (e != hipSuccess) { printf("# hipMemCreate(%zu) failed: %s\n", chunk, hipGetErrorString(e)); vmm_ok = false; return false; }
        p.h.push_back(hd);
    }
    return true;
}
static void* vmm_map(const VmmPool& p, const std::vector<int>& which) {
    void* va = nullptr;
    const size_t sz = p.chunk * which.size();
    if (hipMemAddressReserve(&va, sz, 0, nullptr, 0) != hipSuccess) { vmm_ok = false; return nullptr; }
    for (size_t i = 0; i < which.size(); ++i)
        if (hipMemMap((char*)va + i * p.chunk, p.chunk, 0, p.h[which[i]], 0) != hipSuccess) { printf("# hipMemMap failed\n"); vmm_ok = false; return nullptr; }
    hipMemAccessDesc ad = {};
    ad.location.type = hipMemLocationTypeDevice;
    ad.location.id = 0;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, sz, &ad, 1) != hipSuccess) { printf("# hipMemSetAccess failed\n"); vmm_ok = false; return nullptr; }
    return va;
}

int main(int argc, char** argv) {
    L = argc > 1 ? atoi(argv[1]) : 100000;
    const int S = argc > 2 ? atoi(argv[2]) : 10016;
    if (argc > 3) reps = atoi(argv[3]);
    const char* sections = argc > 4 ? argv[4] : "pairs,geometry,onestream,vmm";
    auto want = [&](const char* k) { return strstr(sections, k) != nullptr; };
    S4 = S / 4;
    plane = (size_t)L * S4 * 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    gx = (S4 + 255) / 256;
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("# device %s, %d CUs; L = %d, S = %d, plane = %.3f GB, reps = %d; free %.1f of %.1f GB\n", prop.name, ncu, L, S,
           plane * 1e-9, reps, fr * 1e-9, tot * 1e-9);
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc(&sink, 64));
    Streams s;
    void* in[3];
    for (int k = 0; k < 3; ++k) {
        CK(hipMalloc(&in[k], plane + 256));
        CK(hipMemset(in[k], k + 1, plane));
        s.in[k] = (const u32x4*)in[k];
    }
    // ---- pairs: candidate output planes, each probed with plane 0 (and a few further pairs) ----
    std::vector<void*> cand;
    std::vector<void*> spacers;
    int fast_b = -1, slow_b = -1;
    double fast_ms = 1e30, slow_ms = 0;
    const int max_cand = getenv("NCAND") ? atoi(getenv("NCAND")) : 10;
    for (int k = 0; k < max_cand; ++k) {
        void* p;
        CK(hipMalloc(&p, plane + 256));
        CK(hipMemset(p, 7, plane));
        cand.push_back(p);
        if (k == 0) continue;
        s.out[0] = (u32x4*)cand[0];
        s.out[1] = (u32x4*)p;
        char nm[160];
        snprintf(nm, sizeof nm, "pairs: plane 0 (%p) + plane %d (%p), product shape", cand[0], k, p);
        const double ms = product_shape(s, nm);
        if (ms < fast_ms) { fast_ms = ms; fast_b = k; }
        if (ms > slow_ms) { slow_ms = ms; slow_b = k; }
        if (k >= 3 && slow_ms > 1.08 * fast_ms) break;
        if (k >= 4 && (k % 2) == 0) {   // still one level: jump 16 GB ahead (the tuner's move)
            void* sp = nullptr;
            if (hipMalloc(&sp, (size_t)16 << 30) == hipSuccess) spacers.push_back(sp);
        }
    }
    for (void* sp : spacers) hipFree(sp);
    const bool two_levels = slow_ms > 1.08 * fast_ms;
    printf("# fast pair: 0 + %d (%.3f ms); slow pair: 0 + %d (%.3f ms); %s\n", fast_b, fast_ms, slow_b, slow_ms,
           two_levels ? "two levels seen" : "ONE level only");
    if (want("pairs") && cand.size() >= 4) {   // the class structure among the first few planes
        const int n = (int)std::min<size_t>(cand.size(), 6);
        printf("# pair matrix (ms), rows = out0, columns = out1\n");
        for (int i = 0; i < n; ++i) {
            printf("#  ");
            for (int j = 0; j < n; ++j) {
                if (i == j) { printf("   -   "); continue; }
                s.out[0] = (u32x4*)cand[i];
                s.out[1] = (u32x4*)cand[j];
                printf(" %6.3f", product_shape(s, "", true));
            }
            printf("\n");
        }
    }
    Streams sf = s, ss = s;
    sf.out[0] = (u32x4*)cand[0]; sf.out[1] = (u32x4*)cand[fast_b];
    ss.out[0] = (u32x4*)cand[0]; ss.out[1] = (u32x4*)cand[slow_b];
    if (want("geometry")) {
        geometry(sf, "fast pair");
        if (two_levels) geometry(ss, "slow pair");
    }
    if (want("compact")) {
        uint32_t* out8 = nullptr;
        CK(hipMalloc((void**)&out8, (size_t)gx * L * 1024 + 4096));     // (the tiled layouts: whole 256-lane tiles)
        compact(s, out8);
        hipFree(out8);
    }
    if (want("burst")) {
        burst(sf, "fast pair");
        if (two_levels) burst(ss, "slow pair");
    }
    if (want("memtype")) {   // the mask plane in memory of another type: does the pair effect follow the cache policy?
        for (unsigned fl : {(unsigned)hipDeviceMallocUncached, (unsigned)hipDeviceMallocFinegrained}) {
            for (int t = 0; t < 3; ++t) {
                void* p = nullptr;
                if (hipExtMallocWithFlags(&p, plane + 256, fl) != hipSuccess) { printf("# hipExtMallocWithFlags(%u) failed\n", fl); (void)hipGetLastError(); break; }
                CK(hipMemset(p, 7, plane));
                Streams o = s;
                o.out[0] = (u32x4*)cand[0];
                o.out[1] = (u32x4*)p;
                char nm[200];
                snprintf(nm, sizeof nm, "memtype: plane 0 + %s plane #%d (%p), product shape", fl == hipDeviceMallocUncached ? "UNCACHED" : "FINE-GRAINED", t, p);
                product_shape(o, nm);
                o.out[0] = (u32x4*)cand[slow_b];
                snprintf(nm, sizeof nm, "memtype: plane %d + %s plane #%d, product shape", slow_b, fl == hipDeviceMallocUncached ? "UNCACHED" : "FINE-GRAINED", t);
                product_shape(o, nm);
            }
        }
    }
    if (want("uc2")) {   // uncached planes among themselves, as inputs, as the masked-genotype plane; their read rate
        std::vector<void*> uc;
        for (int t = 0; t < 6; ++t) {
            void* p = nullptr;
            if (hipExtMallocWithFlags(&p, plane + 256, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemset(p, 7, plane));
            uc.push_back(p);
        }
        const int n = (int)uc.size();
        printf("# uc2: %d uncached planes; pair matrix (ms), rows = out0, columns = out1, product shape\n", n);
        for (int i = 0; i < n; ++i) {
            printf("#  ");
            for (int j = 0; j < n; ++j) {
                if (i == j) { printf("   -   "); continue; }
                Streams o = s;
                o.out[0] = (u32x4*)uc[i];
                o.out[1] = (u32x4*)uc[j];
                printf(" %6.3f", product_shape(o, "", true));
            }
            printf("\n");
        }
        printf("# uc2: out0 = cached candidate k, out1 = uncached plane 0 / 1\n#  ");
        for (size_t k = 0; k < cand.size(); ++k) {
            Streams o = s;
            o.out[0] = (u32x4*)cand[k];
            o.out[1] = (u32x4*)uc[0];
            printf(" %6.3f", product_shape(o, "", true));
            o.out[1] = (u32x4*)uc[1 % n];
            printf("/%6.3f", product_shape(o, "", true));
        }
        printf("\n# uc2: out0 = uncached plane 0, out1 = cached candidate k\n#  ");
        for (size_t k = 0; k < cand.size(); ++k) {
            Streams o = s;
            o.out[0] = (u32x4*)uc[0];
            o.out[1] = (u32x4*)cand[k];
            printf(" %6.3f", product_shape(o, "", true));
        }
        printf("\n");
        if (n >= 5) {
            char nm[200];
            Streams o = s;
            o.out[0] = (u32x4*)uc[0]; o.out[1] = (u32x4*)uc[1];
            const int lpb = lpb_product(), gy = (L + lpb - 1) / lpb;
            run("uc2: cached inputs, both outputs uncached, product shape", [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, o, L, S4, lpb); });
            run("uc2: same, write only", [&] { hipLaunchKernelGGL((k_mix<0, 2>), dim3(gx, gy), dim3(256), lds_for(5), 0, o, L, S4, lpb, sink); });
            const int ny = std::max(8, 4 * ncu / gx / 8 * 8), lpx = (L + ny - 1) / ny;
            run("uc2: same, XCD-aware persistent 4 WG/CU", [&] { hipLaunchKernelGGL((k_walk_xcd<0>), dim3(ny * gx), dim3(256), lds_for(4), 0, o, L, S4, lpx, gx); });
            const unsigned gflat = (unsigned)(((size_t)L * S4 + 255) / 256);
            run("uc2: same, flat", [&] { hipLaunchKernelGGL((k_flat<0>), dim3(gflat), dim3(256), 0, 0, o, L, S4); });
            Streams u = o;
            for (int k = 0; k < 3; ++k) u.in[k] = (const u32x4*)uc[2 + k];
            run("uc2: UNCACHED inputs too (5 uncached planes), product shape", [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, u, L, S4, lpb); });
            run("uc2: read only, 3 uncached inputs, product grid", [&] { hipLaunchKernelGGL((k_mix<3, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, u, L, S4, lpb, sink); });
            run("uc2: read only, 3 cached inputs, product grid", [&] { hipLaunchKernelGGL((k_mix<3, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, s, L, S4, lpb, sink); });
            Streams v = s;
            v.out[0] = (u32x4*)cand[0]; v.out[1] = (u32x4*)cand[slow_b];
            v.in[0] = (const u32x4*)uc[2];
            run("uc2: slow cached pair as outputs, input 0 uncached", [&] { hipLaunchKernelGGL((k_walk<0, 0, 0>), dim3(gx, gy), dim3(256), lds_for(5), 0, v, L, S4, lpb); });
            (void)nm;
        }
        for (void* p : uc) hipFree(p);
    }
    // free every candidate but the two pairs' planes before the big single allocations
    for (size_t k = 1; k < cand.size(); ++k)
        if ((int)k != fast_b && (int)k != slow_b) { hipFree(cand[k]); cand[k] = nullptr; }
    if (want("onestream")) {
        for (int trial = 0; trial < 3; ++trial) {
            void* big;
            if (hipMalloc(&big, 2 * plane + 4096) != hipSuccess) { printf("# 2-plane allocation failed\n"); break; }
            CK(hipMemset(big, 9, 2 * plane));
            Streams o = s;
            o.out[0] = (u32x4*)big;
            o.out[1] = (u32x4*)((char*)big + plane);
            char tag[64];
            snprintf(tag, sizeof tag, "one allocation #%d %p", trial, big);
            char nm[200];
            snprintf(nm, sizeof nm, "[%s] halves (out1 = out0 + plane): product shape", tag);
            product_shape(o, nm);
            one_stream_runs<1>(o, tag, "rows interleaved [L][2][S4]");
            one_stream_runs<2>(o, tag, "1 KB tiles interleaved [L][S4/64][2][64]");
            one_stream_runs<3>(o, tag, "16 B chunks interleaved [L][S4][2]");
            // keep the allocation (so that the next trial lands elsewhere) unless memory is short
            size_t f2 = 0, t2 = 0;
            CK(hipMemGetInfo(&f2, &t2));
            if (f2 < 4 * plane) hipFree(big);
        }
    }
    if (want("vmm")) {
        size_t gran = 0;
        hipMemAllocationProp mp = {};
        mp.type = hipMemAllocationTypePinned;
        mp.location.type = hipMemLocationTypeDevice;
        mp.location.id = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &mp, hipMemAllocationGranularityRecommended);
        printf("# vmm: recommended granularity %zu bytes (%s)\n", gran, hipGetErrorString(e));
        if (e == hipSuccess && gran > 0) {
            // (a) one handle per plane, six planes: do the two levels exist for VMM planes too?
            {
                const size_t psz = (plane + gran - 1) / gran * gran;
                VmmPool pool;
                if (vmm_create(pool, psz, 6)) {
                    std::vector<void*> pl;
                    for (int k = 0; k < 6 && vmm_ok; ++k) pl.push_back(vmm_map(pool, {k}));
                    if (vmm_ok) {
                        printf("# vmm (a): one handle per plane; pair matrix (ms)\n");
                        for (int i = 0; i < 6; ++i) {
                            printf("#  ");
                            for (int j = 0; j < 6; ++j) {
                                if (i == j) { printf("   -   "); continue; }
                                Streams o = s;
                                o.out[0] = (u32x4*)pl[i];
                                o.out[1] = (u32x4*)pl[j];
                                printf(" %6.3f", product_shape(o, "", true));
                            }
                            printf("\n");
                        }
                        // halves swapped: out1's first half backed by the second half of its handle?  (not possible with
                        // one handle per plane; see (b))
                    }
                    for (void* p : pl) if (p) { hipMemUnmap(p, psz); hipMemAddressFree(p, psz); }
                    for (auto hd : pool.h) hipMemRelease(hd);
                }
            }
            // (b) a common pool of chunks; planes A / B from alternating chunks, from the two halves of the pool, and
            // B with its chunk order reversed -- for chunk sizes 2 MB ... 1 GB
            for (size_t chunk : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
                if (chunk % gran) continue;
                vmm_ok = true;
                const int per = (int)((plane + chunk - 1) / chunk);
                if (chunk == ((size_t)2 << 20) && per > 4096) { }
                VmmPool pool;
                if (!vmm_create(pool, chunk, 2 * per)) continue;
                std::vector<int> even, odd, lo, hi, hi_rev;
                for (int i = 0; i < per; ++i) { even.push_back(2 * i); odd.push_back(2 * i + 1); lo.push_back(i); hi.push_back(per + i); }
                hi_rev = hi;
                std::reverse(hi_rev.begin(), hi_rev.end());
                struct Case { const char* name; std::vector<int>* a; std::vector<int>* b; };
                Case cases[] = {{"A = even chunks, B = odd chunks", &even, &odd}, {"A = first half, B = second half", &lo, &hi},
                                {"A = first half, B = second half reversed", &lo, &hi_rev}};
                for (auto& cs : cases) {
                    void* pa = vmm_map(pool, *cs.a);
                    void* pb = vmm_ok ? vmm_map(pool, *cs.b) : nullptr;
                    if (vmm_ok) {
                        Streams o = s;
                        o.out[0] = (u32x4*)pa;
                        o.out[1] = (u32x4*)pb;
                        char nm[200];
                        snprintf(nm, sizeof nm, "vmm (b) chunks of %zu MB: %s, product shape", chunk >> 20, cs.name);
                        product_shape(o, nm);
                    }
                    const size_t sz = chunk * per;
                    if (pa) { hipMemUnmap(pa, sz); hipMemAddressFree(pa, sz); }
                    if (pb) { hipMemUnmap(pb, sz); hipMemAddressFree(pb, sz); }
                }
                for (auto hd : pool.h) hipMemRelease(hd);
            }
        }
    }
    printf("JSON [");
    for (size_t i = 0; i < results.size(); ++i)
        printf("%s{\"name\": \"%s\", \"min_ms\": %.4f, \"avg_ms\": %.4f}", i ? ", " : "", results[i].name.c_str(), results[i].mn, results[i].avg);
    printf("]\n");
    return 0;
}
