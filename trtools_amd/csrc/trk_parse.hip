// trk_parse.hip -- the sample columns of VCF records parsed ON THE DEVICE (round 4, SURVEY 8(f1) widened: the text of a
// batch goes over PCIe once and the genotype tensor and the scalar FORMAT planes come into being in HBM, instead of
// being parsed by host threads -- 26 ns per call there, the bound of the 1 GB command lines on a 16-CPU grant,
// profiles/r04_notes.md section 13 -- and uploaded).
//
// What it replaces: the per-sample loop of parse_record (trk_vcf.cpp), i.e. cyvcf2's genotype.array() and
// format('DP') / format('Q') of the reference's record loop (tr_harmonizer.py:1420-1499, dumpSTR.py:613-700) for L
// records at once.  The grammar is the one-scan form's of parse_record: alleles '.' or up to four digits, '/' or '|';
// scalar Integer planes -?d{1,9} or '.'; scalar Float planes -?d*(.d*)? of at most fifteen digits, value w / 10^k in
// float64 (correctly rounded: Clinger's exact case; the same double strtod gives) cast to float32.  A token that is
// anything else raises the record's flag and the caller parses THAT record with the host code.
//
// One workgroup per record.  The sample region is walked in tiles of 16 KB: every lane loads four 16-byte chunks
// (coalesced), the tabs of a chunk are a 16-bit mask, their number goes to LDS, a workgroup scan turns the counts into
// the index of each chunk's first token, the lanes scatter their chunks' token starts into an LDS list, and then lane t
// parses tokens t, t + 256, ... of the tile straight from global memory (the bytes are in L2: the tile was just read).
// HBM traffic: the text once + the arrays once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trk.h"
#include "trk_internal.h"

namespace {

constexpr int PS_THREADS = 256;
constexpr int PS_TILE = 16384;                 // bytes of text per tile
constexpr int PS_CHUNKS = PS_TILE / 16;        // 1024 chunks, four per lane
constexpr int PS_MAXTOK = PS_TILE / 2 + 2;     // tokens that can START in a tile (one byte + a tab each)
constexpr int32_t PS_INT_MISSING = INT32_MIN;

struct ParseArgs {
    trk_parse_in in;
    trk_parse_out out;
};

__constant__ double c_p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};

__device__ __forceinline__ bool is_digit(unsigned char c) { return (unsigned)(c - '0') <= 9u; }

// bit i of the result: byte i of the 16-byte chunk is a tab
__device__ __forceinline__ uint32_t tab_mask(const uint4 v) {
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = w[k] ^ 0x09090909u;                       // a zero byte where the text has a tab
        // 0x80 in every zero byte and nowhere else (the carry-free form: (x - 0x01..) & ~x flags a 0x01 above a zero byte too)
        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
        // gather the four flag bits: bytes 0..3 -> bits 0..3
        m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * k);
    }
    return m;
}

__global__ __launch_bounds__(PS_THREADS) void k_parse_samples(const ParseArgs a) {
    __shared__ uint16_t s_cnt[PS_CHUNKS];          // tabs per chunk, then the index of the chunk's first token in the tile
    __shared__ uint32_t s_tok[PS_MAXTOK];          // start offsets (from the region's start) of the tokens of this tile
    __shared__ uint32_t s_wsum[PS_THREADS / 64];
    __shared__ uint32_t s_flags, s_maxpl, s_base;
    const int rec = blockIdx.x;
    const int tid = threadIdx.x;
    const int S = a.in.n_samples, P = a.in.ploidy;
    const unsigned char* text = a.in.text;
    const int64_t r0 = a.in.smp_off[rec], r1 = a.in.line_end[rec];
    const int gt_idx = a.in.gt_idx ? a.in.gt_idx[rec] : 0;
    int pidx[TRK_PARSE_MAX_PLANES];
    int max_needed = gt_idx;
#pragma unroll
    for (int i = 0; i < TRK_PARSE_MAX_PLANES; ++i) {
        pidx[i] = i < a.in.n_planes ? a.in.plane_idx[i][rec] : -1;
        max_needed = pidx[i] > max_needed ? pidx[i] : max_needed;
    }
    if (tid == 0) {
        s_flags = 0;
        s_maxpl = 1;
        s_base = 0;            // tokens that started in earlier tiles
    }
    __syncthreads();
    if (r1 <= r0 || max_needed < 0) {      // no sample columns, or neither GT nor a plane among the FORMAT keys: padding /
                                           // missing values, and the host has the last word on the record
        for (int s = tid; s < S; s += PS_THREADS) {
            for (int j = 0; j < P; ++j) a.out.gt[((int64_t)rec * S + s) * P + j] = -2;
            if (a.out.phased) a.out.phased[(int64_t)rec * S + s] = 0;
            for (int i = 0; i < a.in.n_planes; ++i) {
                if (a.in.plane_kind[i] == TRK_PARSE_FLOAT) static_cast<float*>(a.out.planes[i])[(int64_t)rec * S + s] = __builtin_nanf("");
                else static_cast<int32_t*>(a.out.planes[i])[(int64_t)rec * S + s] = PS_INT_MISSING;
            }
        }
        if (tid == 0) {
            a.out.locus_ploidy[rec] = 1;
            a.out.flags[rec] = S > 0 ? (r1 > r0 ? TRK_PARSE_HOST : TRK_PARSE_COLUMNS) : 0;
        }
        return;
    }
    const unsigned char* reg = text + r0;
    const int64_t n = r1 - r0;                           // bytes of the region; reg[n] is the newline
    // tiles start on 16-byte boundaries of the ABSOLUTE address (aligned vector loads)
    const int64_t lead = (int64_t)((uintptr_t)reg & 15u);
    const int64_t span = n + lead;
    for (int64_t t0 = 0; t0 < span; t0 += PS_TILE) {
        // ---- the tile's chunks: tab masks, counts ------------------------------------------------------------------
        uint32_t mk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid + q * PS_THREADS;              // chunk of the tile
            const int64_t o = t0 + (int64_t)c * 16 - lead;   // offset of the chunk's first byte from the region's start
            uint32_t m = 0;
            if (o < n && o + 16 > 0) {
                const uint4 v = *reinterpret_cast<const uint4*>(reg + o);
                m = tab_mask(v);
                if (o < 0) m &= ~((1u << (int)(-o)) - 1u);                    // bytes before the region
                if (o + 16 > n) m &= (1u << (int)(n - o)) - 1u;               // bytes from the newline on
            }
            mk[q] = m;
            s_cnt[c] = (uint16_t)__popc(m);
        }
        __syncthreads();
        // ---- exclusive scan of the 1024 counts: lane t owns chunks 4t .. 4t + 3 -----------------------------------
        uint32_t own[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            own[q] = s_cnt[4 * tid + q];
            sum += own[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((tid & 63) >= o) incl += y;
        }
        if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (tid >> 6); ++w) wbase += s_wsum[w];
        const uint32_t tile_tabs = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        uint32_t run = wbase + incl - sum;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s_cnt[4 * tid + q] = (uint16_t)run;
            run += own[q];
        }
        __syncthreads();
        // ---- token starts of the tile: the byte after every tab (+ the region's first token in the first tile) -----
        const uint32_t first = t0 == 0 ? 1u : 0u;
        if (t0 == 0 && tid == 0) s_tok[0] = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid + q * PS_THREADS;
            uint32_t m = mk[q];
            uint32_t k = first + s_cnt[c];
            const int64_t o = t0 + (int64_t)c * 16 - lead;
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                if (k < (uint32_t)PS_MAXTOK) s_tok[k] = (uint32_t)(o + b + 1);
                ++k;
            }
        }
        __syncthreads();
        const uint32_t base = s_base;
        const uint32_t ntok = first + tile_tabs;
        if (ntok > (uint32_t)PS_MAXTOK) {          // more one-byte tokens than any record has: the host's business (uniform)
            if (tid == 0) s_flags |= TRK_PARSE_HOST;
            __syncthreads();
            break;
        }
        // ---- parse: lane t takes tokens t, t + 256, ... of the tile ------------------------------------------------
        uint32_t lflags = 0, lmaxpl = 1;
        for (uint32_t k = tid; k < ntok; k += PS_THREADS) {
            const int64_t s = (int64_t)base + k;
            if (s >= S) {
                lflags |= TRK_PARSE_COLUMNS;
                continue;
            }
            const unsigned char* c = reg + s_tok[k];
            const unsigned char* const end = reg + n;
            int16_t* g = a.out.gt + ((int64_t)rec * S + s) * P;
            int j = 0;
            bool phased = false, ok = true;
            uint32_t got = 0;
            int kf = 0;
            for (;;) {
                if (kf == gt_idx) {
                    for (;;) {
                        int al;
                        if (*c == '.') {
                            al = -1;
                            ++c;
                        } else if (is_digit(*c)) {
                            const unsigned char* a0 = c;
                            al = 0;
                            do al = al * 10 + (*c++ - '0'); while (is_digit(*c));
                            if (c - a0 > 4) { ok = false; break; }
                        } else if (*c == '/' || *c == '|' || *c == ':' || *c == '\t' || c == end) {
                            al = -1;
                        } else {
                            ok = false;
                            break;
                        }
                        if (j >= P) { ok = false; lflags |= TRK_PARSE_PLOIDY; break; }
                        g[j++] = (int16_t)al;
                        if (*c == '|') phased = true;
                        else if (*c != '/') break;
                        ++c;
                    }
                    if (!ok) break;
                } else {
                    int pl = -1;
#pragma unroll
                    for (int i = 0; i < TRK_PARSE_MAX_PLANES; ++i) pl = pidx[i] == kf ? i : pl;
                    if (pl >= 0) {
                        const bool missing = *c == '.' && (c[1] == ':' || c[1] == '\t' || c + 1 == end);
                        if (a.in.plane_kind[pl] == TRK_PARSE_FLOAT) {
                            float x;
                            if (missing) {
                                x = __builtin_nanf("");
                                ++c;
                            } else {
                                const bool neg = *c == '-';
                                if (neg) ++c;
                                const unsigned char* a0 = c;
                                uint64_t w = 0;
                                while (is_digit(*c)) w = w * 10 + (uint64_t)(*c++ - '0');
                                int nd = (int)(c - a0), kd = 0;
                                if (*c == '.') {
                                    const unsigned char* f0 = ++c;
                                    while (is_digit(*c)) w = w * 10 + (uint64_t)(*c++ - '0');
                                    kd = (int)(c - f0);
                                    nd += kd;
                                }
                                if (nd < 1 || nd > 15) { ok = false; break; }
                                const double d = (double)w / c_p10[kd];
                                x = (float)(neg ? -d : d);
                            }
                            static_cast<float*>(a.out.planes[pl])[(int64_t)rec * S + s] = x;
                        } else {
                            int32_t x;
                            if (missing) {
                                x = PS_INT_MISSING;
                                ++c;
                            } else {
                                const bool neg = *c == '-';
                                if (neg) ++c;
                                if (!is_digit(*c)) { ok = false; break; }
                                const unsigned char* a0 = c;
                                uint32_t u = 0;
                                do u = u * 10 + (uint32_t)(*c++ - '0'); while (is_digit(*c));
                                if (c - a0 > 9) { ok = false; break; }
                                x = neg ? -(int32_t)u : (int32_t)u;
                            }
                            static_cast<int32_t*>(a.out.planes[pl])[(int64_t)rec * S + s] = x;
                        }
                        got |= 1u << pl;
                    } else {
                        while (*c != ':' && *c != '\t' && *c != '\n' && *c != '\r') ++c;
                    }
                }
                if (*c == ':') {
                    if (kf == max_needed) break;
                    ++c;
                    ++kf;
                    continue;
                }
                if (*c == '\t' || c == end) break;
                ok = false;
                break;
            }
            if (!ok) {
                lflags |= TRK_PARSE_HOST;
                continue;
            }
            for (int jj = j; jj < P; ++jj) g[jj] = -2;
            if (a.out.phased) a.out.phased[(int64_t)rec * S + s] = phased ? 1 : 0;
            lmaxpl = (uint32_t)j > lmaxpl ? (uint32_t)j : lmaxpl;
            for (int i = 0; i < a.in.n_planes; ++i)
                if (!((got >> i) & 1u)) {
                    if (a.in.plane_kind[i] == TRK_PARSE_FLOAT) static_cast<float*>(a.out.planes[i])[(int64_t)rec * S + s] = __builtin_nanf("");
                    else static_cast<int32_t*>(a.out.planes[i])[(int64_t)rec * S + s] = PS_INT_MISSING;
                }
        }
        if (lflags) atomicOr(&s_flags, lflags);
        if (lmaxpl > 1) atomicMax(&s_maxpl, lmaxpl);
        __syncthreads();
        if (tid == 0) s_base = base + ntok;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t f = s_flags;
        if ((int64_t)s_base != S) f |= TRK_PARSE_COLUMNS;      // fewer (or more) sample columns than the header announces
        a.out.flags[rec] = (uint8_t)f;
        a.out.locus_ploidy[rec] = (uint8_t)s_maxpl;
    }
}

}  // namespace

namespace trk {

hipError_t launch_parse_samples(const trk_parse_in& in, const trk_parse_out& out, hipStream_t stream) {
    if (in.n_records <= 0) return hipSuccess;
    ParseArgs a{in, out};
    hipLaunchKernelGGL(k_parse_samples, dim3((unsigned)in.n_records), dim3(PS_THREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace trk
