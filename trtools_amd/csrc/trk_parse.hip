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
// parses tokens t, t + 256, ... of the tile from an LDS copy of it (the chunks are stored there as they arrive, with
// 256 bytes beyond the tile and a closing tab: a token whose needed fields reach that tab is the host's).
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
constexpr int PS_OVER = 256;                   // k_parse_samples: text staged beyond a tile (its last token's needed fields end there)
// k_format_samples: tiles of 8 KB (two chunks per lane) -- text, token list and the staged output of a tile together stay
// under 40 KB of LDS, four workgroups per CU
constexpr int FS_TILE = 8192;
constexpr int FS_CHUNKS = FS_TILE / 16;        // 512 chunks, two per lane
constexpr int FS_Q = FS_CHUNKS / PS_THREADS;
constexpr int FS_MAXTOK = FS_TILE / 2 + 2;
constexpr int FS_OVER = 256;                   // text staged beyond the tile: the tile's last token ends there, or the host writes the record
constexpr int FS_STAGE = 20480;                // bytes of a tile's output staged in LDS by k_format_samples<true>

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
    __shared__ uint16_t s_tok[PS_MAXTOK];          // start offsets of the tokens of this tile, from the tile's first chunk
    // the tile's TEXT and PS_OVER bytes beyond it, closed by a tab: the lanes walk their tokens from LDS; a token whose
    // needed fields reach the closing tab is the host's
    __shared__ uint4 s_text[PS_CHUNKS + PS_OVER / 16 + 1];
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
            if (o <= n && o + 16 > 0) {                        // (the chunk that holds the newline too)
                const uint4 v = *reinterpret_cast<const uint4*>(reg + o);
                s_text[c] = v;
                m = tab_mask(v);
                if (o < 0) m &= ~((1u << (int)(-o)) - 1u);                    // bytes before the region
                if (o + 16 > n) m &= (1u << (int)(n - o)) - 1u;               // bytes from the newline on
            }
            mk[q] = m;
            s_cnt[c] = (uint16_t)__popc(m);
        }
        if (tid <= PS_OVER / 16) {
            const int c = PS_CHUNKS + tid;
            const int64_t o = t0 + (int64_t)c * 16 - lead;
            uint4 v = {0x09090909u, 0x09090909u, 0x09090909u, 0x09090909u};      // beyond the staged text: tabs
            if (tid < PS_OVER / 16 && o <= n) v = *reinterpret_cast<const uint4*>(reg + o);
            s_text[c] = v;
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
        if (t0 == 0 && tid == 0) s_tok[0] = (uint16_t)lead;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid + q * PS_THREADS;
            uint32_t m = mk[q];
            uint32_t k = first + s_cnt[c];
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                if (k < (uint32_t)PS_MAXTOK) s_tok[k] = (uint16_t)(c * 16 + b + 1);
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
            const unsigned char* const tb = reinterpret_cast<const unsigned char*>(s_text);
            const unsigned char* c = tb + s_tok[k];
            // the region's end (the newline) and the end of the staged text, as addresses in the tile's LDS copy
            const int64_t rel_end = span - t0;
            const unsigned char* const end = tb + (rel_end < (int64_t)(PS_TILE + PS_OVER) ? rel_end : (int64_t)(PS_TILE + PS_OVER + 16));
            const unsigned char* const lim = tb + PS_TILE + PS_OVER;
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
            if (!ok || (c >= lim && c != end)) {       // (the walk ran into the end of the staged text: a token that long is the host's)
                lflags |= TRK_PARSE_HOST;
                continue;
            }
            // (a token that ends before the record's GT key -- GT not the first key, trailing fields dropped -- is a
            // missing call, '.', as htslib fills a dropped field: tests/test_gpu_text_fuzz.py)
            const int j0 = (gt_idx >= 0 && j == 0) ? 1 : j;
            if (j0 > j) g[0] = -1;
            for (int jj = j0; jj < P; ++jj) g[jj] = -2;
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


// ---------------------------------------------------------------------------------------------------------------------
// k_format_samples : dumpSTR's sample columns WRITTEN on the device (round 4; the host form is fast_samples_scalar in
// trk_vcf.cpp, dumpSTR.py:648-683 and 715-746 in the reference): per sample a tab, then either the token as it stands
// (+ ':.' per FORMAT key it does not hold) + ':PASS' / ':NOCALL', or -- a filtered call -- the nulled token
// ('./.:.:.') + ':' + '<filter>_<value>,...'.  A kept token is copied only if decode -> format would print it back
// unchanged (the canonical-number rules of the host tier, checked in one forward scan); anything else flags the
// record and the host writer takes it.  Same tiling as k_parse_samples, but a lane owns CONSECUTIVE tokens of the tile
// so that its output is one contiguous span: pass 1 (EMIT = false) leaves the record's length and flag, the caller
// turns lengths into offsets, pass 2 writes.
// ---------------------------------------------------------------------------------------------------------------------
struct FormatArgs {
    trk_format_in in;
    trk_format_out out;
};

// "%g" of x into buf: the length, or -1 when the host has to print it (non-finite, a rounding tie, an exponent beyond
// the table).  Six significant digits by ONE rounded product / quotient with a power of ten; fixed notation for decimal
// exponents -4 .. 5, d[.ddddd]e+XX from 10^6 on (depths of seven digits), trailing zeros dropped -- printf's rules.
__device__ int g6(char* buf, double x) {
    int n = 0;
    if (!(x == x) || x - x != 0.0) return -1;
    if (x == 0.0) {
        if (__builtin_signbit(x)) return -1;
        buf[0] = '0';
        return 1;
    }
    if (x < 0) {
        buf[n++] = '-';
        x = -x;
    }
    // (below 1e-4 printf writes an exponent too, but no such value gets here: a token that spells one without an exponent is
    // not canonical -- the record is the host's -- and one with an exponent is outside the device parser's grammar)
    // (-- except the values a hair below 1e-4 that round up to it: float32 0.0001 is 9.99999975e-05)
    if (x < 9.99999e-5 || x >= 1e15) return -1;
    int e;                                               // decimal exponent of the first significant digit
    if (x >= 1.0) {
        e = 0;
        while (e < 14 && x >= c_p10[e + 1]) ++e;
    } else {
        e = x >= 1e-1 ? -1 : x >= 1e-2 ? -2 : x >= 1e-3 ? -3 : x >= 1e-4 ? -4 : -5;
    }
    const double y = e <= 5 ? x * c_p10[5 - e] : x / c_p10[e - 5];   // six significant digits before the point
    const double fl = floor(y), fr = y - fl;
    if (fabs(fr - 0.5) < 1e-6) return -1;               // too close to a tie for one rounded product to decide
    uint32_t d = (uint32_t)fl + (fr > 0.5 ? 1u : 0u);
    if (d >= 1000000u) {                                // 999999.6 -> 1000000: one more digit before the point
        d = 100000u;
        ++e;
    } else if (d < 100000u) {                           // the exponent guessed one too high (x a hair below a power of ten)
        return -1;
    }
    if (e < -4) return -1;                              // (stayed below 1e-4 after rounding: an exponent form, the host's)
    char dig[6];
    for (int i = 5; i >= 0; --i) {
        dig[i] = (char)('0' + d % 10u);
        d /= 10u;
    }
    int last = 5;
    while (last > 0 && dig[last] == '0') --last;        // significant digits dig[0 .. last]
    if (e < -4 || e >= 6) {                             // d[.ddddd]e[+-]XX
        buf[n++] = dig[0];
        if (last > 0) {
            buf[n++] = '.';
            for (int i = 1; i <= last; ++i) buf[n++] = dig[i];
        }
        buf[n++] = 'e';
        int ae = e;
        if (ae < 0) {
            buf[n++] = '-';
            ae = -ae;
        } else {
            buf[n++] = '+';
        }
        buf[n++] = (char)('0' + ae / 10);
        buf[n++] = (char)('0' + ae % 10);
        return n;
    }
    if (e >= 0) {
        for (int i = 0; i <= e; ++i) buf[n++] = dig[i];
        if (last > e) {
            buf[n++] = '.';
            for (int i = e + 1; i <= last; ++i) buf[n++] = dig[i];
        }
    } else {
        buf[n++] = '0';
        buf[n++] = '.';
        for (int i = 0; i < -e - 1; ++i) buf[n++] = '0';
        for (int i = 0; i <= last; ++i) buf[n++] = dig[i];
    }
    return n;
}

// one token: its length, the fields a kept call lacks, false when the host has to write the record
__device__ __forceinline__ bool tok_check(const unsigned char* tok, const unsigned char* tok_end, const uint8_t* kinds, int nf,
                                          int pl, bool flt, int& pad) {
    const unsigned char* c = tok;
    pad = 0;
    for (int f = 0; f < nf; ++f) {
        const int kind = kinds[f];
        if (flt) {
            while (*c != ':' && *c != '\t' && *c != ',' && *c != '\n' && *c != '\r' && *c != 0) ++c;
            if (*c == ',') return false;
            if (c == tok && f == 0) return false;
        } else if (kind == 1) {
            int na = 0;
            unsigned char sep = 0;
            for (;;) {
                if (*c == '.') {
                    ++c;
                } else if (*c == '0') {
                    ++c;
                    if (is_digit(*c)) return false;
                } else if (is_digit(*c)) {
                    const unsigned char* a = c;
                    do ++c; while (is_digit(*c));
                    if (c - a > 9) return false;
                } else {
                    return false;
                }
                ++na;
                if (*c != '/' && *c != '|') break;
                if (sep && *c != sep) return false;
                sep = *c++;
            }
            if (na > pl) return false;
        } else if (kind == 2) {
            if (*c == '.') {
                ++c;
            } else {
                if (*c == '-') {
                    ++c;
                    if (*c == '0') return false;
                }
                if (*c == '0') {
                    ++c;
                    if (is_digit(*c)) return false;
                } else if (is_digit(*c)) {
                    const unsigned char* a = c;
                    do ++c; while (is_digit(*c));
                    if (c - a > 9) return false;
                } else {
                    return false;
                }
            }
        } else if (kind == 4) {
            const unsigned char* a = c;
            while (*c != ':' && *c != '\t' && *c != '\n' && *c != '\r') {
                if (*c >= 0x80 || *c == 0) return false;
                ++c;
            }
            if (c == a) return false;
        } else {
            if (*c == '.' && !is_digit(c[1])) {
                ++c;
            } else {
                if (*c == '-') ++c;
                const unsigned char* ip = c;
                while (is_digit(*c)) ++c;
                const long ni = c - ip;
                if (ni < 1 || (ni > 1 && *ip == '0') || ni > 6) return false;
                if (*c == '.') {
                    const unsigned char* fp = ++c;
                    while (is_digit(*c)) ++c;
                    const long nfr = c - fp;
                    if (nfr < 1 || c[-1] == '0') return false;
                    if (*ip != '0') {
                        if (ni + nfr > 6) return false;
                    } else {
                        const unsigned char* z = fp;
                        while (*z == '0') ++z;
                        if (z - fp > 3 || c - z > 6) return false;
                    }
                }
            }
        }
        if (f + 1 < nf) {
            if (*c == ':') { ++c; continue; }
            if (c == tok_end) {
                pad = flt ? 0 : nf - 1 - f;
                break;
            }
            return false;
        }
    }
    return c == tok_end;
}

// the bytes of the lane's tokens k0 .. k1 - 1 at w (LDS stage or global memory: one inlined copy per address space)
struct EmitCtx {
    const unsigned char* tb;      // the tile's text in LDS
    const uint16_t* tokv;         // token starts (LDS), tokv[ntok] = the end of the tile's last token + 1
    const uint8_t* mrow;
    const char* nullv;
    int nl, nf;
    int64_t base;
    int64_t prow;                 // rec * plane_stride
};

template <typename W>
__device__ __forceinline__ void emit_tokens(W w, const EmitCtx& e, const trk_format_in& in, uint32_t k0, uint32_t k1) {
    for (uint32_t k = k0; k < k1; ++k) {
        const int64_t s = e.base + k;
        const unsigned char* tok = e.tb + e.tokv[k];
        const unsigned char* tok_end = e.tb + e.tokv[k + 1] - 1;
        const uint8_t mb = e.mrow[s];
        const bool flt = (mb & 0x7f) != 0 && !(mb & 0x80);
        *w++ = '\t';
        if (flt) {
            for (int i = 0; i < e.nl; ++i) *w++ = (unsigned char)e.nullv[i];
            *w++ = ':';
            int cnt = 0;
            for (int b = 0; b < in.n_filters; ++b) {
                if (!((mb >> b) & 1)) continue;
                if (cnt++) *w++ = ',';
                for (int i = 0; in.filter_name[b][i]; ++i) *w++ = (unsigned char)in.filter_name[b][i];
                *w++ = '_';
                char tmp[24];
                const double x = in.filter_dtype[b] ? (double)static_cast<const float*>(in.filter_plane[b])[e.prow + s]
                                                    : (double)static_cast<const int32_t*>(in.filter_plane[b])[e.prow + s];
                const int gl = g6(tmp, x);
                for (int i = 0; i < gl; ++i) *w++ = (unsigned char)tmp[i];
            }
            if (!cnt) *w++ = '.';
        } else {
            int colons = 0;
            for (const unsigned char* c = tok; c < tok_end; ++c) {
                const unsigned char ch = *c;
                colons += ch == ':';
                *w++ = ch;
            }
            int pad = e.nf - 1 - colons;    // fields the token lacks: its colons against the record's keys
            for (int i = 0; i < pad; ++i) {
                *w++ = ':';
                *w++ = '.';
            }
            const char* tag = (mb & 0x80) ? ":NOCALL" : ":PASS";
            for (int i = 0; tag[i]; ++i) *w++ = (unsigned char)tag[i];
        }
    }
}

template <bool EMIT>
__global__ __launch_bounds__(PS_THREADS) void k_format_samples(const FormatArgs a) {
    __shared__ uint16_t s_cnt[FS_CHUNKS];
    __shared__ uint16_t s_tok[FS_MAXTOK + 1];      // token starts of the tile, relative to the tile's first byte (t0 - lead)
    // the tile's TEXT (+ FS_OVER bytes beyond it): the lanes walk their tokens byte by byte, from LDS
    __shared__ uint4 s_text[FS_CHUNKS + FS_OVER / 16];
    // EMIT: the tile's output is assembled in LDS (every lane writes its tokens' bytes there) and leaves for global
    // memory four bytes per lane, coalesced; a tile whose output does not fit is written byte by byte
    __shared__ uint32_t s_stage[EMIT ? FS_STAGE / 4 + 2 : 1];
    __shared__ uint32_t s_wsum[PS_THREADS / 64];
    __shared__ uint32_t s_flags, s_base, s_obase;
    __shared__ uint8_t s_kinds[TRK_FORMAT_MAX_FIELDS];
    __shared__ char s_null[64];
    __shared__ int s_nl;
    const int rec = blockIdx.x;
    const int tid = threadIdx.x;
    const int S = a.in.n_samples;
    const int nf = a.in.n_fields[rec], pl = a.in.ploidy[rec];
    if (EMIT && a.out.flags[rec]) return;                  // the host writes this record
    const int64_t r0 = a.in.smp_off[rec], r1 = a.in.line_end[rec];
    if (tid == 0) {
        s_flags = (nf < 1 || nf > TRK_FORMAT_MAX_FIELDS || pl < 1 || pl > 8 || r1 <= r0) ? TRK_PARSE_HOST : 0u;
        s_base = 0;
        s_obase = 0;
        int nl = 0;
        for (int f = 0; f < nf && f < TRK_FORMAT_MAX_FIELDS; ++f) {
            const uint8_t k = a.in.field_kind[(int64_t)rec * TRK_FORMAT_MAX_FIELDS + f];
            s_kinds[f] = k;
            if (k < 1 || k > 4) s_flags = TRK_PARSE_HOST;
            if (f) s_null[nl++] = ':';
            if (k == 1) {
                for (int j = 0; j < pl && j < 8; ++j) {
                    if (j) s_null[nl++] = '/';
                    s_null[nl++] = '.';
                }
            } else {
                s_null[nl++] = '.';
            }
        }
        s_nl = nl;
    }
    __syncthreads();
    if (s_flags) {
        if (!EMIT && tid == 0) {
            a.out.flags[rec] = (uint8_t)s_flags;
            a.out.rec_len[rec] = 0;
        }
        return;
    }
    const unsigned char* reg = a.in.text + r0;
    const int64_t n = r1 - r0;                              // reg[n] is the newline
    const int64_t lead = (int64_t)((uintptr_t)reg & 15u);
    const int64_t span = n + lead;
    const uint8_t* mrow = a.in.mask8 + (int64_t)rec * a.in.mask_stride;
    unsigned char* obuf = EMIT ? a.out.out + a.out.out_off[rec] : nullptr;
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(s_text);
    for (int64_t t0 = 0; t0 < span; t0 += FS_TILE) {
        uint32_t mk[FS_Q];
#pragma unroll
        for (int q = 0; q < FS_Q; ++q) {
            const int c = tid + q * PS_THREADS;
            const int64_t o = t0 + (int64_t)c * 16 - lead;
            uint32_t m = 0;
            if (o <= n && o + 16 > 0) {                     // (the chunk that holds the newline too: the last token ends on it)
                const uint4 v = *reinterpret_cast<const uint4*>(reg + o);
                s_text[c] = v;
                m = tab_mask(v);
                if (o < 0) m &= ~((1u << (int)(-o)) - 1u);
                if (o + 16 > n) m &= (1u << (int)(n - o)) - 1u;
            }
            mk[q] = m;
            s_cnt[c] = (uint16_t)__popc(m);
        }
        if (tid < FS_OVER / 16) {
            const int c = FS_CHUNKS + tid;
            const int64_t o = t0 + (int64_t)c * 16 - lead;
            if (o <= n) s_text[c] = *reinterpret_cast<const uint4*>(reg + o);
        }
        __syncthreads();
        uint32_t own[FS_Q], sum = 0;
#pragma unroll
        for (int q = 0; q < FS_Q; ++q) {
            own[q] = s_cnt[FS_Q * tid + q];
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
        for (int q = 0; q < FS_Q; ++q) {
            s_cnt[FS_Q * tid + q] = (uint16_t)run;
            run += own[q];
        }
        __syncthreads();
        const uint32_t first = t0 == 0 ? 1u : 0u;
        if (t0 == 0 && tid == 0) s_tok[0] = (uint16_t)lead;      // (the region's first byte, relative to t0 - lead = -lead)
#pragma unroll
        for (int q = 0; q < FS_Q; ++q) {
            const int c = tid + q * PS_THREADS;
            uint32_t m = mk[q];
            uint32_t k = first + s_cnt[c];
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                if (k < (uint32_t)FS_MAXTOK) s_tok[k] = (uint16_t)(c * 16 + b + 1);
                ++k;
            }
        }
        __syncthreads();
        const uint32_t base = s_base, obase = s_obase;
        const uint32_t ntok = first + tile_tabs;
        if (ntok > (uint32_t)FS_MAXTOK || (int64_t)base + ntok > S) {     // (uniform) too many columns, or degenerate text
            if (tid == 0) s_flags |= TRK_PARSE_HOST;
            __syncthreads();
            break;
        }
        // the tile's last token ends at the next tab or at the newline, within the staged text -- or the host writes the record
        if (tid == 0 && ntok) {
            const int64_t lim = min((int64_t)(FS_TILE + FS_OVER), span - t0);   // staged bytes; span - t0: the newline's offset
            int64_t e = s_tok[ntok - 1];
            while (e < lim && tb[e] != '\t') ++e;
            if (e >= lim && lim != span - t0) s_flags |= TRK_PARSE_HOST;
            s_tok[ntok] = (uint16_t)(e + 1);
        }
        __syncthreads();
        if (s_flags) break;                 // (uniform)
        // ---- lane t owns tokens [t per, (t + 1) per) of the tile: lengths, then (EMIT) the bytes at its running offset ----
        const uint32_t per = (ntok + PS_THREADS - 1) / PS_THREADS;
        const uint32_t k0 = min(ntok, (uint32_t)tid * per), k1 = min(ntok, k0 + per);
        uint32_t lflags = 0;
        uint32_t mylen = 0;
        for (uint32_t k = k0; k < k1; ++k) {
            const int64_t s = (int64_t)base + k;
            const unsigned char* tok = tb + s_tok[k];
            const unsigned char* tok_end = tb + s_tok[k + 1] - 1;     // the next token's tab, or the newline
            const uint8_t mb = mrow[s];
            const bool flt = (mb & 0x7f) != 0 && !(mb & 0x80);
            int pad;
            if (!tok_check(tok, tok_end, s_kinds, nf, pl, flt, pad)) {
                lflags |= TRK_PARSE_HOST;
                continue;
            }
            uint32_t len = 1;
            if (flt) {
                len += (uint32_t)s_nl + 1;
                int cnt = 0;
                for (int b = 0; b < a.in.n_filters; ++b) {
                    if (!((mb >> b) & 1)) continue;
                    if (cnt++) ++len;
                    int ln = 0;
                    while (a.in.filter_name[b][ln]) ++ln;
                    char tmp[24];
                    const double x = a.in.filter_dtype[b] ? (double)static_cast<const float*>(a.in.filter_plane[b])[(int64_t)rec * a.in.plane_stride + s]
                                                          : (double)static_cast<const int32_t*>(a.in.filter_plane[b])[(int64_t)rec * a.in.plane_stride + s];
                    const int gl = g6(tmp, x);
                    if (gl < 0) { lflags |= TRK_PARSE_HOST; break; }
                    len += (uint32_t)ln + 1 + (uint32_t)gl;
                }
                if (!cnt) ++len;
            } else {
                len += (uint32_t)(tok_end - tok) + 2u * (uint32_t)pad + ((mb & 0x80) ? 7u : 5u);
            }
            mylen += len;
        }
        // exclusive scan of the lanes' lengths over the workgroup
        uint32_t li = mylen;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)li, o, 64);
            if ((tid & 63) >= o) li += y;
        }
        __syncthreads();
        if ((tid & 63) == 63) s_wsum[tid >> 6] = li;
        __syncthreads();
        uint32_t lbase = 0;
        for (int w = 0; w < (tid >> 6); ++w) lbase += s_wsum[w];
        const uint32_t tile_len = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        if (EMIT) {
            const EmitCtx ec{tb, s_tok, mrow, s_null, s_nl, nf, (int64_t)base, (int64_t)rec * a.in.plane_stride};
            const uint32_t woff = lbase + li - mylen;
            if (tile_len <= (uint32_t)FS_STAGE) {       // (uniform)
                emit_tokens(reinterpret_cast<unsigned char*>(s_stage) + woff, ec, a.in, k0, k1);
                __syncthreads();
                unsigned char* dst = obuf + obase;
                const unsigned char* st8 = reinterpret_cast<const unsigned char*>(s_stage);
                const uint32_t head = min(tile_len, (uint32_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u));
                if ((uint32_t)tid < head) dst[tid] = st8[tid];
                const uint32_t nd = (tile_len - head) >> 2;
                uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
                const uint32_t sh = 8u * (head & 3u);
                for (uint32_t i = tid; i < nd; i += PS_THREADS) {
                    const uint32_t sb = head + 4u * i;
                    const uint32_t lo = s_stage[sb >> 2], hi = s_stage[(sb >> 2) + 1];
                    d32[i] = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
                }
                const uint32_t done = head + 4u * nd;
                if ((uint32_t)tid < tile_len - done) dst[done + tid] = st8[done + tid];
            } else {
                emit_tokens(obuf + obase + woff, ec, a.in, k0, k1);
            }
        }
        if (lflags) atomicOr(&s_flags, lflags);
        __syncthreads();
        if (tid == 0) {
            s_base = base + ntok;
            s_obase = obase + tile_len;
        }
        __syncthreads();
        if (s_flags) break;                 // (uniform) the host writes this record: no need to go on
    }
    if (!EMIT && tid == 0) {
        uint32_t f = s_flags;
        if (!f && (int64_t)s_base != S) f |= TRK_PARSE_HOST;
        a.out.flags[rec] = (uint8_t)f;
        a.out.rec_len[rec] = f ? 0u : s_obase;
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

hipError_t launch_format_samples(const trk_format_in& in, const trk_format_out& out, int pass, hipStream_t stream) {
    if (in.n_records <= 0) return hipSuccess;
    FormatArgs a{in, out};
    if (pass == 1) hipLaunchKernelGGL(k_format_samples<false>, dim3((unsigned)in.n_records), dim3(PS_THREADS), 0, stream, a);
    else hipLaunchKernelGGL(k_format_samples<true>, dim3((unsigned)in.n_records), dim3(PS_THREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace trk
