// trk_deflate.hip -- BGZF members DEFLATED on the device (round 6; the mirror of trk_inflate.hip).
//
// What it replaces: `bgzip -f` over dumpSTR's output VCF (the reference shells out: dumpSTR.py:1241-1245, 1347-1352), i.e.
// zlib's deflate over 0xff00-byte members.  Any DEFLATE stream that inflates to the member's text is a right answer; the one
// made here is defined by tests/deflate_model.py (a line-by-line model, test infrastructure), and the tests check the bytes
// against that model and -- the actual requirement -- through zlib's inflate.
//
// One WAVE per member of 16 KB of text, twenty members per CU, thousands in flight: as in k_inflate_bgzf the work of a member is a serial
// chain and the parallelism is ACROSS members (a 1.5 GB output is 92 000 of them).
//   stage A  greedy LZ77.  The candidate table (LDS, 4 KB): 256 buckets by the hash of four bytes, eight places each, position q
//            in place q % 8 -- the last eight residues seen.  At a position the bucket's eight candidates are compared SIDE BY
//            SIDE, eight lanes and 32 bytes each, in one round trip to the text (the longest wins, the nearest among equals;
//            only the winner is followed further); every position of a token enters the table, the lanes in parallel.
//            Tokens go to a per-wave stream in global memory (32 bits each), symbol frequencies to LDS.  (Until the end of
//            round 6: one candidate per position, token starts only -- files 12-15 % larger.)
//   stage B  code lengths of the three alphabets by the two-queue Huffman construction (leaves ranked by all lanes, the
//            merge by one), frequencies halved while a code is longer than its limit; canonical codes;
//   stage C  the block header (code-length runs) by one lane, then the tokens sixty-four at a time: every lane makes one
//            token's bits, a scan places them, the lanes OR them into a window of the stream in LDS; a member that would
//            not get smaller is STORED.
// The payload of member m goes to slots + m * SLOT, its length to sizes[m]; k_deflate_pack lays the members out back to
// back with their BGZF headers and trailers (the CRC-32 is the host's: the text came from there).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trk.h"
#include "trk_internal.h"

namespace {

constexpr int WAVE = 64;
constexpr int DF_WAVES = 4;                   // members per workgroup
constexpr int DF_MEMBER = TRK_DEFLATE_MEMBER; // bytes of text per member (include/trk.h): 16 KB, a quarter of bgzip's -- a
                                              // member is ONE wave's serial work, and a 150 MB block of dumpSTR's output is
                                              // 9 000 of these against 2 300 of bgzip's size: the chip has 6 000 wave slots
constexpr int DF_HB = 8, DF_WAYS = 8;          // the candidate table: 256 buckets (hash of four bytes) of eight places
constexpr int DF_FIRST = 16, DF_INSERT = 64;  // bytes of every candidate compared at once; positions of a match that enter the table
constexpr int DF_MIN = 4, DF_MAX = 258, DF_DIST = 32768;
static_assert(DF_MEMBER <= DF_DIST, "every earlier position of a member is within reach of a distance code");
constexpr int DF_NLL = 288, DF_NDL = 32, DF_NCL = 32;     // alphabet array sizes (286 / 30 / 19 used)
constexpr uint32_t DF_SLOT = DF_MEMBER + 64;  // bytes of a member's payload slot (a stored member: text + 5)
constexpr uint32_t DF_TOKCAP = DF_MEMBER + 64;   // tokens (32 bits each) of a wave's stream
#ifndef DF_OCC
#define DF_OCC 5
#endif
constexpr int DF_WGS_PER_CU = DF_OCC;         // workgroups (of one wave per SIMD) resident per CU: the kernel is held to 96
                                              // registers for it (it took 129, three waves per SIMD, while the LDS had room for six)

__constant__ uint16_t c_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint16_t c_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                     2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_clord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ int len_sym(int l) {            // 3 ... 258 -> 0 ... 28
    if (l == 258) return 28;
    const int x = l - 3;
    if (x < 8) return x;
    const int b = 31 - __builtin_clz((unsigned)x);
    return 4 * (b - 1) + ((x >> (b - 2)) & 3);
}
__device__ __forceinline__ int len_extra(int sym) { return sym < 8 || sym == 28 ? 0 : (sym >> 2) - 1; }
__device__ __forceinline__ int dist_sym(int d) {           // 1 ... 32768 -> 0 ... 29
    if (d <= 4) return d - 1;
    const int x = d - 1;
    const int b = 31 - __builtin_clz((unsigned)x);
    return 2 * b + ((x >> (b - 1)) & 1);
}
__device__ __forceinline__ int dist_extra(int sym) { return sym < 4 ? 0 : (sym >> 1) - 1; }

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p) {      // any alignment (one global_load_dword)
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ int grp4_min(int x) {       // the minimum over the four lanes of a group, in every one of them
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ uint32_t row_max(uint32_t x) {      // the maximum over a row of sixteen lanes whose groups of four agree
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x124, 0xf, 0xf, false));     // row_ror:4
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xf, 0xf, false));     // row_ror:8
    return x;
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct DeflArgs {
    const uint8_t* text;
    int64_t n_total;
    int32_t n_members;
    uint8_t* slots;        // [n_members][DF_SLOT]
    uint32_t* sizes;       // [n_members] payload bytes
    uint32_t* tok;         // [gridDim.x * DF_WAVES][DF_TOKCAP]: a literal's byte, or 0x80000000 | (length - 3) << 16 | distance - 1
};

// per-wave LDS (6.4 KB): the hash table of stage A is the scratch of stage B.  Sixteen-bit counters throughout: a member
// holds at most 16 385 symbols.
struct __attribute__((aligned(16))) WaveLds {
    union {
        uint16_t htab[DF_WAYS << DF_HB];                             // 4096 B
        struct {
            uint16_t order[DF_NLL];     // leaves in (frequency, symbol) order
            uint16_t weight[2 * DF_NLL];
            uint16_t parent[2 * DF_NLL];
            uint8_t depth[2 * DF_NLL];
            uint16_t fcopy[DF_NLL];     // the frequencies the tree is built from (halved while a code is too long)
        } hb;
        struct {
            uint16_t count[16], next[16];      // canonical_codes: codes per length, the next code of a length
        } cn;
    } u;
    uint16_t lf[DF_NLL], df[DF_NDL], cf[DF_NCL];                     // frequencies as counted
    uint8_t ll[DF_NLL], dl[DF_NDL], cl[DF_NCL];                      // code lengths
    uint16_t lc[DF_NLL], dc[DF_NDL], cc[DF_NCL];                     // codes, bit-reversed
    uint8_t run_sym[DF_NLL + DF_NDL], run_ext[DF_NLL + DF_NDL];      // the code-length sequence, run-length coded
};

// Code lengths of one alphabet (tests/deflate_model.py: huffman_lengths).  f[0 .. n) is left as it is.
__device__ void huffman_lengths(WaveLds& w, const uint16_t* f0, int n, int limit, uint8_t* len, int lane) {
    uint16_t* f = w.u.hb.fcopy;
    for (int s = lane; s < n; s += WAVE) f[s] = f0[s];
    wave_sync();
    for (;;) {
        // rank of every used symbol among the used ones by (frequency, symbol)
        int used_here = 0;
        for (int s = lane; s < n; s += WAVE) {
            const uint32_t fs = f[s];
            if (!fs) continue;
            ++used_here;
            int rank = 0;
#pragma unroll 4
            for (int t = 0; t < n; ++t) {
                const uint32_t ft = f[t];
                rank += (ft != 0u) & ((ft < fs) | ((ft == fs) & (t < s)));
            }
            w.u.hb.order[rank] = (uint16_t)s;
        }
        int m = used_here;
        for (int o = 32; o; o >>= 1) m += __shfl_xor(m, o, WAVE);
        for (int s = lane; s < n; s += WAVE) len[s] = 0;
        wave_sync();
        if (m == 0) return;
        if (m == 1) {
            if (lane == 0) len[w.u.hb.order[0]] = 1;
            wave_sync();
            return;
        }
        for (int k = lane; k < m; k += WAVE) w.u.hb.weight[k] = f[w.u.hb.order[k]];
        wave_sync();
        if (lane == 0) {
            // two queues: leaves 0 .. m-1 (sorted), internal nodes m .. 2m-2 in the order they are made
            int li = 0, ii = m, made = m;
            for (int it = 0; it < m - 1; ++it) {
                int pick[2];
                for (int k = 0; k < 2; ++k) {
                    if (li < m && (ii >= made || w.u.hb.weight[li] <= w.u.hb.weight[ii])) pick[k] = li++;
                    else pick[k] = ii++;
                }
                w.u.hb.weight[made] = (uint16_t)(w.u.hb.weight[pick[0]] + w.u.hb.weight[pick[1]]);
                w.u.hb.parent[pick[0]] = w.u.hb.parent[pick[1]] = (uint16_t)made;
                ++made;
            }
            w.u.hb.depth[2 * m - 2] = 0;
            for (int node = 2 * m - 3; node >= 0; --node) w.u.hb.depth[node] = (uint8_t)(w.u.hb.depth[w.u.hb.parent[node]] + 1);
        }
        wave_sync();
        int maxd = 0;
        for (int k = lane; k < m; k += WAVE) maxd = max(maxd, (int)w.u.hb.depth[k]);
        for (int o = 32; o; o >>= 1) maxd = max(maxd, __shfl_xor(maxd, o, WAVE));
        if (maxd <= limit) {
            for (int k = lane; k < m; k += WAVE) len[w.u.hb.order[k]] = w.u.hb.depth[k];
            wave_sync();
            return;
        }
        for (int s = lane; s < n; s += WAVE) {
            const uint32_t x = f[s];
            f[s] = (uint16_t)(x ? (x + 1) >> 1 : 0);
        }
        wave_sync();
    }
}

// canonical codes, bit-reversed (one lane; n <= 288).  The two small tables are indexed by a code LENGTH: in the LDS (the
// tree's scratch is free by now), not in thirty-two registers behind select chains.
__device__ void canonical_codes(WaveLds& w, const uint8_t* len, int n, uint16_t* code) {
    uint16_t* count = w.u.cn.count;
    uint16_t* nxt = w.u.cn.next;
    for (int b = 0; b < 16; ++b) count[b] = 0;
    for (int s = 0; s < n; ++s) ++count[len[s] & 15];
    count[0] = 0;
    uint32_t c = 0;
    nxt[0] = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + count[b - 1]) << 1;
        nxt[b] = (uint16_t)c;
    }
    for (int s = 0; s < n; ++s) {
        const int l = len[s];
        uint32_t r = 0;
        if (l) {
            const uint32_t v = nxt[l]++;
            r = __builtin_bitreverse32(v) >> (32 - l);
        }
        code[s] = (uint16_t)r;
    }
}

struct BitOut {
    uint32_t* out;
    uint64_t acc;
    int n;
    uint32_t words;
    __device__ __forceinline__ void put(uint32_t v, int b) {
        acc |= (uint64_t)v << n;
        n += b;
        if (n >= 32) {
            out[words++] = (uint32_t)acc;
            acc >>= 32;
            n -= 32;
        }
    }
    __device__ __forceinline__ uint32_t finish() {     // bytes written
        const uint32_t bytes = words * 4u + (uint32_t)((n + 7) >> 3);
        if (n > 0) out[words] = (uint32_t)acc;
        return bytes;
    }
};

__global__ __launch_bounds__(WAVE* DF_WAVES) __attribute__((amdgpu_waves_per_eu(DF_OCC, DF_OCC))) void k_deflate_bgzf(const DeflArgs a) {
    __shared__ WaveLds lds[DF_WAVES];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    WaveLds& w = lds[wid];
    const int wave_slot = blockIdx.x * DF_WAVES + wid, n_slots = gridDim.x * DF_WAVES;
    uint32_t* tok = a.tok + (size_t)wave_slot * DF_TOKCAP;
    for (int m = wave_slot; m < a.n_members; m += n_slots) {
        const uint8_t* text = a.text + (int64_t)m * DF_MEMBER;
        const int n = (int)min<int64_t>(DF_MEMBER, a.n_total - (int64_t)m * DF_MEMBER);
        uint32_t* payload = reinterpret_cast<uint32_t*>(a.slots + (size_t)m * DF_SLOT);
        // ---- stage A: tokens and frequencies ------------------------------------------------------------------------
        {
            uint4* h4 = reinterpret_cast<uint4*>(w.u.htab);
            const uint4 z = {0u, 0u, 0u, 0u};
            for (int i = lane; i < (int)(sizeof w.u.htab / 16); i += WAVE) h4[i] = z;
            for (int s = lane; s < DF_NLL; s += WAVE) w.lf[s] = 0;
            if (lane < DF_NDL) w.df[lane] = 0;
        }
        wave_sync();
        // The text is read through a WINDOW of 256 bytes in registers (a dword per lane, refilled with one coalesced load
        // when the position leaves it): the four bytes at p are two v_readlane and a shift -- a global load per position
        // was a round trip to the L2 per TOKEN (19 ms per member of 64 KB).  Only a candidate's bytes come from memory.
        const uint32_t* text32 = reinterpret_cast<const uint32_t*>(text);      // (members start on 16 KB boundaries of the buffer)
        int wb = 0;                                        // dword index of the window's first dword
        uint32_t win = text32[lane];                       // (the buffer is padded: reads beyond the text's end stay inside it)
        int p = 0;
        uint32_t tc = 0;
        const int grp = lane >> 2, j4 = (lane & 3) * 4;    // sixteen groups of four lanes: a candidate each, 16 bytes of it
        const int second = grp >> 3;                       // groups 8 ... 15 look at p + 1
        while (p < n) {
            int wi = (p >> 2) - wb;
            if (wi >= WAVE - 1) {
                wb = p >> 2;
                win = text32[wb + lane];
                wi = 0;
            }
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)win, wi), hi = (uint32_t)__builtin_amdgcn_readlane((int)win, wi + 1);
            const uint64_t v8 = (((uint64_t)hi << 32) | lo) >> (8 * (p & 3));
            const uint32_t v0 = (uint32_t)v8, v1 = (uint32_t)(v8 >> 8);      // the four bytes at p, at p + 1
            // ---- the match: TWO positions against the table as it is, their sixteen candidates side by side, sixteen bytes
            // ---- each, in ONE round trip to the text; the better one wins (p on a tie), only the winner is followed further
            int b0 = 0, d0 = 0, b1 = 0, d1 = 0;
            uint32_t tq0 = v0, tq1 = v1;                   // the four bytes at p + lane, p + 1 + lane (loaded when a match may be found)
            {
                const int pos = p + second;
                const uint32_t h = ((second ? v1 : v0) * 2654435761u) >> (32 - DF_HB);
                const int c = pos + 4 <= n ? (int)w.u.htab[h * DF_WAYS + (grp & 7)] : 0;      // position + 1, 0: none
                if (__ballot(c != 0)) {
                    tq0 = load_u32(text + p + lane);
                    tq1 = load_u32(text + p + 1 + lane);
                    int lj = 999;
                    if (c) {
                        const uint32_t x = load_u32(text + pos + j4) ^ load_u32(text + (c - 1) + j4);
                        if (x) lj = j4 + (__builtin_ctz(x) >> 3);
                    }
                    lj = grp4_min(lj);
                    const int l = min(min(lj, DF_FIRST), n - pos);
                    uint32_t key = c ? ((uint32_t)l << 16) | (uint32_t)(0xffff - (pos - (c - 1))) : 0u;      // longest, then nearest
                    key = row_max(key);                    // (a row of sixteen lanes = four candidates)
                    const uint32_t k0 = max((uint32_t)__builtin_amdgcn_readlane((int)key, 0), (uint32_t)__builtin_amdgcn_readlane((int)key, 16));
                    const uint32_t k1 = max((uint32_t)__builtin_amdgcn_readlane((int)key, 32), (uint32_t)__builtin_amdgcn_readlane((int)key, 48));
                    if ((int)(k0 >> 16) >= DF_MIN) {
                        b0 = (int)(k0 >> 16);
                        d0 = 0xffff - (int)(k0 & 0xffffu);
                    }
                    if ((int)(k1 >> 16) >= DF_MIN) {
                        b1 = (int)(k1 >> 16);
                        d1 = 0xffff - (int)(k1 & 0xffffu);
                    }
                }
            }
            if (b1 > b0) {                                 // p + 1 has the better match: the byte at p is a literal
                if (lane == 0) {
                    if (p + 4 <= n) w.u.htab[((v0 * 2654435761u) >> (32 - DF_HB)) * DF_WAYS + (p & (DF_WAYS - 1))] = (uint16_t)(p + 1);
                    tok[tc] = v0 & 0xffu;
                    ++w.lf[v0 & 0xffu];
                }
                wave_sync();
                tc += 1;
                p += 1;
                b0 = b1;
                d0 = d1;
                tq0 = tq1;
            }
            if (b0) {
                int best = b0;
                const int limit = min(DF_MAX, n - p);
                if (best == DF_FIRST && limit > DF_FIRST) {
                    const int o = DF_FIRST + 4 * lane;     // (one step reaches 272 bytes: the limit is 258)
                    const uint32_t x = load_u32(text + p - d0 + o) ^ load_u32(text + p + o);
                    const int ml = x ? (__builtin_ctz(x) >> 3) : 4;
                    const uint64_t mm = __ballot(ml < 4);
                    int l2 = DF_FIRST + 4 * WAVE;
                    if (mm) {
                        const int t = __ffsll((unsigned long long)mm) - 1;
                        l2 = DF_FIRST + 4 * t + __builtin_amdgcn_readlane(ml, t);
                    }
                    best = min(l2, limit);
                }
                // the token's positions enter the table (a match's first 64): place q % 8 of the bucket of the four bytes at q
                const int q = p + lane;
                const bool act = lane < min(best, DF_INSERT) && q + 4 <= n;
                uint16_t* slot = &w.u.htab[((tq0 * 2654435761u) >> (32 - DF_HB)) * DF_WAYS + (q & (DF_WAYS - 1))];
                const uint16_t val = (uint16_t)(q + 1);
                // two positions of one token may share a place (a period of 8 in the text): the LATER one stays.  Which lane's
                // store the LDS keeps is not promised, so the lanes look and the ones that lost to an earlier position store again.
                bool pend = act;
                do {
                    if (pend) *slot = val;
                    wave_sync();
                    pend = act && *slot < val;
                } while (__ballot(pend));
                if (lane == 0) {
                    tok[tc] = 0x80000000u | ((uint32_t)(best - 3) << 16) | (uint32_t)(d0 - 1);
                    ++w.lf[257 + len_sym(best)];
                    ++w.df[dist_sym(d0)];
                }
                tc += 1;
                p += best;
            } else {
                // no match at p and none at p + 1: two literals
                const int two = p + 1 < n;
                if (lane < 1 + two) {
                    const uint32_t vq = lane ? v1 : v0;
                    const int q = p + lane;
                    if (q + 4 <= n) w.u.htab[((vq * 2654435761u) >> (32 - DF_HB)) * DF_WAYS + (q & (DF_WAYS - 1))] = (uint16_t)(q + 1);
                    tok[tc + lane] = vq & 0xffu;
                }
                if (lane == 0) {
                    ++w.lf[v0 & 0xffu];
                    if (two) ++w.lf[v1 & 0xffu];
                }
                wave_sync();
                tc += 1 + two;
                p += 1 + two;
            }
        }
        wave_sync();
        if (lane == 0) w.lf[256] = 1;
        wave_sync();
        // ---- stage B: code lengths and codes ------------------------------------------------------------------------
        huffman_lengths(w, w.lf, 286, 15, w.ll, lane);
        huffman_lengths(w, w.df, 30, 15, w.dl, lane);
        uint32_t payload_bytes = 0;
        if (lane == 0) {
            bool any = false;
            for (int s = 0; s < 30; ++s) any |= w.dl[s] != 0;
            if (!any) w.dl[0] = 1;           // (a block without matches still declares one distance code)
        }
        wave_sync();
        int hlit = 286, hdist = 30, n_runs = 0;
        if (lane == 0) {
#pragma nounroll
            while (hlit > 257 && w.ll[hlit - 1] == 0) --hlit;
#pragma nounroll
            while (hdist > 1 && w.dl[hdist - 1] == 0) --hdist;
            // the code-length sequence, run-length coded (tests/deflate_model.py: code_length_runs)
            for (int s = 0; s < DF_NCL; ++s) w.cf[s] = 0;
            const int total = hlit + hdist;
            int i = 0;
            auto at = [&](int k) -> int { return k < hlit ? w.ll[k] : w.dl[k - hlit]; };
            auto emit = [&](int sym, int ext) {
                w.run_sym[n_runs] = (uint8_t)sym;
                w.run_ext[n_runs] = (uint8_t)ext;
                ++n_runs;
                ++w.cf[sym];
            };
            while (i < total) {
                const int v = at(i);
                int j = i;
                while (j < total && at(j) == v) ++j;
                int run = j - i;
                if (v == 0) {
                    while (run >= 11) {
                        const int r = min(run, 138);
                        emit(18, r - 11);
                        run -= r;
                    }
                    if (run >= 3) {
                        emit(17, run - 3);
                        run = 0;
                    }
                    while (run > 0) {
                        emit(0, 0);
                        --run;
                    }
                } else {
                    emit(v, 0);
                    --run;
                    while (run >= 3) {
                        const int r = min(run, 6);
                        emit(16, r - 3);
                        run -= r;
                    }
                    while (run > 0) {
                        emit(v, 0);
                        --run;
                    }
                }
                i = j;
            }
        }
        hlit = __builtin_amdgcn_readfirstlane(hlit);
        hdist = __builtin_amdgcn_readfirstlane(hdist);
        n_runs = __builtin_amdgcn_readfirstlane(n_runs);
        wave_sync();
        huffman_lengths(w, w.cf, 19, 7, w.cl, lane);
        // ---- stage C: the bits ------------------------------------------------------------------------------------------
        uint32_t hdr_words = 0, hdr_acc = 0;
        int hdr_n = 0, dynamic = 0;
        if (lane == 0) {
            canonical_codes(w, w.ll, 286, w.lc);
            canonical_codes(w, w.dl, 30, w.dc);
            canonical_codes(w, w.cl, 19, w.cc);
            int hclen = 19;
            while (hclen > 4 && w.cl[c_clord[hclen - 1]] == 0) --hclen;
            // size of the dynamic block, in bits, from the frequencies as counted
            uint64_t bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
            for (int k = 0; k < n_runs; ++k) {
                const int s = w.run_sym[k];
                bits += w.cl[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0);
            }
#pragma unroll 2
            for (int s = 0; s < 286; ++s) bits += (uint64_t)w.lf[s] * (w.ll[s] + (s >= 257 ? len_extra(s - 257) : 0));
#pragma unroll 2
            for (int s = 0; s < 30; ++s) bits += (uint64_t)w.df[s] * (w.dl[s] + dist_extra(s));
            const uint32_t dyn_bytes = (uint32_t)((bits + 7) >> 3);
            if (dyn_bytes <= (uint32_t)n + 5u) {
                BitOut bo{payload, 0, 0, 0};
                bo.put(1, 1);              // BFINAL
                bo.put(2, 2);              // dynamic Huffman
                bo.put((uint32_t)(hlit - 257), 5);
                bo.put((uint32_t)(hdist - 1), 5);
                bo.put((uint32_t)(hclen - 4), 4);
                for (int k = 0; k < hclen; ++k) bo.put(w.cl[c_clord[k]], 3);
                for (int k = 0; k < n_runs; ++k) {
                    const int s = w.run_sym[k];
                    bo.put(w.cc[s], w.cl[s]);
                    if (s == 16) bo.put(w.run_ext[k], 2);
                    else if (s == 17) bo.put(w.run_ext[k], 3);
                    else if (s == 18) bo.put(w.run_ext[k], 7);
                }
                // the header is written; the tokens follow from bit `hdr_bits` on (all lanes, below)
                hdr_words = bo.words;
                hdr_acc = (uint32_t)bo.acc;
                hdr_n = bo.n;
                dynamic = 1;
            }
        }
        dynamic = __builtin_amdgcn_readfirstlane(dynamic);
        if (dynamic) {
            // ---- the tokens, SIXTY-FOUR AT A TIME: every lane makes the bits of one token (at most 48), a scan of the lengths
            // says where they go, the lanes OR them into a window of the stream in LDS and the window's full dwords leave for
            // the payload.  (One lane walking the token stream took a dependent global load per token: half of the member's
            // time.)  The end-of-block symbol is token number tc.
            uint32_t* stage = reinterpret_cast<uint32_t*>(w.u.htab);         // (stage B's scratch is done with: 100 dwords)
            uint32_t out_word = __builtin_amdgcn_readfirstlane(hdr_words);   // dwords of the payload already complete
            uint32_t carry_bits = (uint32_t)__builtin_amdgcn_readfirstlane(hdr_n);      // bits of the open dword
            uint32_t carry = __builtin_amdgcn_readfirstlane(hdr_acc);
            // (the token stream was written by lane 0: its stores have completed before the other lanes read it -- one L1 per
            // CU, so workgroup scope is enough)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (uint32_t t0 = 0; t0 <= tc; t0 += WAVE) {
                const uint32_t t = t0 + (uint32_t)lane;
                uint64_t v = 0;
                int nb = 0;
                if (t < tc) {
                    const uint32_t u = tok[t];
                    if (u & 0x80000000u) {
                        const int l = (int)((u >> 16) & 0x7fffu) + 3, d = (int)(u & 0xffffu) + 1;
                        const int ls = len_sym(l);
                        v = w.lc[257 + ls];
                        nb = w.ll[257 + ls];
                        const int le = len_extra(ls);
                        v |= (uint64_t)(uint32_t)(l - c_lbase[ls]) << nb;
                        nb += le;
                        const int ds = dist_sym(d);
                        v |= (uint64_t)w.dc[ds] << nb;
                        nb += w.dl[ds];
                        const int de = dist_extra(ds);
                        v |= (uint64_t)(uint32_t)(d - c_dbase[ds]) << nb;
                        nb += de;
                    } else {
                        v = w.lc[u];
                        nb = w.ll[u];
                    }
                } else if (t == tc) {
                    v = w.lc[256];
                    nb = w.ll[256];
                }
                // where this token's bits start, counted from the open dword's first bit
                int incl = nb;
                for (int o = 1; o < WAVE; o <<= 1) {
                    const int x = __shfl_up(incl, o, WAVE);
                    if (lane >= o) incl += x;
                }
                const uint32_t total = (uint32_t)__shfl(incl, WAVE - 1, WAVE);
                const uint32_t at = carry_bits + (uint32_t)(incl - nb);
                // the window: the open dword, then what 64 tokens can fill (64 x 48 bits = 96 dwords)
                for (int i = lane; i < 100; i += WAVE) stage[i] = i == 0 ? carry : 0u;
                wave_sync();
                if (nb) {
                    const uint32_t wd = at >> 5, sh = at & 31u;
                    atomicOr(&stage[wd], (uint32_t)(v << sh));
                    const uint64_t rest = sh ? (v >> (32 - sh)) : (v >> 32);      // the bits beyond the first dword
                    if (sh + (uint32_t)nb > 32u) atomicOr(&stage[wd + 1], (uint32_t)rest);
                    if (sh + (uint32_t)nb > 64u) atomicOr(&stage[wd + 2], (uint32_t)(rest >> 32));
                }
                wave_sync();
                const uint32_t bits_now = carry_bits + total;
                const uint32_t full = bits_now >> 5;
                for (uint32_t i = (uint32_t)lane; i < full; i += WAVE) payload[out_word + i] = stage[i];
                carry = stage[full];
                carry_bits = bits_now & 31u;
                out_word += full;
                wave_sync();
            }
            if (lane == 0 && carry_bits) payload[out_word] = carry;
            payload_bytes = out_word * 4u + ((carry_bits + 7u) >> 3);
        }
        payload_bytes = __builtin_amdgcn_readfirstlane(payload_bytes);
        if (payload_bytes == 0) {
            // stored: BFINAL = 1, type 0, LEN, ~LEN, the text (all lanes copy)
            uint8_t* pb = reinterpret_cast<uint8_t*>(payload);
            if (lane == 0) {
                pb[0] = 1;
                pb[1] = (uint8_t)(n & 0xff);
                pb[2] = (uint8_t)(n >> 8);
                pb[3] = (uint8_t)(~n & 0xff);
                pb[4] = (uint8_t)((~n >> 8) & 0xff);
            }
            for (int i = lane; i < n; i += WAVE) pb[5 + i] = text[i];
            payload_bytes = (uint32_t)n + 5u;
        }
        if (lane == 0) a.sizes[m] = payload_bytes;
        wave_sync();
    }
}

// exclusive scan of (sizes[m] + 26) by one workgroup: off[m], off[n] = total
__global__ __launch_bounds__(1024) void k_deflate_scan(const uint32_t* sizes, int n, uint64_t* off) {
    __shared__ uint64_t tot[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    uint64_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += (uint64_t)sizes[i] + 26u;
    tot[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint64_t v = threadIdx.x >= (unsigned)o ? tot[threadIdx.x - o] : 0ull;
        __syncthreads();
        tot[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = threadIdx.x ? tot[threadIdx.x - 1] : 0ull;
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        run += (uint64_t)sizes[i] + 26u;
    }
    if (threadIdx.x == 1023) off[n] = tot[1023];
}

// member m at out + off[m]: the 18-byte BGZF header, the payload, four zero bytes where the host puts the CRC-32, ISIZE
__global__ __launch_bounds__(256) void k_deflate_pack(const uint8_t* slots, const uint32_t* sizes, const uint64_t* off, int64_t n_total,
                                                      uint8_t* out) {
    const int m = blockIdx.x;
    const uint32_t pay = sizes[m];
    uint8_t* dst = out + off[m];
    const uint8_t* src = slots + (size_t)m * DF_SLOT;
    const uint32_t total = pay + 26u;
    const uint32_t isize = (uint32_t)min<int64_t>(DF_MEMBER, n_total - (int64_t)m * DF_MEMBER);
    if (threadIdx.x < 18) {
        const uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                  (uint8_t)((total - 1) & 0xff), (uint8_t)((total - 1) >> 8)};
        dst[threadIdx.x] = head[threadIdx.x];
    }
    for (uint32_t i = threadIdx.x; i < pay; i += 256) dst[18 + i] = src[i];
    if (threadIdx.x < 8) dst[18 + pay + threadIdx.x] = threadIdx.x < 4 ? 0 : (uint8_t)(isize >> (8 * (threadIdx.x - 4)));
}

}  // namespace

namespace trk {

size_t deflate_slot_bytes() { return DF_SLOT; }
size_t deflate_tok_bytes(int n_cu, int n_members) {
    const int wgs = min((n_members + DF_WAVES - 1) / DF_WAVES, n_cu * DF_WGS_PER_CU);
    return (size_t)max(wgs, 1) * DF_WAVES * DF_TOKCAP * sizeof(uint32_t);
}

// text[0 .. n) on the device -> the BGZF members back to back in `out` (capacity: n_members * (DF_SLOT + 26); slots:
// n_members * deflate_slot_bytes()), their
// offsets in off[0 .. n_members] (off[n_members] = total bytes), the CRC-32 fields zero.  slots / sizes / tok: workspace.
hipError_t launch_deflate(const uint8_t* text, int64_t n, uint8_t* slots, uint32_t* sizes, uint32_t* tok, uint64_t* off, uint8_t* out,
                          int n_cu, hipStream_t stream) {
    const int n_members = (int)((n + DF_MEMBER - 1) / DF_MEMBER);
    if (n_members < 1) return hipSuccess;
    DeflArgs a{text, n, n_members, slots, sizes, tok};
    // five workgroups of four members per CU (DF_OCC)
    const int wgs = min((n_members + DF_WAVES - 1) / DF_WAVES, n_cu * DF_WGS_PER_CU);
    hipLaunchKernelGGL(k_deflate_bgzf, dim3(wgs), dim3(WAVE * DF_WAVES), 0, stream, a);
    hipLaunchKernelGGL(k_deflate_scan, dim3(1), dim3(1024), 0, stream, sizes, n_members, off);
    hipLaunchKernelGGL(k_deflate_pack, dim3(n_members), dim3(256), 0, stream, slots, sizes, off, n, out);
    return hipGetLastError();
}

}  // namespace trk
