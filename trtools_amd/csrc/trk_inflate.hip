// trk_inflate.hip -- BGZF blocks inflated on the device (round 5; SURVEY.md section 8(f1): "BGZF block inflate" is part of
// the reader row -- the reference reads through htslib, /root/reference/trtools/utils/utils.py:19-67).
//
// A bgzip'ed VCF is a chain of independent gzip members of at most 64 KiB of text each (RFC 1952 member, 'BC' extra
// subfield with the member's size; RFC 1951 DEFLATE inside, no preset dictionary).  The host finds the members (their
// headers give sizes; include/trk_vcf.h) and hands the raw DEFLATE payloads over; one WAVE inflates one member:
//   * the bit reader, the Huffman decode and the control flow are wave-uniform (every lane computes the same thing: on
//     this machine that is the scalar unit plus broadcast LDS reads -- there is nothing to hand out inside one symbol);
//   * the compressed bytes are read 256 at a time, a dword per lane (loaded when the reader gets there); a dword of the
//     stream is one v_readlane;
//   * the code tables are built by all lanes: counts by LDS atomics, a symbol's rank among the symbols of its length by
//     ballots in symbol order, every lane fills the primary-table slots of its own symbols.  Codes longer than the
//     primary table's index (rare by construction) are decoded by the canonical comparison (first code / count per
//     length, symbols sorted by length), not through secondary tables;
//   * the text goes straight to the buffer in HBM; the last 4 KiB of it are also kept in an LDS ring: a match inside the
//     ring is copied there by the lanes (a period shorter than the length by doubling), a match further back reads the
//     text buffer behind a workgroup-scope fence (the wave's own stores have completed: lanes read other lanes' bytes).
// A member the kernel cannot finish (corrupt stream, more text than its ISIZE, a distance before the member's start) is
// FLAGGED and left to the host inflater; nothing is trusted about the input beyond the bounds the caller gives.
// Roofline note: a serial symbol decode per wave is bound by the latency of its dependent table reads, not by HBM: the
// pass is sized by members in flight (sixteen waves per CU: 8 KiB of LDS and 87 registers per member), not by bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/trk.h"
#include "trk_internal.h"

namespace {

constexpr int LL_ROOT = 9, D_ROOT = 8;    // index bits of the primary tables
constexpr int LL_MAX = 288, D_MAX = 32;

struct InfArgs {
    trk_inflate_in in;
    trk_inflate_out out;
};

struct CodeSet {          // per code (literal/length, distance): canonical description by length
    uint32_t cnt[16];     // symbols of each length
    uint32_t first[16];   // first canonical code of each length
    uint32_t offs[16];    // where the length's symbols start in `sorted`
};

// A wave's LDS: the decode tables and a RING of the last 4 KiB of text.  The text itself goes straight to the buffer in
// HBM; a match whose source lies inside the ring is copied there (LDS is in order for a wave: no fence), a match
// further back reads the text buffer once the wave's own stores up to there have completed (a workgroup-scope fence).
// 8 KiB of LDS per member instead of a 32 KiB window: sixteen members per CU instead of four, and a member is one
// long chain of dependent look-ups that only other members' chains can overlap with (first form, window in LDS:
// 7 ms per member, 9.6 GB/s of text on the whole chip -- tools/inflate_probe.py, profiles/r05_notes.md).
constexpr int RING = 4096;
constexpr int NEAR = RING - 258;          // a match at most this far back is served by the ring alone
constexpr int INF_WAVES = 4;              // members per workgroup (one per wave)

struct Lds {
    uint8_t ring[RING];
    uint32_t ll_tab[1 << LL_ROOT];
    uint32_t d_tab[1 << D_ROOT];
    uint16_t ll_sorted[LL_MAX];
    uint16_t d_sorted[D_MAX];
    uint8_t lens[LL_MAX + D_MAX];
    CodeSet ll, d;
    uint8_t pre_sorted[19], pre_len[19];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// A branch the common symbol does not take must not cost it a TAKEN branch (a jump over the rare block): the rare blocks go
// out of line.  The time of a symbol is its taken branches more than its instructions (tools/inflate_symbol_probe.py).
#define RARE(x) __builtin_expect(!!(x), 0)

// What a symbol MEANS, ready for the decode loop -- the primary tables hold this beside the code's length, so that a
// symbol costs one look-up and no arithmetic on its number:
//   bits 0-3  length of the code word (0: the word is longer than the table's index, or no code word at all)
//   bits 4-7  extra bits that follow it
//   bits 8-9  0 a literal, 1 a length / a distance, 2 end of block, 3 a symbol no valid stream holds
//   bits 16+  the literal, or the base of the length (3 ... 258) / of the distance (1 ... 24577)
constexpr uint32_t E_KIND = 0x300u, E_BASE = 0x100u, E_EOB = 0x200u, E_BAD = 0x300u;
__device__ __forceinline__ uint32_t ll_entry(int sym) {
    if (sym < 256) return (uint32_t)sym << 16;
    if (sym == 256) return E_EOB;
    const int li = sym - 257;
    if (li > 28) return E_BAD;
    if (li < 8) return ((uint32_t)(li + 3) << 16) | E_BASE;
    if (li == 28) return (258u << 16) | E_BASE;
    const int eb = (li >> 2) - 1;
    return ((uint32_t)(((4 + (li & 3)) << eb) + 3) << 16) | E_BASE | ((uint32_t)eb << 4);
}
__device__ __forceinline__ uint32_t d_entry(int ds) {
    if (ds > 29) return E_BAD;
    if (ds < 4) return ((uint32_t)(ds + 1) << 16) | E_BASE;
    const int eb = (ds >> 1) - 1;
    return ((uint32_t)(((2 + (ds & 1)) << eb) + 1) << 16) | E_BASE | ((uint32_t)eb << 4);
}

// the compressed stream of one member: 256 bytes per register, a dword per lane; `next` is the following 256
struct BitReader {
    const uint32_t* base;      // dword-aligned address at or before the payload
    uint32_t n_dwords;         // dwords that may be read (payload + slack the caller guarantees); a payload is < 64 KiB
    uint32_t cur;              // 256 bytes of the stream, a dword per lane
    uint32_t dw;               // next dword to take
    uint64_t bb;               // bit buffer (uniform)
    int bc;                    // valid bits in bb
    int lane;
    int skip;                  // bytes between `base` and the payload's first byte

    __device__ __forceinline__ uint32_t load_reg(uint32_t w) const {
        const uint32_t i = w + (uint32_t)lane;
        typedef const __attribute__((address_space(1))) uint32_t* gptr;
        return i < n_dwords ? __builtin_nontemporal_load((gptr)(uintptr_t)base + i) : 0u;
    }
    __device__ __forceinline__ void init(const uint8_t* p, int64_t n_bytes, int ln) {
        lane = ln;
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        base = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        skip = (int)(a & 3);
        n_dwords = (uint32_t)((n_bytes + skip + 3) / 4 + 2);      // (+ 8 bytes: the member's CRC32 / ISIZE trailer follows the payload)
        seek(0);
    }
    // continue at byte `off` of the payload
    __device__ __forceinline__ void seek(int64_t off) {
        const uint32_t b = (uint32_t)off + (uint32_t)skip;
        dw = b >> 2;
        cur = (dw & 63u) ? load_reg(dw & ~63u) : 0u;     // (lane k holds dword (dw & ~63) + k; at a window's start take_dword loads it)
        bb = 0;
        bc = 0;
        refill();
        const int sk = 8 * (int)(b & 3);
        bb >>= sk;
        bc -= sk;
    }
    __device__ __forceinline__ uint32_t take_dword() {
        // (uniform) the next 256 bytes.  Loaded when needed, not ahead: a register that is in flight across the symbol
        // loop makes every use of the reader wait for ALL the wave's memory operations -- its stores of text included --
        // at every dword (one counter, vmcnt, for loads and stores); this way the wave waits once per 256 bytes
        if (RARE((dw & 63u) == 0u)) {
            cur = load_reg(dw);
            asm volatile("" : "+v"(cur));      // (a use HERE: the wait for the load stays on this path, not on every dword's)
        }
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(dw & 63u));
        ++dw;
        return v;
    }
    __device__ __forceinline__ void refill() {     // at least 33 valid bits afterwards
        while (bc <= 32) {
            bb |= (uint64_t)take_dword() << bc;
            bc += 32;
        }
    }
    // the same where bc >= 0 is known (everywhere but after seek): one dword is enough
    __device__ __forceinline__ void refill1() {
        if (bc <= 32) {
            bb |= (uint64_t)take_dword() << bc;
            bc += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) {
        bb >>= n;
        bc -= n;
    }
    __device__ __forceinline__ uint32_t take(int n) {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    // bits consumed since the payload's first byte
    __device__ __forceinline__ int64_t bits_used() const { return (int64_t)dw * 32 - bc - 8 * skip; }
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __builtin_bitreverse32(v) >> (32 - n); }

// Build the decode structures of one code from lens[0 .. n): counts, first codes, sorted symbols, primary table.
// Returns false (uniform) for an over-subscribed code, or an incomplete one that zlib would refuse (anything but a
// single one-bit code or an empty code).
template <int ROOT, bool DIST>
__device__ __noinline__ bool build_code(const uint8_t* lens, int n, CodeSet& cs, uint32_t* tab, uint16_t* sorted, int lane) {
    if (lane < 16) cs.cnt[lane] = 0;
    for (int i = lane; i < (1 << ROOT); i += 64) tab[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < n; s += 64) {
        const int l = lens[s];
        if (l) atomicAdd(&cs.cnt[l], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    // first code and offset per length: fifteen uniform steps over the counts (lane q keeps what belongs to length q)
    uint32_t code = 0, off = 0, total = 0, my_first = 0, my_off = 0;
    int left = 1, max_len = 0;
    bool over = false;
#pragma unroll 1
    for (int l = 1; l <= 15; ++l) {
        const uint32_t c = uni(cs.cnt[l]);
        left = (left << 1) - (int)c;
        over |= left < 0;
        if (lane == l) {
            my_first = code;
            my_off = off;
        }
        code = (code + c) << 1;
        off += c;
        total += c;
        if (c) max_len = l;
    }
    if (over) return false;
    if (left > 0 && !(total == 0 || (total == 1 && max_len == 1))) return false;
    if (lane < 16) {
        cs.first[lane] = my_first;
        cs.offs[lane] = my_off;
    }
    __builtin_amdgcn_wave_barrier();
    // a symbol's rank among the symbols of its length, in symbol order: ballots over chunks of 64 symbols; lane q of
    // `run` counts the symbols of length q met so far
    uint32_t run = 0;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        const int l = s < n ? (int)lens[s] : 0;
        uint32_t rank = 0, add = 0;
#pragma unroll 1
        for (int q = 1; q <= 15; ++q) {
            const uint64_t m = __ballot(l == q);
            if (l == q) rank = (uint32_t)__popcll(m & lt);
            if (lane == q) add = (uint32_t)__popcll(m);
        }
        const uint32_t before = (uint32_t)__shfl((int)run, l);       // the count of MY length before this chunk
        run += add;
        if (l) {
            rank += before;
            const uint32_t fc = cs.first[l], fo = cs.offs[l];
            sorted[fo + rank] = (uint16_t)s;
            if (l <= ROOT) {
                const uint32_t e = (DIST ? d_entry(s) : ll_entry(s)) | (uint32_t)l;
                for (uint32_t k = rev_bits(fc + rank, l); k < (1u << ROOT); k += 1u << l) tab[k] = e;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// A code word the primary table does not hold (uniform; the rare path): longer than ROOT bits, or -- in an incomplete
// code -- a word of at most ROOT bits that is unused.  By the canonical comparison: first code / count per length,
// symbols sorted by length.  `bits` = the stream's next 15 bits or more.  Returns (symbol << 4) | length of the word, or
// -1 for bits that are no code word.
template <int ROOT>
__device__ __noinline__ int decode_long(uint32_t bits, const CodeSet& cs, const uint16_t* sorted) {
    const uint32_t r = __builtin_bitreverse32(bits);    // first bit on top
#pragma unroll 1
    for (int k = 0; k < 15; ++k) {
        const int l = k < 15 - ROOT ? ROOT + 1 + k : k - (15 - ROOT) + 1;      // ROOT + 1 ... 15, then 1 ... ROOT
        const uint32_t c = r >> (32 - l);
        const uint32_t idx = c - uni(cs.first[l]);
        if (idx < uni(cs.cnt[l])) return ((int)uni(sorted[uni(cs.offs[l]) + idx]) << 4) | l;
    }
    return -1;
}

__global__ __launch_bounds__(64 * INF_WAVES) void k_inflate_bgzf(const InfArgs a) {
    __shared__ Lds s_all[INF_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Lds& s = s_all[wave];
    for (int blk = blockIdx.x * INF_WAVES + wave; blk < a.in.n_blocks; blk += gridDim.x * INF_WAVES) {
        const int64_t in_off = a.in.in_off[blk];
        const int32_t in_len = a.in.in_len[blk];
        const int32_t out_len = a.in.out_len[blk];
        uint8_t* dst = a.out.text + a.in.out_off[blk];
        uint32_t err = 0;
        if (in_len < 0 || out_len < 0 || out_len > 65536 || in_off < 0 || in_off + in_len > a.in.n_comp_bytes) err = TRK_INFLATE_INPUT;
        int pos = 0;          // bytes of text produced (the pending literals included)
        // literals wait in a register, lane k the k-th of the run, and leave together -- one ring write and one store per
        // run instead of per byte (the per-symbol vector instructions are what bounds this kernel: profiles/r05_sq_inflate.txt)
        uint32_t pend = 0;
        int npend = 0;
        // (pos may run past out_len by the literals that wait: they are not stored then, and the member is flagged)
        auto flush_literals = [&]() {
            if (npend) {
                if (pos > out_len) err = TRK_INFLATE_OVERRUN;
                else if (lane < npend) {
                    const uint32_t o = (uint32_t)(pos - npend) + (uint32_t)lane;
                    s.ring[o & (RING - 1)] = (uint8_t)pend;
                    dst[o] = (uint8_t)pend;
                }
                npend = 0;
            }
        };
        if (!err && out_len > 0) {
            BitReader br;
            const uint8_t* p = a.in.comp + in_off;
            br.init(p, in_len, lane);
            const int64_t bit_limit = (int64_t)in_len * 8;
            // the symbol loop looks at the dword counter only (the exact test follows the last block): a stream that runs
            // past its payload reads zeros, and zeros end in an error or at out_len whatever they decode to
            const uint32_t dw_limit = ((uint32_t)br.skip + (uint32_t)in_len + 3u) / 4u + 4u;
            bool last = false;
            while (!last && !err) {
                br.refill();
                last = br.take(1) != 0;
                const uint32_t type = br.take(2);
                if (type == 0) {
                    // stored: to the next byte boundary, LEN / NLEN, LEN raw bytes
                    flush_literals();
                    br.drop(br.bc & 7);
                    br.refill();
                    const uint32_t len = br.take(16);
                    br.refill();
                    const uint32_t nlen = br.take(16);
                    if ((len ^ nlen) != 0xffffu) { err = TRK_INFLATE_STREAM; break; }
                    if (pos + (int)len > out_len) { err = TRK_INFLATE_OVERRUN; break; }
                    const int64_t byte0 = br.bits_used() / 8;         // payload offset of the first raw byte
                    if (byte0 + (int64_t)len > (int64_t)in_len) { err = TRK_INFLATE_STREAM; break; }
                    for (int i = lane; i < (int)len; i += 64) {
                        const uint8_t b = p[byte0 + i];
                        dst[pos + i] = b;
                        if ((int)len - i <= RING) s.ring[(pos + i) & (RING - 1)] = b;     // (the run's last RING bytes)
                    }
                    __builtin_amdgcn_wave_barrier();
                    pos += (int)len;
                    br.seek(byte0 + len);
                    continue;
                }
                if (type == 3) { err = TRK_INFLATE_STREAM; break; }
                int n_ll, n_d;
                if (type == 1) {
                    // fixed code: lengths 8 / 9 / 7 / 8 for the literal / length alphabet, 5 for the 30 distances
                    for (int i = lane; i < LL_MAX; i += 64) s.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
                    if (lane < 32) s.lens[LL_MAX + lane] = 5;
                    n_ll = 288;
                    n_d = 32;     // (30 and 31 never occur in a valid stream: met, they are an error below)
                    __builtin_amdgcn_wave_barrier();
                } else {
                    br.refill();
                    n_ll = (int)br.take(5) + 257;
                    n_d = (int)br.take(5) + 1;
                    const int n_pre = (int)br.take(4) + 4;
                    if (n_ll > 286 || n_d > 30) { err = TRK_INFLATE_STREAM; break; }
                    // code lengths of the code-length alphabet, in their transmitted order
                    if (lane < 19) s.pre_len[lane] = 0;
                    __builtin_amdgcn_wave_barrier();
                    for (int i = 0; i < n_pre; ++i) {
                        br.refill();
                        const uint32_t v = br.take(3);
                        // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                        const int sym = i < 3 ? 16 + i : i == 3 ? 0 : ((i & 1) ? 7 - ((i - 5) >> 1) : 8 + ((i - 4) >> 1));
                        if (lane == 0) s.pre_len[sym] = (uint8_t)v;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // the code-length code: at most 7 bits, decoded by the canonical comparison alone
                    uint32_t pc[8], pf[8], po[8];
                    {
                        const int l = lane < 19 ? (int)s.pre_len[lane] : 0;
                        uint32_t code = 0, off = 0;
                        int left = 1;
                        bool over = false;
                        uint32_t total = 0;
                        int max_len = 0;
                        uint32_t rank = 0, fo = 0;
                        const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
                        for (int q = 1; q <= 7; ++q) {
                            const uint64_t m = __ballot(l == q);
                            const uint32_t c = (uint32_t)__popcll(m);
                            left = (left << 1) - (int)c;
                            over |= left < 0;
                            pc[q] = c;
                            pf[q] = code;
                            po[q] = off;
                            if (l == q) {
                                rank = (uint32_t)__popcll(m & lt);
                                fo = off;
                            }
                            code = (code + c) << 1;
                            off += c;
                            total += c;
                            if (c) max_len = q;
                        }
                        if (over || (left > 0 && !(total == 1 && max_len == 1))) { err = TRK_INFLATE_STREAM; break; }
                        if (l) s.pre_sorted[fo + rank] = (uint8_t)lane;
                        __builtin_amdgcn_wave_barrier();
                    }
                    // the literal / length and distance code lengths, run-length coded
                    int i = 0;
                    uint32_t prev = 0;
                    const int n_all = n_ll + n_d;
                    while (i < n_all && !err) {
                        br.refill();
                        const uint32_t r = __builtin_bitreverse32((uint32_t)br.bb);
                        int sym = -1;
#pragma unroll
                        for (int l = 1; l <= 7; ++l) {
                            const uint32_t idx = (r >> (32 - l)) - pf[l];
                            if (sym < 0 && idx < pc[l]) {
                                sym = (int)uni(s.pre_sorted[po[l] + idx]);
                                br.drop(l);
                            }
                        }
                        if (sym < 0) { err = TRK_INFLATE_STREAM; break; }
                        if (sym < 16) {
                            if (lane == 0) s.lens[i < n_ll ? i : LL_MAX + (i - n_ll)] = (uint8_t)sym;
                            prev = (uint32_t)sym;
                            ++i;
                        } else {
                            uint32_t rep, val = 0;
                            if (sym == 16) {
                                if (i == 0) { err = TRK_INFLATE_STREAM; break; }
                                rep = 3 + br.take(2);
                                val = prev;
                            } else if (sym == 17) {
                                rep = 3 + br.take(3);
                            } else {
                                rep = 11 + br.take(7);
                            }
                            if (i + (int)rep > n_all) { err = TRK_INFLATE_STREAM; break; }
                            for (int k = lane; k < (int)rep; k += 64) {
                                const int q = i + k;
                                s.lens[q < n_ll ? q : LL_MAX + (q - n_ll)] = (uint8_t)val;
                            }
                            if (sym != 16) prev = 0;
                            i += (int)rep;
                        }
                    }
                    if (err) break;
                    __builtin_amdgcn_wave_barrier();
                    if (uni(s.lens[256]) == 0) { err = TRK_INFLATE_STREAM; break; }     // no end-of-block code
                }
                // (the result of a call that is not inlined is "divergent" to the compiler, and a divergent exit from this
                // loop would move the whole bit reader into vector registers: 92 vector instructions per symbol instead
                // of a dozen -- readfirstlane says what it is)
                const bool ok_ll = uni(build_code<LL_ROOT, false>(s.lens, n_ll, s.ll, s.ll_tab, s.ll_sorted, lane) ? 1u : 0u) != 0;
                const bool ok_d = uni(build_code<D_ROOT, true>(s.lens + LL_MAX, n_d, s.d, s.d_tab, s.d_sorted, lane) ? 1u : 0u) != 0;
                if (!ok_ll || !ok_d) {
                    err = TRK_INFLATE_STREAM;
                    break;
                }
                // ---- the symbols of this block ----
                // One look-up per symbol: the table's entry says what the symbol means (ll_entry / d_entry).  At the top
                // the reader holds >= 33 bits: a literal / length word and its extra bits take at most 20, a distance
                // word and its extra bits at most 28 after one more dword.
                for (;;) {
                    br.refill1();
                    uint32_t e = uni(s.ll_tab[(uint32_t)br.bb & ((1u << LL_ROOT) - 1u)]);
                    if (RARE(!(e & 15u))) {
                        const int sl = (int)uni((uint32_t)decode_long<LL_ROOT>((uint32_t)br.bb, s.ll, s.ll_sorted));
                        if (RARE(sl < 0)) { err = TRK_INFLATE_STREAM; break; }
                        e = ll_entry(sl >> 4) | (uint32_t)(sl & 15);
                    }
                    br.drop((int)(e & 15u));
                    if (!(e & E_KIND)) {
                        pend = lane == npend ? e >> 16 : pend;
                        ++npend;
                        ++pos;
                        if (RARE(npend == 64)) {
                            flush_literals();
                            if (err || br.dw > dw_limit) { err = err ? err : TRK_INFLATE_STREAM; break; }
                        }
                    } else {
                        flush_literals();
                        if (RARE(err != 0)) break;
                        if (RARE((e & E_EOB) != 0)) {              // end of block, or a symbol that is none
                            if ((e & E_KIND) == E_BAD) err = TRK_INFLATE_STREAM;
                            break;
                        }
                        const int leb = (int)((e >> 4) & 15u);
                        const int len = (int)(e >> 16) + (int)br.take(leb);
                        br.refill1();
                        uint32_t d = uni(s.d_tab[(uint32_t)br.bb & ((1u << D_ROOT) - 1u)]);
                        if (RARE(!(d & 15u))) {
                            const int sl = (int)uni((uint32_t)decode_long<D_ROOT>((uint32_t)br.bb, s.d, s.d_sorted));
                            if (RARE(sl < 0)) { err = TRK_INFLATE_STREAM; break; }
                            d = d_entry(sl >> 4) | (uint32_t)(sl & 15);
                        }
                        br.drop((int)(d & 15u));
                        if (RARE((d & E_KIND) != E_BASE)) { err = TRK_INFLATE_STREAM; break; }
                        const int deb = (int)((d >> 4) & 15u);
                        const int dist = (int)(d >> 16) + (int)br.take(deb);
                        if (RARE(dist > pos)) { err = TRK_INFLATE_STREAM; break; }
                        if (RARE(pos + len > out_len)) { err = TRK_INFLATE_OVERRUN; break; }
                        if (!RARE(dist > NEAR)) {
                            // inside the ring: lanes take bytes; a period shorter than what is left doubles as the copy proceeds
                            int done = 0, d_eff = dist;
                            if (len <= 64 && dist >= len) {     // (most matches: one step, source and destination apart)
                                if (lane < len) {
                                    const uint32_t o = (uint32_t)pos + (uint32_t)lane;
                                    const uint8_t b = s.ring[(o - (uint32_t)dist) & (RING - 1)];
                                    s.ring[o & (RING - 1)] = b;
                                    dst[o] = b;
                                }
                                done = len;
                            }
                            while (done < len) {
                                const int n = min(min(len - done, 64), d_eff);
                                if (lane < n) {
                                    const uint32_t o = (uint32_t)(pos + done) + (uint32_t)lane;
                                    const uint8_t b = s.ring[(o - (uint32_t)d_eff) & (RING - 1)];
                                    s.ring[o & (RING - 1)] = b;
                                    dst[o] = b;
                                }
                                done += n;
                                if (n == d_eff && d_eff < 64) d_eff *= 2;
                            }
                        } else {
                            // further back than the ring: from the text buffer (no overlap: the distance exceeds the length)
                            // Lanes read what OTHER lanes of this wave stored: a fence of workgroup scope -- the wave's
                            // stores have reached the CU's L1 / the L2 it reads through (s_waitcnt vmcnt(0)); nothing
                            // leaves this CU.  (Agent scope here wrote the whole L2 back, `buffer_wbl2`, per fence.)
                            // (every far match: they are few -- a VCF's repeats lie within the ring -- and keeping track
                            // of what is fenced was one more value carried around the symbol loop)
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                            for (int done = 0; done < len; done += 64) {
                                if (done + lane < len) {
                                    const uint32_t o = (uint32_t)(pos + done) + (uint32_t)lane;
                                    const uint8_t b = dst[o - (uint32_t)dist];
                                    s.ring[o & (RING - 1)] = b;
                                    dst[o] = b;
                                }
                            }
                        }
                        pos += len;
                        if (RARE(br.dw > dw_limit)) { err = TRK_INFLATE_STREAM; break; }
                    }
                }
            }
            if (!err && br.bits_used() > bit_limit) err = TRK_INFLATE_STREAM;     // read beyond the payload
        }
        if (!err && pos != out_len) err = TRK_INFLATE_OVERRUN;
        if (lane == 0) a.out.flags[blk] = (uint8_t)err;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the line index of inflated text, and the heads of its lines (the reader's inflate hook, trk_api.hip) ----------
// What the host side of a batch reads of a 60 KB record is its first hundred bytes: CHROM ... FORMAT.  With the text
// inflated in HBM the host gets (a) where the newlines are and (b) those heads, packed; the sample columns never cross
// PCIe as text.  Three steps over a segment of text: newlines counted per 16 KB tile, scanned, scattered in order
// (k_nl_*); one wave per line looks for the line's ninth tab (k_line_heads); head lengths scanned, heads gathered.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NL_TILE = 16384;     // bytes per workgroup of the newline passes: thread t owns 64 consecutive bytes

__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t pattern) {   // 0x80 in every byte of x equal to the pattern's
    const uint32_t t = x ^ pattern;
    return ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
}

// the 64 bytes of thread `tid` of tile `tile` as sixteen words, bytes at or beyond n zeroed
__device__ __forceinline__ void load_slice(const uint8_t* text, int64_t n, int64_t at, uint32_t (&w)[16]) {
    const u32x4* p = reinterpret_cast<const u32x4*>(text + at);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32x4 v = {0, 0, 0, 0};
        if (at + 16 * k < n) v = p[k];      // (the segment is padded: a vector that starts inside it is readable)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t b = at + 16 * k + 4 * j;
            uint32_t x = v[j];
            if (b + 4 > n) x = b >= n ? 0u : (x & (0xffffffffu >> (8 * (int)(b + 4 - n))));
            w[4 * k + j] = x;
        }
    }
}

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t* text, int64_t n, uint32_t* counts) {
    __shared__ uint32_t part[4];
    const int64_t at = (int64_t)blockIdx.x * NL_TILE + threadIdx.x * 64;
    uint32_t c = 0;
    if (at < n) {
        uint32_t w[16];
        load_slice(text, n, at, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) c += __popc(eq_bytes(w[k], 0x0a0a0a0au));
    }
    for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of counts[0 .. n) in place by ONE workgroup; total to *total
__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t* counts, int n, uint32_t* total) {
    __shared__ uint32_t tot[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    tot[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)o ? tot[threadIdx.x - o] : 0u;
        __syncthreads();
        tot[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? tot[threadIdx.x - 1] : 0u;
    for (int i = lo; i < hi; ++i) {
        const uint32_t c = counts[i];
        counts[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) *total = tot[1023];
}

__global__ __launch_bounds__(256) void k_nl_scatter(const uint8_t* text, int64_t n, const uint32_t* offs, uint64_t* nl, uint32_t cap) {
    __shared__ uint32_t cnt[256];
    const int64_t at = (int64_t)blockIdx.x * NL_TILE + threadIdx.x * 64;
    uint32_t w[16];
    uint32_t c = 0;
    if (at < n) {
        load_slice(text, n, at, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) c += __popc(eq_bytes(w[k], 0x0a0a0a0au));
    }
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)o ? cnt[threadIdx.x - o] : 0u;
        __syncthreads();
        cnt[threadIdx.x] += v;
        __syncthreads();
    }
    if (!c) return;
    uint32_t at_out = offs[blockIdx.x] + cnt[threadIdx.x] - c;
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
        uint32_t z = eq_bytes(w[k], 0x0a0a0a0au);
        while (z) {
            const int b = (__ffs(z) - 1) >> 3;
            z &= z - 1;
            const int64_t pos = at + 4 * k + b;
            if (at_out < cap) nl[at_out] = (uint64_t)pos | ((pos > 0 && text[pos - 1] == '\r') ? (1ull << 63) : 0ull);
            ++at_out;
        }
    }
}

// One wave per line.  Line i = [start, end): start = 0 / nl[i - 1] + 1, end = nl[i] / n (the last, unfinished one).
// The head of a line ends behind its ninth tab (the first line: behind tab 9 - tabs_in of what this segment holds of
// it); a line with fewer tabs is all head.  head_len[i] bytes from head_off[i]; *state_out: the tabs (0 ... 9) seen
// in the unfinished last line.
__global__ __launch_bounds__(64) void k_line_heads(const uint8_t* text, int64_t n, const uint64_t* nl, const uint32_t* n_nl_p,
                                                   int tabs_in, uint64_t* head_off, uint32_t* head_len, int32_t* state_out,
                                                   uint32_t cap) {
    const uint32_t n_nl = min(*n_nl_p, cap);     // (the tables hold cap + 1 lines; the host sizes them from the count and
    const int lane = threadIdx.x;                //  launches this after it has read it -- the clamp is the belt to that)
    for (uint32_t i = blockIdx.x; i <= n_nl; i += gridDim.x) {
        const int64_t start = i == 0 ? 0 : (int64_t)(nl[i - 1] & ~(1ull << 63)) + 1;
        const int64_t end = i < n_nl ? (int64_t)(nl[i] & ~(1ull << 63)) : n;
        const int carried = i == 0 ? tabs_in : 0;
        const int need = 9 - carried;
        int found = 0;
        int64_t head_end = end;
        if (need <= 0) {
            head_end = start;
        } else {
            for (int64_t base = start & ~(int64_t)15; base < end && found < need; base += 1024) {
                const int64_t at = base + 16 * lane;
                uint32_t m = 0;                                  // bit b: byte at + b is a tab inside [start, end)
                if (at < end) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(text + at);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint32_t z = eq_bytes(v[j], 0x09090909u);
                        m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * j);
                    }
                    if (at < start) m &= 0xffffu << (int)(start - at);
                    if (at + 16 > end) m &= 0xffffu >> (int)(at + 16 - end);
                    m &= 0xffffu;
                }
                int c = __popc(m), inc = c;
                for (int o = 1; o < 64; o <<= 1) {
                    const int v2 = __shfl_up(inc, o);
                    if (lane >= o) inc += v2;
                }
                const int total = __shfl(inc, 63);
                if (found + total >= need) {
                    // the lane that holds tab number `need`: the k-th set bit of its mask
                    const bool mine = found + inc >= need && found + inc - c < need;
                    int64_t he = 0;
                    if (mine) {
                        int k = need - (found + inc - c);        // 1-based among this lane's tabs
                        uint32_t mm = m;
                        while (--k) mm &= mm - 1;
                        he = at + (__ffs(mm) - 1) + 1;
                    }
                    const uint64_t who = __ballot(mine);
                    const int src = __ffsll((unsigned long long)who) - 1;
                    head_end = __shfl(he, src);
                    found = need;
                } else {
                    found += total;
                }
            }
        }
        if (lane == 0) {
            head_off[i] = (uint64_t)start;
            head_len[i] = (uint32_t)(head_end - start);
            if (i == n_nl) {
                state_out[0] = min(9, carried + found);
                state_out[1] = n > 0 ? (int32_t)text[n - 1] : -1;     // the run's last byte: a '\r' there belongs to the
            }                                                          // newline that opens the next run
        }
    }
}

// heads gathered back to back: pack_off = exclusive scan of head_len
__global__ __launch_bounds__(64) void k_head_gather(const uint8_t* text, const uint64_t* head_off, const uint32_t* head_len,
                                                    const uint32_t* pack_off, const uint32_t* n_nl_p, uint8_t* packed, uint32_t cap,
                                                    uint32_t nl_cap) {
    const uint32_t n_lines = min(*n_nl_p, nl_cap) + 1;
    for (uint32_t i = blockIdx.x; i < n_lines; i += gridDim.x) {
        const uint8_t* src = text + head_off[i];
        const uint32_t len = head_len[i], o = pack_off[i];
        for (uint32_t k = threadIdx.x; k < len; k += 64)
            if (o + k < cap) packed[o + k] = src[k];
    }
}

// exclusive scan of v[0 .. *n_nl_p] in place (one workgroup), total to *total
__global__ __launch_bounds__(1024) void k_scan_lines(uint32_t* v, const uint32_t* n_nl_p, uint32_t* total, uint32_t nl_cap) {
    __shared__ uint32_t tot[1024];
    const int n = (int)min(*n_nl_p, nl_cap) + 1;
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += v[i];
    tot[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t x = threadIdx.x >= (unsigned)o ? tot[threadIdx.x - o] : 0u;
        __syncthreads();
        tot[threadIdx.x] += x;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? tot[threadIdx.x - 1] : 0u;
    for (int i = lo; i < hi; ++i) {
        const uint32_t c = v[i];
        v[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) *total = tot[1023];
}

__global__ void k_copy_u32(const uint32_t* src, uint32_t* dst, const uint32_t* n_nl_p, uint32_t nl_cap) {   // dst[i] = src[i], i <= n_nl
    const uint32_t n = min(*n_nl_p, nl_cap) + 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

namespace trk {
hipError_t launch_inflate(const trk_inflate_in& in, const trk_inflate_out& out, int n_cu, hipStream_t stream, int wgs_per_cu) {
    if (in.n_blocks <= 0) return hipSuccess;
    InfArgs a{in, out};
    // members walk the grid: sixteen waves per CU when the registers allow (four workgroups of four members)
    const int wgs = (in.n_blocks + INF_WAVES - 1) / INF_WAVES;
    int per_cu = wgs_per_cu >= 1 && wgs_per_cu <= 4 ? wgs_per_cu : 4;
    if (const char* o = trk_opt("TRK_INFLATE_WGS_PER_CU")) per_cu = atoi(o) > 0 ? atoi(o) : per_cu;
    const int grid = wgs < n_cu * per_cu ? wgs : n_cu * per_cu;
    hipLaunchKernelGGL(k_inflate_bgzf, dim3(grid), dim3(64 * INF_WAVES), 0, stream, a);
    return hipGetLastError();
}

// The line index of text[0 .. n) and the heads of its lines (see k_line_heads), in two steps so that the tables are
// sized from the COUNT and not from a guess about the text (lines of fewer than 16 bytes -- blank lines, text that is
// no VCF -- used to overrun a workspace sized total / 16: ADVICE r05).  All results stay on the device:
//   ws.counts [n_tiles + 1] scratch, ws.n_nl (one word), ws.nl [nl_cap], ws.head_off / head_len / pack_off [nl_cap + 1],
//   ws.head_total (one word), ws.state (two words: tabs of the unfinished last line, the text's last byte),
//   ws.packed [packed_cap]
// Step 1: the newlines per 16 KB tile and their total in *ws.n_nl (the caller reads it and makes sure nl_cap >= it).
hipError_t launch_line_count(const uint8_t* text, int64_t n, const LineIndexWs& ws, hipStream_t stream) {
    const int n_tiles = (int)((n + NL_TILE - 1) / NL_TILE);
    if (n_tiles < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_nl_count, dim3(n_tiles), dim3(256), 0, stream, text, n, ws.counts);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, stream, ws.counts, n_tiles, ws.n_nl);
    return hipGetLastError();
}
// Step 2: everything else.  Every kernel clamps the line count to ws.nl_cap.
hipError_t launch_line_index(const uint8_t* text, int64_t n, int tabs_in, const LineIndexWs& ws, hipStream_t stream) {
    const int n_tiles = (int)((n + NL_TILE - 1) / NL_TILE);
    if (n_tiles < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_nl_scatter, dim3(n_tiles), dim3(256), 0, stream, text, n, ws.counts, ws.nl, ws.nl_cap);
    const int lines_grid = 4096;
    hipLaunchKernelGGL(k_line_heads, dim3(lines_grid), dim3(64), 0, stream, text, n, ws.nl, ws.n_nl, tabs_in, ws.head_off,
                       ws.head_len, ws.state, ws.nl_cap);
    // pack_off = exclusive scan of head_len (a copy, scanned in place by one workgroup)
    hipLaunchKernelGGL(k_copy_u32, dim3(256), dim3(256), 0, stream, ws.head_len, ws.pack_off, ws.n_nl, ws.nl_cap);
    hipLaunchKernelGGL(k_scan_lines, dim3(1), dim3(1024), 0, stream, ws.pack_off, ws.n_nl, ws.head_total, ws.nl_cap);
    hipLaunchKernelGGL(k_head_gather, dim3(lines_grid), dim3(64), 0, stream, text, ws.head_off, ws.head_len, ws.pack_off, ws.n_nl,
                       ws.packed, ws.packed_cap, ws.nl_cap);
    return hipGetLastError();
}
}  // namespace trk
