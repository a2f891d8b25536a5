// trk_inflate.hip -- BGZF blocks inflated on the device (round 5; SURVEY.md section 8(f1): "BGZF block inflate" is part of
// the reader row -- the reference reads through htslib, /root/reference/trtools/utils/utils.py:19-67).
//
// A bgzip'ed VCF is a chain of independent gzip members of at most 64 KiB of text each (RFC 1952 member, 'BC' extra
// subfield with the member's size; RFC 1951 DEFLATE inside, no preset dictionary).  The host finds the members (their
// headers give sizes; include/trk_vcf.h) and hands the raw DEFLATE payloads over; one WAVE inflates one member:
//   * the bit reader, the Huffman decode and the control flow are wave-uniform (every lane computes the same thing: on
//     this machine that is the scalar unit plus broadcast LDS reads -- there is nothing to hand out inside one symbol);
//   * the compressed bytes are read 256 at a time, a dword per lane, the next 256 already in flight; a dword of the
//     stream is one v_readlane;
//   * the code tables are built by all lanes: counts by LDS atomics, a symbol's rank among the symbols of its length by
//     ballots in symbol order, every lane fills the primary-table slots of its own symbols.  Codes longer than the
//     primary table's index (rare by construction) are decoded by the canonical comparison (first code / count per
//     length, symbols sorted by length), not through secondary tables;
//   * the text is assembled in a 32 KiB LDS window (DEFLATE's maximum match distance), matches are copied by the lanes
//     (a period shorter than the length by doubling), and the window is written out to the text buffer in HBM as it
//     fills -- 64 bytes per instruction.
// A member the kernel cannot finish (corrupt stream, more text than its ISIZE, a distance before the member's start) is
// FLAGGED and left to the host inflater; nothing is trusted about the input beyond the bounds the caller gives.
// Roofline note: a serial symbol decode per wave is bound by the latency of its dependent table reads, not by HBM: the
// pass is sized by members in flight (four waves per CU by the LDS window), not by bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trk.h"
#include "trk_internal.h"

namespace {

constexpr int INF_WIN = 32768;            // window bytes (power of two >= 32768)
constexpr int LL_ROOT = 10, D_ROOT = 8;   // index bits of the primary tables
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

struct Lds {
    uint8_t win[INF_WIN];
    uint16_t ll_tab[1 << LL_ROOT];
    uint16_t d_tab[1 << D_ROOT];
    uint16_t ll_sorted[LL_MAX];
    uint16_t d_sorted[D_MAX];
    uint8_t lens[LL_MAX + D_MAX];
    CodeSet ll, d;
    uint32_t pre_cnt[8], pre_first[8], pre_offs[8];
    uint8_t pre_sorted[19], pre_len[19];
    uint32_t err;
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// the compressed stream of one member: 256 bytes per register, a dword per lane; `next` is the following 256
struct BitReader {
    const uint32_t* base;      // dword-aligned address at or before the payload
    int64_t n_dwords;          // dwords that may be read (payload + slack the caller guarantees)
    int64_t w0;                // dword index held by lane 0 of `cur`
    uint32_t cur, next;
    int64_t dw;                // next dword to take
    uint64_t bb;               // bit buffer (uniform)
    int bc;                    // valid bits in bb
    int lane;

    __device__ __forceinline__ uint32_t load_reg(int64_t w) const {
        const int64_t i = w + lane;
        return i < n_dwords ? __builtin_nontemporal_load(base + i) : 0u;
    }
    int skip;                  // bytes between `base` and the payload's first byte
    __device__ __forceinline__ void init(const uint8_t* p, int64_t n_bytes, int ln) {
        lane = ln;
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        base = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        skip = (int)(a & 3);
        n_dwords = (n_bytes + skip + 3) / 4 + 2;      // (+ 8 bytes: the member's CRC32 / ISIZE trailer follows the payload)
        seek(0);
    }
    // continue at byte `off` of the payload
    __device__ __forceinline__ void seek(int64_t off) {
        const int64_t b = off + skip;
        dw = b >> 2;
        w0 = dw & ~(int64_t)63;
        cur = load_reg(w0);
        next = load_reg(w0 + 64);
        bb = 0;
        bc = 0;
        refill();
        const int sk = 8 * (int)(b & 3);
        bb >>= sk;
        bc -= sk;
    }
    __device__ __forceinline__ uint32_t take_dword() {
        if (dw >= w0 + 64) {           // (uniform) the next register becomes the current one
            cur = next;
            w0 += 64;
            next = load_reg(w0 + 64);
        }
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(dw - w0));
        ++dw;
        return v;
    }
    __device__ __forceinline__ void refill() {     // at least 33 valid bits afterwards
        while (bc <= 32) {
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
    __device__ __forceinline__ int64_t bits_used() const { return dw * 32 - bc - 8 * skip; }
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __builtin_bitreverse32(v) >> (32 - n); }

// Build the decode structures of one code from lens[0 .. n): counts, first codes, sorted symbols, primary table.
// Returns false (uniform) for an over-subscribed code, or an incomplete one that zlib would refuse (anything but a
// single one-bit code or an empty code).
template <int ROOT>
__device__ bool build_code(const uint8_t* lens, int n, CodeSet& cs, uint16_t* tab, uint16_t* sorted, int lane) {
    if (lane < 16) cs.cnt[lane] = 0;
    for (int i = lane; i < (1 << ROOT); i += 64) tab[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < n; s += 64) {
        const int l = lens[s];
        if (l) atomicAdd(&cs.cnt[l], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t code = 0, off = 0, total = 0;
    int left = 1, max_len = 0;
    bool over = false;
    uint32_t firstv[16], offv[16];
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const uint32_t c = uni(cs.cnt[l]);
        left = (left << 1) - (int)c;
        over |= left < 0;
        firstv[l] = code;
        offv[l] = off;
        code = (code + c) << 1;
        off += c;
        total += c;
        if (c) max_len = l;
    }
    if (over) return false;
    if (left > 0 && !(total == 0 || (total == 1 && max_len == 1))) return false;
    if (lane >= 1 && lane < 16) {
        cs.first[lane] = 0;
        cs.offs[lane] = 0;
    }
#pragma unroll
    for (int l = 1; l <= 15; ++l)
        if (lane == l) {
            cs.first[l] = firstv[l];
            cs.offs[l] = offv[l];
        }
    // a symbol's rank among the symbols of its length, in symbol order: ballots over chunks of 64 symbols
    uint32_t run[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) run[l] = 0;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        const int l = s < n ? (int)lens[s] : 0;
        uint32_t rank = 0, fc = 0, fo = 0;
#pragma unroll
        for (int q = 1; q <= 15; ++q) {
            const uint64_t m = __ballot(l == q);
            if (l == q) {
                rank = run[q] + (uint32_t)__popcll(m & lt);
                fc = firstv[q];
                fo = offv[q];
            }
            run[q] += (uint32_t)__popcll(m);
        }
        if (l) {
            sorted[fo + rank] = (uint16_t)s;
            if (l <= ROOT) {
                const uint16_t e = (uint16_t)((s << 4) | l);
                for (uint32_t k = rev_bits(fc + rank, l); k < (1u << ROOT); k += 1u << l) tab[k] = e;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// one symbol of a code (uniform).  Returns the symbol, or -1 for bits that are no code word.  `br` holds >= 15 bits.
template <int ROOT>
__device__ __forceinline__ int decode_sym(BitReader& br, const CodeSet& cs, const uint16_t* tab, const uint16_t* sorted) {
    const uint32_t e = uni(tab[br.peek(ROOT)]);
    if (e) {
        br.drop((int)(e & 15u));
        return (int)(e >> 4);
    }
    const uint32_t r = __builtin_bitreverse32((uint32_t)br.bb);    // the stream's next bits, first bit on top
#pragma unroll 1
    for (int l = ROOT + 1; l <= 15; ++l) {
        const uint32_t c = r >> (32 - l);
        const uint32_t idx = c - uni(cs.first[l]);
        if (idx < uni(cs.cnt[l])) {
            br.drop(l);
            return (int)uni(sorted[uni(cs.offs[l]) + idx]);
        }
    }
    // codes of at most ROOT bits that the table does not hold: an incomplete code's unused words
#pragma unroll 1
    for (int l = 1; l <= ROOT; ++l) {
        const uint32_t c = r >> (32 - l);
        const uint32_t idx = c - uni(cs.first[l]);
        if (idx < uni(cs.cnt[l])) {
            br.drop(l);
            return (int)uni(sorted[uni(cs.offs[l]) + idx]);
        }
    }
    return -1;
}

__global__ __launch_bounds__(64) void k_inflate_bgzf(const InfArgs a) {
    __shared__ Lds s;
    const int lane = threadIdx.x;
    for (int blk = blockIdx.x; blk < a.in.n_blocks; blk += gridDim.x) {
        const int64_t in_off = a.in.in_off[blk];
        const int32_t in_len = a.in.in_len[blk];
        const int32_t out_len = a.in.out_len[blk];
        uint8_t* dst = a.out.text + a.in.out_off[blk];
        uint32_t err = 0;
        if (in_len < 0 || out_len < 0 || out_len > 65536 || in_off < 0 || in_off + in_len > a.in.n_comp_bytes) err = TRK_INFLATE_INPUT;
        int pos = 0;          // bytes of text produced
        int flushed = 0;      // bytes of the window already written out
        auto flush_to = [&](int upto) {
            for (int i = flushed + lane; i < upto; i += 64) dst[i] = s.win[i & (INF_WIN - 1)];
            flushed = upto;
        };
        if (!err && out_len > 0) {
            BitReader br;
            const uint8_t* p = a.in.comp + in_off;
            br.init(p, in_len, lane);
            const int64_t bit_limit = (int64_t)in_len * 8;
            bool last = false;
            while (!last && !err) {
                br.refill();
                last = br.take(1) != 0;
                const uint32_t type = br.take(2);
                if (type == 0) {
                    // stored: to the next byte boundary, LEN / NLEN, LEN raw bytes
                    br.drop(br.bc & 7);
                    br.refill();
                    const uint32_t len = br.take(16);
                    br.refill();
                    const uint32_t nlen = br.take(16);
                    if ((len ^ nlen) != 0xffffu) { err = TRK_INFLATE_STREAM; break; }
                    if (pos + (int)len > out_len) { err = TRK_INFLATE_OVERRUN; break; }
                    const int64_t byte0 = br.bits_used() / 8;         // payload offset of the first raw byte
                    if (byte0 + (int64_t)len > (int64_t)in_len) { err = TRK_INFLATE_STREAM; break; }
                    // through the window in pieces (a run may be longer than the window; later matches may reach into it)
                    for (int done = 0; done < (int)len;) {
                        const int n = min((int)len - done, INF_WIN / 2);
                        for (int i = lane; i < n; i += 64) s.win[(pos + done + i) & (INF_WIN - 1)] = p[byte0 + done + i];
                        __builtin_amdgcn_wave_barrier();
                        flush_to(pos + done + n);
                        done += n;
                    }
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
                if (!build_code<LL_ROOT>(s.lens, n_ll, s.ll, s.ll_tab, s.ll_sorted, lane) ||
                    !build_code<D_ROOT>(s.lens + LL_MAX, n_d, s.d, s.d_tab, s.d_sorted, lane)) {
                    err = TRK_INFLATE_STREAM;
                    break;
                }
                // ---- the symbols of this block ----
                for (;;) {
                    br.refill();
                    const int sym = decode_sym<LL_ROOT>(br, s.ll, s.ll_tab, s.ll_sorted);
                    if (sym < 0) { err = TRK_INFLATE_STREAM; break; }
                    if (sym < 256) {
                        if (pos >= out_len) { err = TRK_INFLATE_OVERRUN; break; }
                        if (lane == 0) s.win[pos & (INF_WIN - 1)] = (uint8_t)sym;
                        ++pos;
                    } else if (sym == 256) {
                        break;
                    } else {
                        const int li = sym - 257;
                        if (li > 28) { err = TRK_INFLATE_STREAM; break; }
                        int len;
                        if (li < 8) len = li + 3;
                        else if (li == 28) len = 258;
                        else {
                            const int eb = (li >> 2) - 1;
                            len = ((4 + (li & 3)) << eb) + 3 + (int)br.take(eb);
                        }
                        br.refill();
                        const int ds = decode_sym<D_ROOT>(br, s.d, s.d_tab, s.d_sorted);
                        if (ds < 0 || ds > 29) { err = TRK_INFLATE_STREAM; break; }
                        int dist;
                        if (ds < 4) dist = ds + 1;
                        else {
                            const int eb = (ds >> 1) - 1;
                            dist = ((2 + (ds & 1)) << eb) + 1 + (int)br.take(eb);
                        }
                        if (dist > pos) { err = TRK_INFLATE_STREAM; break; }
                        if (pos + len > out_len) { err = TRK_INFLATE_OVERRUN; break; }
                        // the copy: lanes take bytes; a period shorter than what is left doubles as the copy proceeds
                        int done = 0, d_eff = dist;
                        while (done < len) {
                            const int n = min(min(len - done, 64), d_eff);
                            if (lane < n) s.win[(pos + done + lane) & (INF_WIN - 1)] = s.win[(pos + done + lane - d_eff) & (INF_WIN - 1)];
                            done += n;
                            if (n == d_eff && d_eff < 64) d_eff *= 2;
                        }
                        pos += len;
                    }
                    if (pos - flushed >= 4096) flush_to(pos & ~63);
                    if (br.bits_used() > bit_limit + 64) { err = TRK_INFLATE_STREAM; break; }
                }
            }
            if (!err && br.bits_used() > bit_limit) err = TRK_INFLATE_STREAM;     // read beyond the payload
        }
        if (!err && pos != out_len) err = TRK_INFLATE_OVERRUN;
        if (!err) flush_to(pos);
        if (lane == 0) a.out.flags[blk] = (uint8_t)err;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

namespace trk {
hipError_t launch_inflate(const trk_inflate_in& in, const trk_inflate_out& out, int n_cu, hipStream_t stream) {
    if (in.n_blocks <= 0) return hipSuccess;
    InfArgs a{in, out};
    const int grid = in.n_blocks < n_cu * 4 ? in.n_blocks : n_cu * 4;
    hipLaunchKernelGGL(k_inflate_bgzf, dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}
}  // namespace trk
