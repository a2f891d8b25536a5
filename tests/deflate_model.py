"""A line-by-line model of k_deflate_bgzf (trtools_amd/csrc/trk_deflate.hip): the DEFLATE stream the device makes of one
BGZF member's text -- greedy LZ77 over a table of eight candidates per hash, one dynamic-Huffman block per member.  TEST
INFRASTRUCTURE: the checker of the device's bytes is zlib's inflate (any valid stream that gives the text back is right);
this model additionally says which valid stream the kernel is meant to produce, so that a difference points at the step
that went wrong.  Nothing here is imported by the product."""

HASH_BITS = 8             # buckets of the candidate table ...
WAYS = 8                  # ... of eight places each: position q goes to place q % 8 of its bucket
MEMBER = 16384            # bytes of text per member (include/trk.h: TRK_DEFLATE_MEMBER)
MIN_MATCH, MAX_MATCH, MAX_DIST = 4, 258, 32768
FIRST = 16                # bytes of every candidate compared side by side; only the winner is followed beyond them
INSERT = 64               # positions of a match that enter the table (its first ones)
LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
         8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CLORD = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def len_sym(l):
    for i in range(28, -1, -1):
        if l >= LBASE[i]:
            return i


def dist_sym(d):
    for i in range(29, -1, -1):
        if d >= DBASE[i]:
            return i


def hash4(text, i):
    v = text[i] | (text[i + 1] << 8) | (text[i + 2] << 16) | (text[i + 3] << 24)
    return ((v * 2654435761) & 0xffffffff) >> (32 - HASH_BITS)


def _common(text, c, p, start, limit):
    l = start
    while l < limit and text[c + l] == text[p + l]:
        l += 1
    return l


def _find(text, table, p):
    """(length over the first FIRST bytes, distance) of the best candidate of p's bucket: the longest, the nearest among
    equals; (0, 0) when that is not a match."""
    n, best, dist = len(text), 0, 0
    if p + 4 <= n:
        h = hash4(text, p)
        lim = min(FIRST, MAX_MATCH, n - p)
        for w in range(WAYS):
            c = table[h * WAYS + w]            # position + 1, 0: none
            if c and p + 1 - c <= MAX_DIST:
                c -= 1
                l = _common(text, c, p, 0, lim)
                if l > best or (l == best and l and p - c < dist):
                    best, dist = l, p - c
    return (best, dist) if best >= MIN_MATCH else (0, 0)


def lz_tokens(text):
    """[(byte, 0) | (length, distance)].  The table holds, per hash of four bytes, the last position of each residue
    modulo WAYS that was entered.  A step looks at TWO positions, p and p + 1, against the table as it is (the kernel: sixteen
    candidates side by side): the better match over the first FIRST bytes wins -- p on a tie -- and only the winner is
    followed beyond FIRST bytes; when it is p + 1's, the byte at p goes out as a literal first; when neither has one, two
    literals go out.  A token's positions -- the first INSERT of a match -- enter the table, later ones over earlier ones."""
    n, table, toks, p = len(text), [0] * (WAYS << HASH_BITS), [], 0

    def enter(q):
        if q + 4 <= n:
            table[hash4(text, q) * WAYS + (q & (WAYS - 1))] = q + 1

    def literal(q):
        toks.append((text[q], 0))
        enter(q)

    while p < n:
        b0, d0 = _find(text, table, p)
        b1, d1 = _find(text, table, p + 1)
        if b1 > b0:
            literal(p)
            p, b0, d0 = p + 1, b1, d1
        if b0:
            limit = min(MAX_MATCH, n - p)
            if b0 == FIRST and limit > FIRST:
                b0 = _common(text, p - d0, p, FIRST, limit)
            for q in range(p, p + min(b0, INSERT)):
                enter(q)
            toks.append((b0, d0))
            p += b0
        else:
            literal(p)
            p += 1
            if p < n:
                literal(p)             # (its bucket had no match for it either)
                p += 1
    return toks


def huffman_lengths(freq, limit):
    """Code lengths by the two-queue construction: leaves in (frequency, symbol) order, a leaf before an internal node of
    the same weight; frequencies halved (never to zero) and the tree rebuilt while a code is longer than ``limit``."""
    n = len(freq)
    f = list(freq)
    used = [s for s in range(n) if f[s]]
    lengths = [0] * n
    if not used:
        return lengths
    if len(used) == 1:
        lengths[used[0]] = 1
        return lengths
    while True:
        leaves = sorted(used, key=lambda s: (f[s], s))
        m = len(leaves)
        weight = [f[s] for s in leaves] + [0] * (m - 1)       # nodes 0 .. m-1 leaves (sorted), m .. 2m-2 internal
        parent = [0] * (2 * m - 1)
        li, ii, made = 0, m, m
        for _ in range(m - 1):
            pick = []
            for _k in range(2):
                if li < m and (ii >= made or weight[li] <= weight[ii]):
                    pick.append(li)
                    li += 1
                else:
                    pick.append(ii)
                    ii += 1
            weight[made] = weight[pick[0]] + weight[pick[1]]
            parent[pick[0]] = parent[pick[1]] = made
            made += 1
        depth = [0] * (2 * m - 1)
        for node in range(2 * m - 3, -1, -1):
            depth[node] = depth[parent[node]] + 1
        if max(depth[:m]) <= limit:
            for k, s in enumerate(leaves):
                lengths[s] = depth[k]
            return lengths
        f = [(x + 1) >> 1 if x else 0 for x in f]


def canonical_codes(lengths):
    """DEFLATE's canonical codes, bit-reversed (the stream is filled from bit 0)."""
    maxl = max(lengths) if lengths else 0
    count = [0] * (maxl + 2)
    for l in lengths:
        if l:
            count[l] += 1
    nxt, code = [0] * (maxl + 2), 0
    for b in range(1, maxl + 1):
        code = (code + count[b - 1]) << 1
        nxt[b] = code
    out = [0] * len(lengths)
    for s, l in enumerate(lengths):
        if l:
            c = nxt[l]
            nxt[l] += 1
            r = 0
            for k in range(l):
                r |= ((c >> k) & 1) << (l - 1 - k)
            out[s] = r
    return out


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, b):
        self.acc |= v << self.n
        self.n += b
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def code_length_runs(seq):
    """The run-length form of the code-length sequence: (symbol 0 ... 18, extra value, extra bits)."""
    out, i = [], 0
    while i < len(seq):
        v, j = seq[i], i
        while j < len(seq) and seq[j] == v:
            j += 1
        run = j - i
        if v == 0:
            while run >= 11:
                r = min(run, 138)
                out.append((18, r - 11, 7))
                run -= r
            if run >= 3:
                out.append((17, run - 3, 3))
                run = 0
            while run > 0:
                out.append((0, 0, 0))
                run -= 1
        else:
            out.append((v, 0, 0))
            run -= 1
            while run >= 3:
                r = min(run, 6)
                out.append((16, r - 3, 2))
                run -= r
            while run > 0:
                out.append((v, 0, 0))
                run -= 1
        i = j
    return out


def encode_tokens(toks):
    lf, df = [0] * 286, [0] * 30
    for a, b in toks:
        if b:
            lf[257 + len_sym(a)] += 1
            df[dist_sym(b)] += 1
        else:
            lf[a] += 1
    lf[256] = 1
    ll, dl = huffman_lengths(lf, 15), huffman_lengths(df, 15)
    if not any(dl):
        dl[0] = 1                      # (a block without matches still declares one distance code)
    hlit = 286
    while hlit > 257 and ll[hlit - 1] == 0:
        hlit -= 1
    hdist = 30
    while hdist > 1 and dl[hdist - 1] == 0:
        hdist -= 1
    runs = code_length_runs(ll[:hlit] + dl[:hdist])
    cf = [0] * 19
    for s, _, _ in runs:
        cf[s] += 1
    cl = huffman_lengths(cf, 7)
    hclen = 19
    while hclen > 4 and cl[CLORD[hclen - 1]] == 0:
        hclen -= 1
    bw = _Bits()
    bw.put(1, 1)                       # BFINAL
    bw.put(2, 2)                       # dynamic Huffman
    bw.put(hlit - 257, 5)
    bw.put(hdist - 1, 5)
    bw.put(hclen - 4, 4)
    for k in range(hclen):
        bw.put(cl[CLORD[k]], 3)
    cc = canonical_codes(cl)
    for s, e, eb in runs:
        bw.put(cc[s], cl[s])
        if eb:
            bw.put(e, eb)
    lc, dc = canonical_codes(ll), canonical_codes(dl)
    for a, b in toks:
        if b:
            ls = len_sym(a)
            bw.put(lc[257 + ls], ll[257 + ls])
            if LEXT[ls]:
                bw.put(a - LBASE[ls], LEXT[ls])
            ds = dist_sym(b)
            bw.put(dc[ds], dl[ds])
            if DEXT[ds]:
                bw.put(b - DBASE[ds], DEXT[ds])
        else:
            bw.put(lc[a], ll[a])
    bw.put(lc[256], ll[256])
    return bw.done()


def deflate_member(text):
    """The raw DEFLATE stream of one member's text (at most MEMBER bytes)."""
    n = len(text)
    assert 0 < n <= MEMBER
    dyn = encode_tokens(lz_tokens(text))
    if len(dyn) <= n + 5:
        return dyn
    # a member that would not get smaller is stored: BFINAL = 1, type 0, LEN, ~LEN, the text
    return bytes([1, n & 0xff, n >> 8, ~n & 0xff, (~n >> 8) & 0xff]) + bytes(text)
