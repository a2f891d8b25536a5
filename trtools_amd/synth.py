"""Synthetic many-sample TR call sets (bench + tests).

Two halves:

* ``make_loci``  -- seeded numpy generation of the per-locus tables: motif,
  allele sequences / lengths (HipSTR-shape: unit and non-unit alleles, same
  length / different sequence alleles, flank-trim duplicates), allele
  frequencies, missingness, inbreeding.  Small (O(L * A)), always on the host.
* ``cells_numpy`` -- the per-call generator (genotypes + DP/Q/DSTUTTER/
  DFLANKINDEL) as a counter-based hash of (seed, locus, sample).  It is the
  bit-for-bit numpy twin of the HIP kernel ``k_synth`` (csrc/trk_kernels.hip),
  so a full-size batch generated on the device can be spot-checked on the CPU
  row by row without ever materialising it on the host.

``pack_alleles`` turns per-locus allele (length, sequence) lists into the
class tables of ``trk_batch`` (include/trk.h).
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
INT_MISSING = -2147483648


# ---------------------------------------------------------------------------
# allele tables -> trk_batch class arrays
# ---------------------------------------------------------------------------

def pack_alleles(allele_lens, allele_strs):
    """Per-locus lists -> (allele_off, len_class, str_class, len_class_value).

    ``allele_lens[l]``: floats, length in repeat units of every allele index
    (TRRecord.ref_allele_length + alt_allele_lengths, tr_harmonizer.py:740-759).
    ``allele_strs[l]``: the trimmed upper-case sequences (ref_allele + alt_alleles).
    len_class = dense rank of the length (ascending, the key order of
    GetAlleleCounts(uselength=True)); str_class = dense rank of the sequence in
    numpy '<U' / python str order (the key order of GetAlleleCounts(uselength=False)).
    """
    n = len(allele_lens)
    off = np.zeros(n + 1, dtype=np.int32)
    for i in range(n):
        off[i + 1] = off[i] + len(allele_lens[i])
    total = int(off[-1])
    lc = np.zeros(total, dtype=np.uint16)
    sc = np.zeros(total, dtype=np.uint16)
    cv = np.zeros(total, dtype=np.float64)
    for i in range(n):
        lens = [float(x) for x in allele_lens[i]]
        if len(lens) > 65535:
            raise ValueError("more than 65535 alleles at one locus")
        o = int(off[i])
        ul = sorted(set(lens))
        rank = {v: r for r, v in enumerate(ul)}
        for a, v in enumerate(lens):
            lc[o + a] = rank[v]
        cv[o:o + len(ul)] = ul
        strs = list(allele_strs[i]) if allele_strs is not None else None
        if strs is None:
            sc[o:o + len(lens)] = lc[o:o + len(lens)]
        else:
            us = sorted(set(strs))
            srank = {v: r for r, v in enumerate(us)}
            for a, v in enumerate(strs):
                sc[o + a] = srank[v]
    return off, lc, sc, cv


def pack_assoc_tables(allele_lens, precision=2):
    """Per-locus length lists -> (allele_len [sumA] float64, rlen_class [sumA] uint16) of
    ``trk_assoc_params`` (include/trk.h).  allele_len is the LUT of GetLengthGenotypes
    (tr_harmonizer.py:1239) by allele index; rlen_class[allele_off[l] + c] is, for length class c
    (classes as in pack_alleles), the dense rank of ``round(np.float64(length), precision)`` --
    the key clean_len_alleles builds (associaTR/load_and_filter_genotypes.py:37-45; the keys are
    numpy scalars there, so numpy's rounding applies)."""
    total = sum(len(x) for x in allele_lens)
    alen = np.zeros(total, dtype=np.float64)
    rcls = np.zeros(total, dtype=np.uint16)
    o = 0
    for lens in allele_lens:
        lens = [float(x) for x in lens]
        alen[o:o + len(lens)] = lens
        ul = sorted(set(lens))
        rk = [round(np.float64(v), precision) for v in ul]
        ur = sorted(set(rk))
        rank = {v: r for r, v in enumerate(ur)}
        for c, v in enumerate(rk):
            rcls[o + c] = rank[v]
        o += len(lens)
    return alen, rcls


def assoc_tables_from_classes(allele_off, allele_len, len_class_value, n_len_classes, precision=2):
    """``pack_assoc_tables`` for a whole batch from the harmoniser's arrays, no Python loop over loci: allele_len [sumA] (by
    allele index) as it is; rlen_class[allele_off[l] + c] = dense rank of ``round(np.float64(length of class c), precision)``
    among locus l's distinct rounded lengths.  The classes of a locus ascend by length and rounding is monotone, so the
    rounded values ascend too: the rank of class c is the number of changes of value before it."""
    off = np.asarray(allele_off, dtype=np.int64)
    sa = int(off[-1]) if len(off) else 0
    rcls = np.zeros(sa, dtype=np.uint16)
    if sa:
        n = np.diff(off)
        loc = np.repeat(np.arange(len(n)), n)
        c = np.arange(sa) - off[loc]
        valid = c < np.asarray(n_len_classes, dtype=np.int64)[loc]
        rv = np.round(np.asarray(len_class_value, dtype=np.float64), precision)   # (== round(np.float64(v), precision))
        new = np.ones(sa, dtype=bool)
        new[1:] = (rv[1:] != rv[:-1]) | (loc[1:] != loc[:-1])
        new &= valid
        run = np.cumsum(new)
        rank = run - run[off[loc]]          # the locus's first class is new: run[first] counts it
        rcls[valid] = rank[valid].astype(np.uint16)
    return np.ascontiguousarray(allele_len, dtype=np.float64).copy(), rcls


def pack_dosage_tables(allele_lens, precision=2):
    """Per-locus length lists -> the per-allele tables of ``trk_assoc_dosage`` (include/trk.h):
    perm int32 (allele indices ordered by (class, index)), dclass uint16 (by allele index: rank of
    ``round(length, precision)`` -- python floats, python round, as ``len_alleles`` in
    load_and_filter_genotypes.py:171-172 -- among the locus's distinct rounded lengths),
    dclass_value float64 (by class) and best_class uint16 (by allele index: the class whose value
    equals ``np.around(length, precision)``, the best-guess call of :199-202; 0xffff when none)."""
    total = sum(len(x) for x in allele_lens)
    perm = np.zeros(total, dtype=np.int32)
    dcls = np.zeros(total, dtype=np.uint16)
    dval = np.zeros(total, dtype=np.float64)
    best = np.full(total, 0xffff, dtype=np.uint16)
    o = 0
    for lens in allele_lens:
        lens = [float(x) for x in lens]
        rl = [round(x, precision) for x in lens]
        uniq = sorted(set(rl))
        rank = {v: r for r, v in enumerate(uniq)}
        n = len(lens)
        for a in range(n):
            dcls[o + a] = rank[rl[a]]
            around = float(np.around(lens[a], precision))
            best[o + a] = rank.get(around, 0xffff)
        dval[o:o + len(uniq)] = uniq
        perm[o:o + n] = sorted(range(n), key=lambda a: (rank[rl[a]], a))
        o += n
    return perm, dcls, dval, best


# ---------------------------------------------------------------------------
# per-locus tables
# ---------------------------------------------------------------------------

class Loci:
    """Per-locus synthetic tables (host)."""

    def __init__(self):
        self.motifs = []
        self.allele_strs = []
        self.allele_lens = []
        self.allele_off = None
        self.cdf24 = None
        self.miss_thr16 = None
        self.inbreed_thr16 = None

    def slice(self, lo, hi):
        """Tables of loci [lo, hi): one rank's contiguous shard of a cohort-wide table."""
        out = Loci()
        out.motifs = self.motifs[lo:hi]
        out.allele_strs = self.allele_strs[lo:hi]
        out.allele_lens = self.allele_lens[lo:hi]
        a0, a1 = int(self.allele_off[lo]), int(self.allele_off[hi])
        out.allele_off = (self.allele_off[lo:hi + 1] - a0).astype(np.int32)
        out.cdf24 = self.cdf24[a0:a1]
        out.miss_thr16 = self.miss_thr16[lo:hi]
        out.inbreed_thr16 = self.inbreed_thr16[lo:hi]
        return out


def make_loci(n_loci, n_samples, seed, max_alleles=None, all_missing_frac=0.01, inbred_frac=0.10,
              miss_rate=0.03, pure_repeats=False):
    """HipSTR-shape loci. ``pure_repeats`` -> GangSTR-shape."""
    rng = np.random.default_rng(seed)
    if max_alleles is None:
        max_alleles = int(min(64, max(4, 4 * np.log2(max(n_samples, 2)))))
    lam = max(1.0, 0.9 * np.log2(max(n_samples, 2)))
    out = Loci()
    off = [0]
    cdfs = []
    period_w = np.array([0.30, 0.35, 0.12, 0.13, 0.06, 0.04])
    bases = np.array(list('ACGT'))
    for _ in range(n_loci):
        period = int(rng.choice(6, p=period_w)) + 1
        motif = ''.join(rng.choice(bases, size=period))
        if period > 1 and len(set(motif)) == 1:
            motif = motif[:-1] + ('C' if motif[0] != 'C' else 'G')
        ref_copies = int(rng.integers(8, 31))
        ref = motif * ref_copies
        n_alt = int(min(max_alleles - 1, 1 + rng.poisson(lam)))
        strs = [ref]
        seen = {ref}
        tries = 0
        while len(strs) < 1 + n_alt and tries < 8 * n_alt + 16:
            tries += 1
            k = int(rng.integers(-6, 9))
            copies = max(1, ref_copies + k)
            s = motif * copies
            u = rng.random()
            if not pure_repeats:
                if u < 0.10 and period > 1:      # non-unit (fractional) allele
                    s = s + motif[: int(rng.integers(1, period))]
                elif u < 0.15:                   # same length, different sequence
                    p = int(rng.integers(0, len(s)))
                    c = 'A' if s[p] != 'A' else 'T'
                    s = s[:p] + c + s[p + 1:]
            if u >= 0.985 and not pure_repeats:
                s = strs[int(rng.integers(0, len(strs)))]   # duplicate after flank trimming
            elif s in seen:
                continue
            seen.add(s)
            strs.append(s)
        A = len(strs)
        w = rng.dirichlet(np.full(A, 0.5))
        w[0] += 0.5
        w /= w.sum()
        c = np.floor(np.cumsum(w) * (1 << 24)).astype(np.int64)
        c = np.minimum(c, (1 << 24))
        c[-1] = 1 << 24
        cdfs.append(c.astype(np.uint32))
        out.motifs.append(motif)
        out.allele_strs.append(strs)
        out.allele_lens.append([len(s) / len(motif) for s in strs])
        off.append(off[-1] + A)
    out.allele_off = np.array(off, dtype=np.int32)
    out.cdf24 = np.concatenate(cdfs) if cdfs else np.zeros(0, dtype=np.uint32)
    miss = np.full(n_loci, int(round(miss_rate * 65536)), dtype=np.uint32)
    miss[rng.random(n_loci) < all_missing_frac] = 65536
    out.miss_thr16 = miss
    inb = np.zeros(n_loci, dtype=np.uint32)
    inb[rng.random(n_loci) < inbred_frac] = int(round(0.3 * 65536))
    out.inbreed_thr16 = inb
    return out


# ---------------------------------------------------------------------------
# per-call generator (numpy twin of k_synth)
# ---------------------------------------------------------------------------

def _mix64(z):
    z = z.copy()
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def _popcount(x):
    x = x.astype(np.uint64)
    c = np.zeros(x.shape, dtype=np.int32)
    for b in range(8):
        c += ((x >> np.uint64(b)) & np.uint64(1)).astype(np.int32)
    return c


def cells_numpy(seed, loci, locus_idx, n_samples, locus_base=0):
    """Generate the calls of loci ``locus_idx`` (indices into ``loci``) for all samples.

    Returns dict(gt int16 [n,S,2], dp int32 [n,S], q float32 [n,S],
    dstutter int32, dflankindel int32).  ``locus_base + locus_idx`` is the
    global locus number fed to the hash (so shards of a larger call set agree).
    """
    locus_idx = np.asarray(locus_idx, dtype=np.int64)
    n = locus_idx.shape[0]
    S = int(n_samples)
    old = np.seterr(over='ignore')
    try:
        gl = (locus_idx + int(locus_base)).astype(np.uint64)[:, None]
        s = np.arange(S, dtype=np.uint64)[None, :]
        gidx = gl * np.uint64(S) + s
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (gidx + np.uint64(1))
        h1 = _mix64(x)
        h2 = _mix64(x + np.uint64(0x632BE59BD9B4E019))
        h3 = _mix64(x + np.uint64(0xD1B54A32D192ED03))
    finally:
        np.seterr(**old)
    u0 = (h1 & np.uint64(0xffffff)).astype(np.uint32)
    u1 = ((h1 >> np.uint64(24)) & np.uint64(0xffffff)).astype(np.uint32)
    um = ((h1 >> np.uint64(48)) & np.uint64(0xffff)).astype(np.uint32)
    ui = (h2 & np.uint64(0xffff)).astype(np.uint32)
    a0 = np.zeros((n, S), dtype=np.int32)
    a1 = np.zeros((n, S), dtype=np.int32)
    for r, li in enumerate(locus_idx):
        o, e = int(loci.allele_off[li]), int(loci.allele_off[li + 1])
        cdf = loci.cdf24[o:e]
        A = e - o
        a0[r] = np.minimum(np.searchsorted(cdf, u0[r], side='right'), A - 1)
        a1[r] = np.minimum(np.searchsorted(cdf, u1[r], side='right'), A - 1)
    inb = loci.inbreed_thr16[locus_idx][:, None]
    a1 = np.where(ui < inb, a0, a1)
    mt = loci.miss_thr16[locus_idx][:, None]
    nocall = um < mt
    partial = (~nocall) & (um < mt + np.uint32(328))
    g0 = np.where(nocall, -1, a0).astype(np.int16)
    g1 = np.where(nocall | partial, -1, a1).astype(np.int16)
    gt = np.stack([g0, g1], axis=2)
    bsum = (((h2 >> np.uint64(16)) & np.uint64(0xff)) + ((h2 >> np.uint64(24)) & np.uint64(0xff)) +
            ((h2 >> np.uint64(32)) & np.uint64(0xff)) + ((h2 >> np.uint64(40)) & np.uint64(0xff))).astype(np.int64)
    v = bsum * 12
    d = np.where(v >= 1680, (v - 1680) // 148, 0).astype(np.int32)
    r32 = (h3 & np.uint64(0xffffffff)).astype(np.uint64)
    # clz32 via frexp (exact for < 2**53)
    _, ex = np.frexp(r32.astype(np.float64))
    lz = np.where(r32 == 0, 32, 32 - ex).astype(np.int32)
    qi = 100 - (2 * lz + ((h3 >> np.uint64(32)) & np.uint64(1)).astype(np.int32))
    qi = np.maximum(qi, 0)
    st = _popcount((h3 >> np.uint64(33)) & np.uint64(0xff)) >> 1
    fl = _popcount((h3 >> np.uint64(41)) & np.uint64(0xf)) >> 1
    st = np.minimum(st, d)
    fl = np.minimum(fl, d)
    q = (qi.astype(np.float32) / np.float32(100.0)).astype(np.float32)
    dp = np.where(nocall, INT_MISSING, d).astype(np.int32)
    q = np.where(nocall, np.float32(np.nan), q).astype(np.float32)
    st = np.where(nocall, INT_MISSING, st).astype(np.int32)
    fl = np.where(nocall, INT_MISSING, fl).astype(np.int32)
    return dict(gt=gt, dp=dp, q=q, dstutter=st, dflankindel=fl)


def allele_repcn(loci):
    """Integer repeat count of every allele (GangSTR REPCN), concatenated in allele_off order."""
    out = []
    for lens in loci.allele_lens:
        out.extend(int(round(x)) for x in lens)
    return np.array(out, dtype=np.int32)


def gangstr_planes_numpy(seed, loci, locus_idx, n_samples, gt, dp, locus_base=0):
    """numpy twin of k_synth_gangstr: QEXP [n,S,3] f32, REPCN [n,S,2], RC [n,S,4], REPCI [n,S,4] int32."""
    locus_idx = np.asarray(locus_idx, dtype=np.int64)
    n, S = locus_idx.shape[0], int(n_samples)
    old = np.seterr(over='ignore')
    try:
        gl = (locus_idx + int(locus_base)).astype(np.uint64)[:, None]
        s = np.arange(S, dtype=np.uint64)[None, :]
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (gl * np.uint64(S) + s + np.uint64(1))
        h4 = _mix64(x + np.uint64(0xA0761D6478BD642F))
        h5 = _mix64(x + np.uint64(0xE7037ED1A0B428DB))
    finally:
        np.seterr(**old)
    g = gt.astype(np.int64)
    nocall = (g[:, :, 0] < 0) & (g[:, :, 1] < 0)
    d = np.where(dp < 0, 0, dp).astype(np.int64)
    rep_all = allele_repcn(loci)
    b0 = (h4 & np.uint64(0xff)).astype(np.int64)
    b1 = ((h4 >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
    p0 = b0 * 1000 // 255
    p1 = ((1000 - p0) * b1) >> 8
    p2 = 1000 - p0 - p1
    qexp = np.stack([p0, p1, p2], axis=2).astype(np.float32) / np.float32(1000.0)
    sent = ((h4 >> np.uint64(16)) & np.uint64(0xf)) == 0
    qexp[sent] = np.float32(-1.0)
    qexp[nocall] = np.float32(np.nan)
    e = ((h4 >> np.uint64(20)) & np.uint64(0xffff)).astype(np.int64)
    repcn = np.full((n, S, 2), INT_MISSING, dtype=np.int64)
    repci = np.full((n, S, 4), INT_MISSING, dtype=np.int64)
    offs = loci.allele_off[locus_idx].astype(np.int64)[:, None]
    for j in range(2):
        gj = g[:, :, j]
        ok = gj >= 0
        r = np.where(ok, rep_all[np.clip(offs + gj, 0, len(rep_all) - 1)], INT_MISSING)
        repcn[:, :, j] = r
        lo = np.maximum(r - ((e >> (4 * j)) & 3), 0)
        hi = r + ((e >> (4 * j + 2)) & 3)
        bad = ((e >> (8 + j)) & 0x1f) == 0
        lo = np.where(bad, r + 1, lo)
        hi = np.where(bad, r + 2, hi)
        repci[:, :, 2 * j] = np.where(ok, lo, INT_MISSING)
        repci[:, :, 2 * j + 1] = np.where(ok, hi, INT_MISSING)
    c0 = (h5 & np.uint64(0xff)).astype(np.int64)
    c1 = ((h5 >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
    c2 = ((h5 >> np.uint64(16)) & np.uint64(0xff)).astype(np.int64)
    encl = (c0 * (d + 1)) >> 8
    rem = d - encl
    span = (c1 * (rem + 1)) >> 8
    rem = rem - span
    frr = (c2 * (rem + 1)) >> 8
    bound = rem - frr
    mode = ((h5 >> np.uint64(24)) & np.uint64(0x3f)).astype(np.int64)
    m0, m1 = mode == 0, mode == 1
    encl = np.where(m0 | m1, 0, encl)
    frr = np.where(m0 | m1, 0, frr)
    span = np.where(m0, d, np.where(m1, d // 2, span))
    bound = np.where(m0, 0, np.where(m1, d - d // 2, bound))
    rc = np.stack([encl, span, frr, bound], axis=2)
    rc[nocall] = INT_MISSING
    repcn[nocall] = INT_MISSING
    repci[nocall] = INT_MISSING
    return dict(qexp=qexp, repcn=repcn.astype(np.int32), rc=rc.astype(np.int32), repci=repci.astype(np.int32))


class SynthBatch:
    """A synthetic call set resident on the device (see Engine.synth_fill)."""

    def __init__(self, eng, n_loci, n_samples, seed, planes=('dp', 'q'), locus_base=0, loci=None,
                 pure_repeats=False):
        self.eng = eng
        self.n_loci, self.n_samples, self.seed = n_loci, n_samples, seed
        self.n_pad, self.n_dev = 0, n_samples
        self.locus_base = locus_base
        self.loci = loci if loci is not None else make_loci(n_loci, n_samples, seed,
                                                            pure_repeats=pure_repeats)
        lo = self.loci
        off, lc, sc, cv = pack_alleles(lo.allele_lens, lo.allele_strs)
        self.tables = (off, lc, sc, cv)
        self.d_off = eng.upload(off, np.int32)
        d_cdf = eng.upload(lo.cdf24, np.uint32)
        d_miss = eng.upload(lo.miss_thr16, np.uint32)
        d_inb = eng.upload(lo.inbreed_thr16, np.uint32)
        self.dev = eng.synth_fill(seed, n_loci, n_samples, self.d_off, d_cdf, d_miss, d_inb,
                                  locus_base=locus_base, planes=planes)
        eng.sync()
        for t in (d_cdf, d_miss, d_inb):
            t.free()
        self.batch = eng.make_batch(self.dev['gt'], self.d_off, lc, sc, cv,
                                    max_alleles=int(np.max(np.diff(off))) if n_loci else 0)

    def pad_rows(self, align=32):
        """Append padding samples (no-call genotypes, missing FORMAT values; trk_batch.n_pad_samples) to every plane
        generated so far so that the rows are a multiple of ``align`` samples -- what compute.DeviceCompute does with
        a cohort it uploads (trk_pad_rows: every row starts on a 128-byte boundary).  n_samples stays the number of
        real samples; n_dev is the row length on the device."""
        n_pad = (-self.n_samples) % align if align else 0
        if n_pad and not self.n_pad:
            for k in list(self.dev):
                self.dev[k] = self.eng.pad_samples(self.dev[k], n_pad)
            self.n_pad = n_pad
            self.n_dev = self.n_samples + n_pad
            off, lc, sc, cv = self.tables
            old = self.batch
            self.batch = self.eng.make_batch(self.dev['gt'], self.d_off, lc, sc, cv, n_pad=n_pad,
                                             max_alleles=int(np.max(np.diff(off))) if self.n_loci else 0)
            for k, a in old.arrays.items():
                if k not in ('gt', 'allele_off'):
                    a.free()
        return self.n_pad

    def host_rows(self, locus_idx):
        """CPU regeneration of selected loci (for spot-check parity at full size)."""
        rows = cells_numpy(self.seed, self.loci, locus_idx, self.n_samples, self.locus_base)
        return self._pad_host(rows)

    def _pad_host(self, rows):
        """The padding samples of pad_rows on regenerated host rows (no-call genotypes, missing values)."""
        if not self.n_pad:
            return rows
        out = {}
        for k, a in rows.items():
            fill = -1 if a.dtype == np.int16 else (np.nan if a.dtype.kind == 'f' else INT_MISSING)
            pad = np.full((a.shape[0], self.n_pad) + a.shape[2:], fill, dtype=a.dtype)
            out[k] = np.concatenate([a, pad], axis=1)
        return out

    def add_gangstr_planes(self):
        """Generate QEXP / REPCN / RC / REPCI on the device for this call set (GangSTR shape)."""
        d_rep = self.eng.upload(allele_repcn(self.loci), np.int32)
        out = self.eng.synth_fill_gangstr(self.seed, self.n_loci, self.n_samples, self.d_off, self.dev['gt'],
                                          self.dev['dp'], d_rep, locus_base=self.locus_base)
        self.eng.sync()
        d_rep.free()
        self.dev.update(out)
        return out

    def host_gangstr_rows(self, locus_idx, base_rows=None):
        h = base_rows if base_rows is not None else self.host_rows(locus_idx)
        S = self.n_samples
        return self._pad_host(gangstr_planes_numpy(self.seed, self.loci, locus_idx, S, h['gt'][:, :S], h['dp'][:, :S],
                                                   self.locus_base))


# ---------------------------------------------------------------------------
# text rendering (small scale: parse -> pack plumbing and CLI parity tests)
# ---------------------------------------------------------------------------

def _fmt_f32(x):
    return '.' if np.isnan(x) else '%g' % float(x)


def _fmt_i32(x):
    return '.' if int(x) == INT_MISSING else str(int(x))


def render_vcf(path, loci, rows, caller='hipstr', n_samples=None, extra=None, chrom='chr1'):
    """Write the call set ``rows`` (output of cells_numpy for all loci of ``loci``) as VCF text.

    caller='hipstr': flanked alleles (one base on each side, removed again by the START/END
    trimming of the harmoniser), FORMAT GT:GB:Q:DP:DSTUTTER:DFLANKINDEL:ALLREADS.
    caller='gangstr': FORMAT GT:DP:Q:REPCN:REPCI:RC:QEXP, ``extra`` = gangstr_planes_numpy output."""
    gt = rows['gt']
    n, S = gt.shape[0], gt.shape[1]
    names = ['S%04d' % i for i in range(S)]
    with open(path, 'w') as fh:
        fh.write('##fileformat=VCFv4.1\n')
        if caller == 'hipstr':
            fh.write('##command=HipSTR-v0.6.2 --synthetic\n')
            for k, t, d in (('START', 'Integer', 'start'), ('END', 'Integer', 'end'), ('PERIOD', 'Integer', 'period')):
                fh.write('##INFO=<ID=%s,Number=1,Type=%s,Description="%s">\n' % (k, t, d))
            fmts = (('GT', 'String'), ('GB', 'String'), ('Q', 'Float'), ('DP', 'Integer'), ('DSTUTTER', 'Integer'),
                    ('DFLANKINDEL', 'Integer'), ('ALLREADS', 'String'))
        else:
            fh.write('##command=GangSTR-2.4 --synthetic\n')
            fh.write('##INFO=<ID=RU,Number=1,Type=String,Description="motif">\n')
            fmts = (('GT', 'String'), ('DP', 'Integer'), ('Q', 'Float'), ('REPCN', 'Integer', '2'), ('REPCI', 'String'),
                    ('RC', 'String'), ('QEXP', 'Float', '3'))
        for f in fmts:
            num = f[2] if len(f) > 2 else '1'
            fh.write('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % (f[0], num, f[1], f[0]))
        fh.write('##contig=<ID=%s>\n' % chrom)
        fh.write('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join(names) + '\n')
        for l in range(n):
            strs = loci.allele_strs[l]
            pos = 1000 + 500 * l
            if caller == 'hipstr':
                ref = 'G' + strs[0] + 'T'
                alts = ['G' + a + 'T' for a in strs[1:]]
                info = 'START=%d;END=%d;PERIOD=%d' % (pos + 1, pos + len(strs[0]), len(loci.motifs[l]))
                fkeys = 'GT:GB:Q:DP:DSTUTTER:DFLANKINDEL:ALLREADS'
            else:
                ref, alts = strs[0], list(strs[1:])
                info = 'RU=%s' % loci.motifs[l].lower()
                fkeys = 'GT:DP:Q:REPCN:REPCI:RC:QEXP'
            cols = [chrom, str(pos), 'STR_%d' % l, ref, ','.join(alts) if alts else '.', '.', '.', info, fkeys]
            for s in range(S):
                a0, a1 = int(gt[l, s, 0]), int(gt[l, s, 1])
                g = '%s|%s' % ('.' if a0 < 0 else a0, '.' if a1 < 0 else a1)
                if a0 < 0 and a1 < 0:
                    cols.append(':'.join([g] + ['.'] * (len(fkeys.split(':')) - 1)))
                    continue
                dp, q = rows['dp'][l, s], rows['q'][l, s]
                if caller == 'hipstr':
                    diffs = [(len(strs[a]) - len(strs[0])) if a >= 0 else 0 for a in (a0, a1)]
                    gb = '%d|%d' % tuple(diffs)
                    half = int(dp) // 2
                    reads = {}
                    reads[diffs[0]] = reads.get(diffs[0], 0) + half
                    reads[diffs[1]] = reads.get(diffs[1], 0) + (int(dp) - half) // (1 + (s % 3 == 0))
                    ar = ';'.join('%d|%d' % kv for kv in sorted(reads.items()) if kv[1] > 0) or '.'
                    cols.append(':'.join([g, gb, _fmt_f32(q), _fmt_i32(dp), _fmt_i32(rows['dstutter'][l, s]),
                                          _fmt_i32(rows['dflankindel'][l, s]), ar]))
                else:
                    e = extra
                    rep = ','.join(_fmt_i32(x) for x in e['repcn'][l, s])
                    ci = e['repci'][l, s]
                    rci = ','.join('%s-%s' % (_fmt_i32(ci[2 * j]), _fmt_i32(ci[2 * j + 1])) for j in range(2))
                    rc = ','.join(_fmt_i32(x) for x in e['rc'][l, s])
                    qx = ','.join(_fmt_f32(x) for x in e['qexp'][l, s])
                    cols.append(':'.join([g, _fmt_i32(dp), _fmt_f32(q), rep, rci, rc, qx]))
            fh.write('\t'.join(cols) + '\n')
    return names
