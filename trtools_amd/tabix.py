"""Tabix (.tbi) indexes for bgzipped VCFs: reading (region seek for the native reader) and
writing (dumpSTR --zip), without the htslib command line tools.

SURVEY.md section 8(f) row 1 names ``.tbi`` for ``--region`` (statSTR.py:568-570; the reference
goes through cyvcf2 -> htslib) and row 2 ``--zip`` + tabix (dumpSTR.py:1347-1352 shells out to
``tabix``).  Format: SAM/tabix specification ("The Tabix index file format"): a BGZF stream holding
the header (format 2 = VCF, sequence/begin columns 1/2, meta '#'), the sequence names and, per
sequence, the binning index (UCSC scheme, 14-bit minimum shift, 5 levels) with the htslib meta bin
37450 and the 16 kb linear index of smallest virtual offsets.  Virtual offset = compressed block
offset << 16 | offset inside the inflated block.
"""
import gzip
import struct
import zlib

from .bgzf import BgzfWriter

META_BIN = 37450
_SHIFT, _DEPTH = 14, 5


def reg2bin(beg, end):
    """Smallest bin holding [beg, end) (0-based, half open)."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def vcf_interval(fields_prefix):
    """(beg, end) 0-based half open of a VCF record from its first 8 columns, as htslib's tabix
    computes it: POS-1 .. POS-1+len(REF), or INFO/END when present."""
    pos = int(fields_prefix[1])
    beg = pos - 1
    end = beg + len(fields_prefix[3])
    info = fields_prefix[7] if len(fields_prefix) > 7 else ''
    if 'END=' in info:
        for item in info.split(';'):
            if item.startswith('END='):
                try:
                    e = int(item[4:])
                    if e > beg:
                        end = e
                except ValueError:
                    pass
                break
    return beg, max(end, beg + 1)


class TabixIndex:
    """A parsed .tbi: ``names`` and, per sequence, ``bins`` {bin: [(beg, end) virtual offsets]} and
    ``linear`` (list of virtual offsets per 16 kb window)."""

    def __init__(self, names, bins, linear, meta=None):
        self.names, self.bins, self.linear = names, bins, linear
        self.meta = meta or {}

    @classmethod
    def load(cls, path):
        with gzip.open(path, 'rb') as fh:
            raw = fh.read()
        if raw[:4] != b'TBI\x01':
            raise ValueError("%s is not a tabix index" % path)
        n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from('<8i', raw, 4)
        p = 36
        names = [s.decode() for s in raw[p:p + l_nm].split(b'\x00')[:n_ref]]
        p += l_nm
        bins, linear = [], []
        for _ in range(n_ref):
            (n_bin,) = struct.unpack_from('<i', raw, p)
            p += 4
            b = {}
            for _ in range(n_bin):
                bin_id, n_chunk = struct.unpack_from('<Ii', raw, p)
                p += 8
                chunks = [struct.unpack_from('<QQ', raw, p + 16 * k) for k in range(n_chunk)]
                p += 16 * n_chunk
                b[bin_id] = chunks
            (n_intv,) = struct.unpack_from('<i', raw, p)
            p += 4
            linear.append(list(struct.unpack_from('<%dQ' % n_intv, raw, p)))
            p += 8 * n_intv
            bins.append(b)
        return cls(names, bins, linear, dict(format=fmt, col_seq=col_seq, col_beg=col_beg, col_end=col_end,
                                             meta=meta, skip=skip))

    def start_offset(self, chrom, start):
        """Virtual offset from which a scan finds every record of ``chrom`` overlapping positions
        >= start (1-based); None when the sequence is not in the index."""
        if chrom not in self.names:
            return None
        r = self.names.index(chrom)
        lin = self.linear[r]
        w = max(0, (start - 1)) >> _SHIFT if start else 0
        if lin and w < len(lin):
            off = lin[w]
        elif lin:
            return -1   # past the last indexed window: nothing to read
        else:
            off = 0
        if off == 0:
            # windows before the first record: the smallest chunk start of the sequence
            cands = [c[0] for b, chunks in self.bins[r].items() if b != META_BIN for c in chunks]
            return min(cands) if cands else -1
        return off


def _blocks(path):
    """(compressed offset, compressed size, inflated bytes) of every BGZF block of the file."""
    with open(path, 'rb') as fh:
        coff = 0
        while True:
            hdr = fh.read(12)
            if len(hdr) < 12:
                return
            if hdr[:3] != b'\x1f\x8b\x08' or not hdr[3] & 4:
                raise ValueError("%s is not BGZF" % path)
            xlen = struct.unpack_from('<H', hdr, 10)[0]
            extra = fh.read(xlen)
            bsize, q = 0, 0
            while q + 4 <= len(extra):          # the 'BC' subfield need not be the first one
                slen = struct.unpack_from('<H', extra, q + 2)[0]
                if extra[q:q + 2] == b'BC' and slen == 2 and q + 6 <= len(extra):
                    bsize = struct.unpack_from('<H', extra, q + 4)[0] + 1
                    break
                q += 4 + slen
            if len(extra) < xlen or bsize < 12 + xlen + 2 + 8:
                raise ValueError("%s: corrupt BGZF block at offset %d" % (path, coff))
            body = fh.read(bsize - 12 - xlen)
            if len(body) < bsize - 12 - xlen:
                raise ValueError("%s: truncated BGZF block at offset %d" % (path, coff))
            if struct.unpack_from('<I', body, len(body) - 4)[0] > 65536:
                raise ValueError("%s: corrupt BGZF block at offset %d (inflated size)" % (path, coff))
            data = zlib.decompress(body[:-8], -15)
            yield coff, bsize, data
            coff += bsize


def _lines(path):
    """(virtual offset of the line start, virtual offset just past its newline, line bytes) for every
    line of a BGZF text file.  A line that ends with its block is followed by the START of the next
    block, as htslib's bgzf_getline reports it."""
    carry, carry_voff = b'', 0
    for coff, bsize, data in _blocks(path):
        if not data:
            continue
        text = carry + data
        base = len(carry)
        pos = 0
        while True:
            nl = text.find(b'\n', pos)
            if nl < 0:
                break
            start = carry_voff if pos < base else (coff << 16) | (pos - base)
            nxt = nl + 1 - base
            after = (coff << 16) | nxt if nxt < len(data) else (coff + bsize) << 16
            yield start, after, text[pos:nl]
            pos = nl + 1
        if pos < len(text):
            if pos >= base:
                carry_voff = (coff << 16) | (pos - base)
            carry = text[pos:]
        else:
            carry = b''
    if carry:
        yield carry_voff, carry_voff, carry


class TabixBuilder:
    """The index of a position-sorted VCF from its records in file order: ``add`` takes a record's sequence, its 0-based
    half-open interval and the virtual offsets of its line's first byte and of the byte behind its newline; ``finish``
    returns the TabixIndex.  ``build`` feeds it from a file; a writer that knows where its lines lie feeds it as it
    writes (vcfio.VCFWriter: dumpSTR --zip indexes its output without reading it back)."""

    def __init__(self, what='the file'):
        self.what = what
        self.names, self.runs, self.linear, self.stats = [], [], [], []
        self._of = {}
        self._last = None          # (sequence index, begin) of the previous record

    def add(self, chrom, beg, end, start, after):
        r = self._of.get(chrom)
        if r is None:
            r = self._of[chrom] = len(self.names)
            self.names.append(chrom)
            self.runs.append([])       # [bin, first start, last end] of consecutive same-bin records
            self.linear.append([])
            self.stats.append([start, after, 0])
        last = self._last
        if last is not None and (r < last[0] or (r == last[0] and beg < last[1])):
            raise ValueError("%s is not sorted by position (record %s:%d)" % (self.what, chrom, beg + 1))
        self._last = (r, beg)
        b = reg2bin(beg, end)
        runs = self.runs[r]
        if runs and runs[-1][0] == b:
            runs[-1][2] = after
        else:
            runs.append([b, start, after])
        lin = self.linear[r]
        for w in range(beg >> _SHIFT, ((end - 1) >> _SHIFT) + 1):
            while len(lin) <= w:
                lin.append(0)
            if lin[w] == 0:
                lin[w] = start
        self.stats[r][1] = after
        self.stats[r][2] += 1

    def finish(self):
        bins = []
        for r in range(len(self.names)):
            by_bin = {}
            for b, u, v in self.runs[r]:
                chunks = by_bin.setdefault(b, [])
                if chunks and chunks[-1][1] >> 16 >= u >> 16:   # continues in the same block: one chunk
                    chunks[-1][1] = v
                else:
                    chunks.append([u, v])
            by_bin = {b: [tuple(c) for c in chunks] for b, chunks in by_bin.items()}
            by_bin[META_BIN] = [(self.stats[r][0], self.stats[r][1]), (self.stats[r][2], 0)]
            bins.append(by_bin)
            lin = self.linear[r]   # empty windows: leading ones take the first record's offset, the others
            first = next((v for v in lin if v), 0)   # inherit the previous window (as htslib writes them)
            for w in range(len(lin)):
                if lin[w] == 0:
                    lin[w] = lin[w - 1] if w and lin[w - 1] else first
        return TabixIndex(self.names, bins, self.linear,
                          dict(format=2, col_seq=1, col_beg=2, col_end=0, meta=ord('#'), skip=0))


def record_interval(line, limit=None):
    """(sequence name, beg, end) of a VCF record given as bytes (its first eight columns are read, the sample columns
    are not touched)."""
    f = line.split(b'\t', 8) if limit is None else line[:limit].split(b'\t', 8)
    beg, end = vcf_interval([x.decode() for x in f[:8]])
    return f[0].decode(), beg, end


def build(vcf_gz_path, out_path=None):
    """Write ``<vcf>.tbi`` for a position-sorted bgzipped VCF.  Returns the TabixIndex."""
    tb = TabixBuilder(vcf_gz_path)
    for start, after, line in _lines(vcf_gz_path):
        if not line or line[:1] == b'#':
            continue
        chrom, beg, end = record_interval(line)
        tb.add(chrom, beg, end, start, after)
    idx = tb.finish()
    write(idx, out_path or vcf_gz_path + '.tbi')
    return idx


def write(idx, path):
    nm = b''.join(n.encode() + b'\x00' for n in idx.names)
    m = idx.meta
    out = bytearray(b'TBI\x01')
    out += struct.pack('<8i', len(idx.names), m.get('format', 2), m.get('col_seq', 1), m.get('col_beg', 2),
                       m.get('col_end', 0), m.get('meta', ord('#')), m.get('skip', 0), len(nm))
    out += nm
    for r in range(len(idx.names)):
        b = idx.bins[r]
        out += struct.pack('<i', len(b))
        for bin_id in sorted(b):
            chunks = b[bin_id]
            out += struct.pack('<Ii', bin_id, len(chunks))
            for c in chunks:
                out += struct.pack('<QQ', c[0], c[1])
        lin = idx.linear[r]
        out += struct.pack('<i', len(lin))
        out += struct.pack('<%dQ' % len(lin), *lin)
    with BgzfWriter(path) as fh:
        fh.write(bytes(out))
