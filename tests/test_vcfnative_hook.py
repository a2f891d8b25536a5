"""The native reader's inflate hook (include/trk_vcf.h: trk_vcf_set_inflate_hook) with a hook written in Python: the
members are inflated by zlib HERE, the hook hands back the newlines and copies only the HEADS of the lines into the
reader's text buffer -- every other byte of it is poisoned -- exactly what the device hook of trk_api.hip does.  The
batches the reader then returns (lines, field offsets, FORMAT key indices, harmonised alleles) must equal the ones of
a plain read of the same file, whatever the fill boundaries cut through (heads, CRLF pairs, the last line without a
newline).  CPU only: this pins the reader's half of the device-inflate path."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from helpers import GOLDEN

SEED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class IBlock(C.Structure):
    _fields_ = [('payload_off', C.c_uint64), ('payload_len', C.c_uint32), ('isize', C.c_uint32), ('dst', C.c_uint64)]


INFLATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(IBlock), C.c_int, C.c_uint64, C.c_size_t,
                         C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t))


SUBMIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(IBlock), C.c_int, C.c_uint64, C.c_size_t)
COLLECT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t))


class PyHook:
    """zlib inflate + line index + heads, with the carry of the unfinished line between calls."""

    def __init__(self):
        self.text = bytearray()          # the whole stream (what a device keeps in its segments)
        self.keep = []
        self.calls = 0
        self.max_seen = 0
        self.queue, self.max_in_flight = [], 0

        def seed(user, text, n):
            self.text += C.string_at(text, n)
            return 0

        def inflate(user, comp, comp_bytes, blocks, n_blocks, abs_base, total, out, line_state, nl, n_nl):
            raw = C.string_at(comp, comp_bytes)
            blks = [(blocks[i].payload_off, blocks[i].payload_len, blocks[i].isize, blocks[i].dst) for i in range(n_blocks)]
            return work(raw, blks, abs_base, total, out, line_state, nl, n_nl)

        # the two halves (trk_vcf_inflate_hook.submit / collect): the reader hands run k + 1 over before it collects run k
        def submit(user, comp, comp_bytes, blocks, n_blocks, abs_base, total):
            blks = [(blocks[i].payload_off, blocks[i].payload_len, blocks[i].isize, blocks[i].dst) for i in range(n_blocks)]
            self.queue.append((C.string_at(comp, comp_bytes), blks, abs_base, total))
            self.max_in_flight = max(self.max_in_flight, len(self.queue))
            return 0

        def collect(user, out, line_state, nl, n_nl):
            raw, blks, abs_base, total = self.queue.pop(0)
            return work(raw, blks, abs_base, total, out, line_state, nl, n_nl)

        def work(raw, blks, abs_base, total, out, line_state, nl, n_nl):
            self.calls += 1
            self.max_seen = max(getattr(self, 'max_seen', 0), len(blks))
            assert abs_base == len(self.text)
            before = bytes(self.text[-1:])      # (a '\r' there belongs to a newline that opens this run)
            seg = bytearray(total)
            for off, ln, isize, dst in blks:
                d = zlib.decompress(raw[off:off + ln], -15)
                assert len(d) == isize
                seg[dst:dst + isize] = d
            self.text += seg
            # poison the reader's buffer, then put the heads where they belong
            if total:
                C.memset(out, 0xEE, total)
            tabs = line_state[0]
            pos, nls = 0, []
            seg_b = bytes(seg)
            while True:
                e = seg_b.find(b'\n', pos)
                end = e if e >= 0 else total
                # head of [pos, end): up to and including tab number 9 - tabs
                k, he = pos, None
                need = 9 - tabs
                if need <= 0:
                    he = pos
                else:
                    found = 0
                    while found < need:
                        t = seg_b.find(b'\t', k, end)
                        if t < 0:
                            break
                        found += 1
                        k = t + 1
                    he = k if found == need else end
                    tabs_now = min(9, tabs + found)
                if he > pos:
                    C.memmove(out + pos, seg_b[pos:he], he - pos)
                if e < 0:
                    line_state[0] = 9 if need <= 0 else tabs_now
                    break
                nls.append(e | ((1 << 63) if (seg_b[e - 1:e] if e > 0 else before) == b'\r' else 0))
                tabs = 0
                pos = e + 1
            arr = (C.c_uint64 * max(len(nls), 1))(*nls)
            self.keep = [arr]
            nl[0] = C.cast(arr, C.POINTER(C.c_uint64))
            n_nl[0] = len(nls)
            return 0
        self._seed, self._inflate = SEED_FN(seed), INFLATE_FN(inflate)
        self._submit, self._collect = SUBMIT_FN(submit), COLLECT_FN(collect)

    def struct(self, max_members=0, pipelined=False):
        from trtools_amd.vcfnative import _InflateHook
        h = _InflateHook(None, C.cast(self._seed, C.c_void_p).value, C.cast(self._inflate, C.c_void_p).value, max_members)
        if pipelined:
            h.submit, h.collect = C.cast(self._submit, C.c_void_p).value, C.cast(self._collect, C.c_void_p).value
        return h


def _batches(path, hooked, batch_records, keys=('DP', 'Q'), max_members=0, pipelined=False):
    """(per record: head text, line length, field offsets, fmt idx), harmonised lists -- read with or without a hook."""
    from trtools_amd import vcfnative
    r = vcfnative.NativeVCFReader(path, batch_records=batch_records)
    for k in keys:
        if k in r.format_types:
            r.select_format(k)
    r._lib.trk_vcf_skip_samples(r._h, 1)
    hook = None
    if hooked:
        hook = PyHook()
        hs = hook.struct(max_members, pipelined)
        assert r._lib.trk_vcf_set_inflate_hook(r._h, C.byref(hs)) == 0, r._lib.trk_vcf_last_error(r._h)
        r._keep_hook = (hook, hs)
    out, absolute = [], []
    while True:
        rb = r._read_raw_batch(batch_records)
        if rb.n == 0:
            break
        b = rb.b
        stride = C.c_int32()
        fptr = r._lib.trk_vcf_format_idx(r._h, C.byref(stride))
        fi = np.ctypeslib.as_array(C.cast(fptr, C.POINTER(C.c_int8)), shape=(rb.n, stride.value)).copy()
        abs0 = int(r._lib.trk_vcf_text_abs(r._h))
        for l in range(rb.n):
            fo = [int(b.field_off[l * 10 + k]) for k in range(10)]
            head = C.string_at(b.text + b.line_off[l], fo[9] if fo[9] > 0 else b.line_end[l] - b.line_off[l])
            out.append((head, int(b.line_end[l] - b.line_off[l]), tuple(fo), tuple(fi[l])))
            absolute.append((abs0 + int(b.line_off[l]), abs0 + int(b.line_end[l])))
        hz = rb.harmonize('hipstr') if 'hipstr' in os.path.basename(path) else None
        if hz is not None:
            out.append(('hz', tuple(map(str, hz.lists()[:3]))))
    r.close()
    return out, absolute, hook


FILES = [os.path.join(GOLDEN, 'dumpstr_synth', f) for f in ('synth_hipstr.vcf', 'synth_gangstr.vcf')
         if os.path.exists(os.path.join(GOLDEN, 'dumpstr_synth', f))]


def _bgzip(tmp_path, name, text, level=6):
    from trtools_amd.bgzf import BgzfWriter
    p = str(tmp_path / name)
    with BgzfWriter(p, threads=1) as fh:
        fh.write(text)
    return p


def _bgzip_members(tmp_path, name, parts, level=6):
    """A BGZF file whose members hold exactly `parts` (bytes each, <= 65280): the test decides what a member boundary cuts."""
    from trtools_amd.bgzf import _compress_block, _EOF
    p = str(tmp_path / name)
    with open(p, 'wb') as fh:
        for part in parts:
            assert 0 < len(part) <= 0xff00
            fh.write(_compress_block((bytes(part), level)))
        fh.write(_EOF)
    return p


def _synthetic(n_rec, S, crlf=False, last_newline=True, seed=0):
    rng = np.random.default_rng(seed)
    nl = '\r\n' if crlf else '\n'
    hdr = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 x', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
           '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">',
           '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    lines = []
    for r in range(n_rec):
        cols = ['%d|%d:%d:0.%02d' % (rng.integers(0, 3), rng.integers(0, 3), rng.integers(0, 90), rng.integers(0, 100)) for _ in range(S)]
        alt = ','.join('AC' * int(k) for k in rng.integers(1, 40, size=int(rng.integers(2, 5))))
        lines.append('\t'.join(['chr1', str(1000 + 37 * r), 'id%d' % r, 'ACAC', alt, '.', '.', 'START=%d;END=%d;PERIOD=2' % (1000 + 37 * r, 1003 + 37 * r),
                                'GT:DP:Q'] + cols))
    text = nl.join(hdr + lines) + (nl if last_newline else '')
    return text.encode()


@pytest.mark.parametrize("case", ["long rows", "short rows", "crlf", "no last newline", "one record"])
def test_hooked_read_equals_the_plain_read(tmp_path, case):
    text = {"long rows": lambda: _synthetic(40, 6000, seed=1), "short rows": lambda: _synthetic(3000, 3, seed=2),
            "crlf": lambda: _synthetic(60, 900, crlf=True, seed=3), "no last newline": lambda: _synthetic(25, 700, last_newline=False, seed=4),
            "one record": lambda: _synthetic(1, 5, seed=5)}[case]()
    path = _bgzip(tmp_path, 'f.vcf.gz', text)
    from trtools_amd import _lib as L
    for br, min_read, mm, pl in ((7, 70000, 0, False), (64, 300000, 0, False), (16, 8 << 20, 0, False), (16, 8 << 20, 3, False),
                                 (7, 70000, 1, False), (7, 70000, 0, True), (16, 8 << 20, 3, True), (64, 300000, 1, True)):
        # (small reads of the compressed file, so that a file of a few megabytes takes many fills: members, heads and
        # CRLF pairs cut by the fill boundaries; mm: the hook's max_members -- runs of at most so many members; pl: the
        # hook's submit / collect halves -- two runs in flight)
        with L.options(TRK_VCF_READ_BYTES=min_read):
            plain, abs_p, _ = _batches(path, False, br)
            hooked, abs_h, hook = _batches(path, True, br, max_members=mm, pipelined=pl)
        assert len(plain) == len(hooked) and len(plain) > 0
        for a, b in zip(plain, hooked):
            assert a == b
        # the absolute offsets the hooked reader reports point at the lines in the hook's copy of the stream
        full = bytes(hook.text)
        recs = [x for x in hooked if x[0] != 'hz']
        for (lo, le), rec in zip(abs_h, recs):
            assert full[lo:lo + len(rec[0])] == rec[0] and (le == len(full) or full[le:le + 1] in (b'\n', b'\r'))
        assert hook.calls >= (1 if min_read < (1 << 20) and len(text) > 2000000 else 0)
        assert mm == 0 or hook.max_seen <= mm
        assert not hook.queue and (not pl or hook.max_in_flight <= 2)


def _crlf_cut_members(n_rec=700, S=40, seed=11):
    """CRLF text in members that END with a line's '\r' and BEGIN with its '\n': whatever run of members a hook is given,
    its boundary cuts a CRLF pair (the flag of a newline at offset 0 of a run comes from the run before)."""
    text = _synthetic(n_rec, S, crlf=True, seed=seed)
    lines = text.split(b'\r\n')
    assert lines[-1] == b''
    lines = lines[:-1]
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b'#CHROM'))
    parts = [b'\r\n'.join(lines[:k + 1]) + b'\r']              # the header and the first '\r'
    for ln in lines[k + 1:]:
        parts.append(b'\n' + ln + b'\r')
    parts.append(b'\n')
    return text, parts


def _blank_runs_text(n_rec=300, S=30, seed=12, run=200000):
    """Records between long runs of blank lines (lines of fewer than 16 bytes on average: ADVICE r05 -- the device's
    line tables were sized total / 16; the host's path skips blank lines)."""
    text = _synthetic(n_rec, S, seed=seed)
    lines = text.split(b'\n')[:-1]
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b'#CHROM'))
    out = lines[:k + 1]
    for i, ln in enumerate(lines[k + 1:]):
        if i in (0, 100, 299):
            out.append(b'\n' * run)                       # (joined below: run + 1 blank lines)
        elif i % 7 == 0:
            out.append(b'')
        out.append(ln)
    return b'\n'.join(out) + b'\n' * 5000


@pytest.mark.parametrize("mm,pl", [(1, False), (1, True), (3, True), (0, False)])
def test_crlf_pair_cut_by_every_run_boundary(tmp_path, mm, pl):
    text, parts = _crlf_cut_members()
    path = _bgzip_members(tmp_path, 'cut.vcf.gz', parts)
    import gzip
    assert gzip.open(path).read() == text
    from trtools_amd import _lib as L
    for br, min_read in ((5, 4000), (64, 1 << 20)):
        with L.options(TRK_VCF_READ_BYTES=min_read):
            plain, _, _ = _batches(path, False, br)
            hooked, abs_h, hook = _batches(path, True, br, max_members=mm, pipelined=pl)
        assert plain == hooked and len(plain) == 700
        full = bytes(hook.text)
        for (lo, le), rec in zip(abs_h, hooked):
            assert full[le:le + 2] == b'\r\n'               # the line ends IN FRONT of its '\r'


def test_runs_of_blank_lines(tmp_path):
    text = _blank_runs_text()
    path = _bgzip(tmp_path, 'blank.vcf.gz', text)
    from trtools_amd import _lib as L
    with L.options(TRK_VCF_READ_BYTES=300000):
        plain, _, _ = _batches(path, False, 64)
        hooked, _, _ = _batches(path, True, 64, max_members=2, pipelined=True)
    assert plain == hooked and len(plain) == 300


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hooked_read_of_the_fixtures(path, tmp_path):
    from trtools_amd import _lib as L
    # (the fixtures are plain text of ~50 KB: their records thirty times over, so that the reader has something to fill)
    lines = open(path, 'rb').read().split(b'\n')
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b'#CHROM'))
    text = b'\n'.join(lines[:k + 1] + [ln for _ in range(30) for ln in lines[k + 1:] if ln]) + b'\n'
    path = _bgzip(tmp_path, os.path.basename(path) + '.gz', text)
    with L.options(TRK_VCF_READ_BYTES=20000):
        plain, _, _ = _batches(path, False, 16)
        hooked, _, hook = _batches(path, True, 16)
        piped, _, hook2 = _batches(path, True, 16, max_members=2, pipelined=True)
    assert plain == hooked and plain == piped and len(plain) > 300 and hook.calls >= 2 and hook2.max_in_flight == 2


def test_hook_is_refused_where_it_cannot_work(tmp_path):
    from trtools_amd import vcfnative
    text = _synthetic(10, 5)
    plain = str(tmp_path / 'p.vcf')
    open(plain, 'wb').write(text)
    r = vcfnative.NativeVCFReader(plain)
    r._lib.trk_vcf_skip_samples(r._h, 1)
    hs = PyHook().struct()
    assert r._lib.trk_vcf_set_inflate_hook(r._h, C.byref(hs)) != 0          # not BGZF
    r.close()
    path = _bgzip(tmp_path, 'g.vcf.gz', text)
    r = vcfnative.NativeVCFReader(path)
    assert r._lib.trk_vcf_set_inflate_hook(r._h, C.byref(hs)) != 0          # the samples are not skipped
    r.close()


def test_chrom_runs_of_a_batch(tmp_path):
    """RawBatch.chroms / chrom_column (array operations on the text) against a per-record reading: contig names of
    different lengths, runs of one record, a contig that comes back."""
    from trtools_amd import vcfnative
    names = ['chr1'] * 5 + ['chr10'] * 3 + ['chr1'] + ['X'] * 4 + ['chrUn_gl000220'] + ['X'] + ['chr2'] * 6
    hdr = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 x', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts0\ts1']
    lines = ['%s\t%d\t.\tACAC\tACACAC\t.\t.\tSTART=%d;END=%d;PERIOD=2\tGT\t0|1\t1|1' % (c, 100 + 10 * i, 100 + 10 * i, 103 + 10 * i)
             for i, c in enumerate(names)]
    path = str(tmp_path / 'c.vcf')
    open(path, 'w').write('\n'.join(hdr + lines) + '\n')
    for br in (len(names), 4, 1):
        r = vcfnative.NativeVCFReader(path, batch_records=br)
        got_col, got_distinct, at = [], [], 0
        while True:
            rb = r._read_raw_batch(br)
            if rb.n == 0:
                break
            col = rb.chrom_column()
            assert col == names[at:at + rb.n]
            want = []
            for c in col:
                if c not in want:
                    want.append(c)
            assert rb.chroms() == want
            got_col += col
            at += rb.n
        r.close()
        assert got_col == names
