"""Tabix indexes without htslib tools (trtools_amd/tabix.py) and the native reader's index seek.
Pinned by the .tbi files htslib wrote for the reference's fixture VCFs."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, 'tests', 'golden', 'data')
FIXTURES = sorted(f for f in glob.glob(DATA + '/**/*.vcf.gz', recursive=True) if os.path.exists(f + '.tbi'))


@pytest.mark.parametrize('vcf', FIXTURES, ids=[os.path.basename(f) for f in FIXTURES])
def test_built_index_agrees_with_htslib(vcf, tmp_path):
    from trtools_amd import tabix
    ref = tabix.TabixIndex.load(vcf + '.tbi')
    blocks = {c for c, _, _ in tabix._blocks(vcf)}
    if any((v >> 16) not in blocks for lin in ref.linear for v in lin):
        pytest.skip('the fixture index is stale: it points between the BGZF blocks of this file')
    out = str(tmp_path / 'x.tbi')
    mine = tabix.build(vcf, out)
    back = tabix.TabixIndex.load(out)
    assert back.names == mine.names == ref.names
    assert back.linear == mine.linear and back.bins == mine.bins          # write/read round trip
    assert ref.meta == back.meta
    for r in range(len(ref.names)):
        # htslib's meta bin: first virtual offset of the sequence and its record count.  The END offset is where
        # the writing htslib version's bgzf_tell stood after the last record (older versions: the start of its
        # block): not compared beyond ordering
        mm, rm = mine.bins[r][tabix.META_BIN], ref.bins[r][tabix.META_BIN]
        assert mm[0][0] == rm[0][0] and mm[1] == rm[1] and mm[0][1] >= rm[0][1] > mm[0][0]
        a, b = ref.linear[r], mine.linear[r]
        assert len(a) == len(b)
        # windows in which a record STARTS carry the same offset in any htslib version; empty windows are
        # filled from a neighbour (which one changed between htslib versions): both fills must be covered
        fwd = all(x == y for x, y in zip(a, b))
        filled = set(a)
        assert fwd or all(y in filled for y in b)
        starts = set()
        for s, e, line in tabix._lines(vcf):
            if line and line[:1] != b'#':
                starts.add(s)
        assert set(b) <= starts and set(a) <= starts
        # every chunk of the binning index starts at a record and the bins are the records' bins
        for bin_id, chunks in mine.bins[r].items():
            if bin_id != tabix.META_BIN:
                assert all(c[0] in starts and c[1] > c[0] for c in chunks)


def _records(reader, region=None):
    it = reader(region) if region else reader
    return [(v.CHROM, v.POS, v.ID) for v in it]


def test_region_seek_returns_what_the_full_scan_returns(tmp_path):
    from trtools_amd import vcfnative, vcfio
    vcf = os.path.join(DATA, 'many_samples.vcf.gz')
    rng = np.random.default_rng(3)
    allpos = [p for _, p, _ in _records(vcfnative.NativeVCFReader(vcf))]
    regions = ['1:1000000-2000000', '1:1-20000', '1:%d-%d' % (allpos[-1], allpos[-1] + 10), '1:9000000-9999999', '2:1-100', '1']
    for _ in range(12):
        a = int(rng.integers(1, allpos[-1]))
        regions.append('1:%d-%d' % (a, a + int(rng.integers(1, 400000))))
    for reg in regions:
        seek = vcfnative.NativeVCFReader(vcf)
        got = _records(seek, reg)
        want = _records(vcfio.VCFReader(vcf), reg)              # python decoder: always a linear scan
        assert got == want, reg
        if ':' in reg and got:
            assert seek._indexed_region
    os.environ['TRK_TABIX'] = '0'
    try:
        r = vcfnative.NativeVCFReader(vcf)
        assert _records(r, regions[0]) == _records(vcfio.VCFReader(vcf), regions[0]) and not r._indexed_region
    finally:
        del os.environ['TRK_TABIX']


def test_seek_reads_fewer_bytes(tmp_path):
    """A large bgzipped file: a region near the end is served without inflating the whole file."""
    from trtools_amd import bgzf, tabix, vcfnative
    path = str(tmp_path / 'big.vcf.gz')
    with bgzf.BgzfWriter(path) as fh:
        fh.write('##fileformat=VCFv4.2\n##contig=<ID=chr1>\n##FORMAT=<ID=GT,Number=1,Type=String,Description="g">\n')
        fh.write('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(200)) + '\n')
        gts = '\t'.join(['0/1'] * 200)
        for i in range(6000):
            fh.write('chr1\t%d\tr%d\tACAC\tAC\t.\t.\t.\tGT\t%s\n' % (1000 + 37 * i, i, gts))
    idx = tabix.build(path)
    assert len(idx.linear[0]) == (1000 + 37 * 5999 + 3) // 16384 + 1
    r = vcfnative.NativeVCFReader(path)
    got = [v.POS for v in r('chr1:200000-200100')]
    assert got == [p for p in range(1000, 1000 + 37 * 6000, 37) if p + 3 >= 200000 and p <= 200100]
    assert r._indexed_region


def test_dumpstr_zip_writes_an_index(tmp_path):
    from trtools_amd import tabix
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    from oracle_compute import OracleCompute
    import gen_golden_dumpstr as gg
    old = runtime.set_compute(OracleCompute())
    try:
        caller, kw = gg.CASES['hipstr_all']
        args = gg.make_args(str(tmp_path / 'z'), os.path.join(ROOT, 'tests', 'golden', 'dumpstr_synth', 'synth_hipstr.vcf'),
                            caller, **kw)
        args.zip = True
        argv, sys.argv = sys.argv, ['dumpSTR']
        try:
            assert dumpSTR.main(args) == 0
        finally:
            sys.argv = argv
    finally:
        runtime.set_compute(old)
    out = str(tmp_path / 'z.vcf.gz')
    idx = tabix.TabixIndex.load(out + '.tbi')
    n = sum(1 for s, e, l in tabix._lines(out) if l and l[:1] != b'#')
    assert sum(idx.bins[r][tabix.META_BIN][1][0] for r in range(len(idx.names))) == n > 0
    # round 6: the index comes from the places the writer noted while it wrote (VCFWriter.write_index) -- it must be the
    # index a scan of the finished file gives
    scan = tabix.build(out, str(tmp_path / 'scan.tbi'))
    assert (idx.names, idx.bins, idx.linear, idx.meta) == (scan.names, scan.bins, scan.linear, scan.meta)


@pytest.mark.parametrize('native', [True, False])
def test_writer_index_equals_the_scan_of_the_file(tmp_path, native):
    """VCFWriter (--zip) notes where its records lie and writes the .tbi itself: records written one by one, as text and as
    blocks of bytes on the writer thread; lines longer than a BGZF member, a line that ends with its member, several
    contigs -- against tabix.build's scan of the finished file, with libtrk's members and with the interpreter's."""
    from trtools_amd import bgzf, tabix, vcfio
    os.environ['TRK_BGZF_PYTHON'] = '0' if native else '1'
    bgzf._native = False
    try:
        rng = np.random.default_rng(5)
        hdr = ['##fileformat=VCFv4.2', '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
               '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(30000))]

        class _Tmpl:
            _header_lines, _chrom_line = hdr[:-1], hdr[-1]
            has_pass_filter, contigs_seen = True, []
        recs, pos = [], 100
        for chrom in ('chr1', 'chr2', 'chrX'):
            for i in range(60):
                pos += int(rng.integers(1, 40000))
                nsamp = int(rng.choice([3, 50, 30000]))
                ref = 'AC' * int(rng.integers(1, 30))
                info = 'END=%d;PERIOD=2' % (pos + len(ref) + int(rng.integers(0, 50))) if rng.random() < 0.5 else '.'
                recs.append('\t'.join([chrom, str(pos), '.', ref, 'ACAC', '.', '.', info, 'GT'] + ['0/1'] * nsamp) + '\n')
            pos = 50
        path = str(tmp_path / 'w.vcf.gz')
        w = vcfio.VCFWriter(path, _Tmpl())
        assert bool(w._fh._lib) == native
        k = 0
        while k < len(recs):
            how, n = int(rng.integers(0, 3)), int(rng.integers(1, 25))
            chunk = recs[k:k + n]
            k += n
            if how == 0:
                for r in chunk:
                    w.write_record(r)
            elif how == 1:
                w.write_text(''.join(chunk))
            elif k % 2:
                w.write_bytes(''.join(chunk).encode())
            else:                                   # (what the batch record writer hands over: a view of its output block)
                w.write_bytes(memoryview(np.frombuffer(''.join(chunk).encode(), dtype=np.uint8).copy()))
        w.close()
        mine = w.write_index()
        scan = tabix.build(path, str(tmp_path / 'scan.tbi'))
        assert (mine.names, mine.bins, mine.linear) == (scan.names, scan.bins, scan.linear)
        back = tabix.TabixIndex.load(path + '.tbi')
        assert (back.names, back.bins, back.linear, back.meta) == (scan.names, scan.bins, scan.linear, scan.meta)
        import gzip
        assert gzip.open(path).read().decode().split('\n')[len(hdr):-1] == [r[:-1] for r in recs]
        # unsorted records: what `tabix` refuses
        w2 = vcfio.VCFWriter(str(tmp_path / 'u.vcf.gz'), _Tmpl())
        w2.write_record(recs[5])
        w2.write_record(recs[2])
        w2.close()
        with pytest.raises(ValueError):
            w2.write_index()
    finally:
        del os.environ['TRK_BGZF_PYTHON']
        bgzf._native = False


def test_stale_index_falls_back_to_a_scan():
    """trio_chr21_hipstr's fixture index was written for another compression of the file: the seek is refused and
    the region is served by the linear scan."""
    from trtools_amd import vcfnative, vcfio
    vcf = os.path.join(DATA, 'dumpSTR', 'trio_chr21_hipstr.sorted.vcf.gz')
    r = vcfnative.NativeVCFReader(vcf)
    got = _records(r, 'chr21:15000000-15500000')
    assert not r._indexed_region and len(got) > 10
    assert got == _records(vcfio.VCFReader(vcf), 'chr21:15000000-15500000')


from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402


@settings(max_examples=60, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n_rec=st.integers(1, 400), n_chrom=st.integers(1, 3), S=st.integers(1, 30),
       spread=st.sampled_from([50, 5000, 200000]))
def test_random_files_index_seek_equals_scan(tmp_path_factory, seed, n_rec, n_chrom, S, spread):
    """Random position-sorted bgzip VCFs (several contigs, dense and sparse positions, long reference alleles that
    span 16 kb windows, many BGZF blocks): tabix.build + index-driven region reads return exactly what a linear scan
    with the Python decoder returns, for random regions including empty ones and regions past the last record."""
    from trtools_amd import tabix, vcfio, vcfnative
    from trtools_amd.bgzf import BgzfWriter
    rng = np.random.default_rng(seed)
    d = tmp_path_factory.mktemp('tbx')
    path = str(d / 'r.vcf.gz')
    chroms = ['chr%d' % (c + 1) for c in range(n_chrom)]
    per = np.sort(rng.integers(0, n_chrom, size=n_rec))
    lines = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 fuzz', '##INFO=<ID=END,Number=1,Type=Integer,Description="e">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
             '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    last = {}
    recs = []
    for r in range(n_rec):
        c = chroms[per[r]]
        pos = last.get(c, 0) + int(rng.integers(0, spread)) + (1 if c not in last else 0)
        pos = max(pos, 1)
        last[c] = pos
        ref = 'AC' * int(rng.integers(1, 40)) if rng.random() < 0.9 else 'A' * int(rng.integers(1, 30000))
        recs.append((c, pos))
        lines.append('\t'.join([c, str(pos), 'r%d' % r, ref, 'ACAC', '.', '.', '.', 'GT'] + ['0/1'] * S))
    with BgzfWriter(path, threads=1) as fh:
        fh.write('\n'.join(lines) + '\n')
    tabix.build(path)
    regions = []
    for _ in range(8):
        c = chroms[int(rng.integers(0, n_chrom))]
        hi = last.get(c, 1000)
        a = int(rng.integers(1, hi + 2000))
        regions.append('%s:%d-%d' % (c, a, a + int(rng.integers(0, max(2, spread * 3)))))
    regions += [chroms[0], '%s:%d-%d' % (chroms[-1], last.get(chroms[-1], 1) + 10, last.get(chroms[-1], 1) + 500), 'chrNone:1-10']
    for reg in regions:
        got = _records(vcfnative.NativeVCFReader(path), reg)
        want = _records(vcfio.VCFReader(path), reg)
        assert got == want, reg


def test_native_record_places_equal_the_python_scan(tmp_path):
    """trk_text_record_places (what VCFWriter notes of a block of the batch writer) against the writer's own per-line scan:
    INFO/END before, behind and without other items, END spellings python's int() reads and this scanner leaves to it, a
    name that is not END, heads of four columns, names beyond ASCII, comment and empty lines, a last line without newline."""
    from trtools_amd import vcfio
    rng = np.random.default_rng(9)
    lines = ['#comment\n', '\n']
    pos = 10
    infos = ['.', 'END=%d', 'PERIOD=2;END=%d', 'END=%d;X=1', 'XEND=%d', 'END=+%d', 'END= %d', 'END=', 'END=1_0', 'END=abc;END=%d',
             'A=1;END=%d;END=7', 'END=0', '']
    for chrom in ('chr1', 'chrü', 'c'):
        for i in range(80):
            pos += int(rng.integers(1, 500))
            ref = 'ACG' * int(rng.integers(1, 9))
            info = infos[int(rng.integers(0, len(infos)))]
            if '%d' in info:
                info = info % (pos + int(rng.integers(-5, 60)))
            cols = [chrom, str(pos) if rng.random() < 0.9 else ' %d' % pos, '.', ref, 'A', '.', '.', info, 'GT', '0/1', '1/1']
            if rng.random() < 0.1:
                cols = cols[:int(rng.integers(4, 9))]
            lines.append('\t'.join(cols) + '\n')
            if rng.random() < 0.05:
                lines.append('#in between\n')
        pos = 5
    lines[-1] = lines[-1][:-1]                     # (no newline at the very end)
    text = ''.join(lines).encode()

    class _Tmpl:
        _header_lines, _chrom_line = ['##fileformat=VCFv4.2'], '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb'
        has_pass_filter, contigs_seen = True, []

    def noted(native_min):
        w = vcfio.VCFWriter(str(tmp_path / ('n%d.vcf.gz' % native_min)), _Tmpl())
        assert w._fh._lib is not None
        w.NOTE_NATIVE_MIN = native_min
        w._note_block(memoryview(np.frombuffer(text, dtype=np.uint8).copy()), 1000)
        out = []
        for rec in w._recs:
            if len(rec) == 2:
                out += [(rec[0][a], b, c, d, e) for a, b, c, d, e in rec[1].tolist()]
            else:
                out.append(rec)
        w.close()
        return out
    native, python = noted(0), noted(1 << 40)
    assert len(python) == 240 and native == python
