"""Native record serialiser (trk_vcf_format_samples, include/trk_vcf.h) against the
Python formatting loop of vcfio.Variant -- which the dumpSTR golden VCFs pin to the reference's cyvcf2/htslib output
(tests/test_dumpstr_cli.py, tests/test_dumpstr_more.py) -- on every record of every fixture VCF, untouched and after
the edits dumpSTR makes (FILTER column added, filtered calls nulled)."""
import glob
import os

import numpy as np
import pytest

from trtools_amd import vcfio

DATA = os.path.join(os.path.dirname(__file__), 'golden', 'data')
FILES = sorted(glob.glob(os.path.join(DATA, '**', '*.vcf'), recursive=True) +
               glob.glob(os.path.join(DATA, '**', '*.vcf.gz'), recursive=True))


def _readers(path):
    yield vcfio.VCFReader(path)
    from trtools_amd import vcfnative
    try:
        yield vcfnative.NativeVCFReader(path)
    except (OSError, ValueError):
        return


@pytest.mark.parametrize('path', FILES, ids=[os.path.relpath(p, DATA) for p in FILES])
def test_native_text_equals_python_text(path):
    assert vcfio._serializer() is not None, "libtrk.so is not built"
    from trtools_amd.dumpSTR import dumpSTR as D
    rng = np.random.default_rng(7)
    n_checked = 0
    for rd in _readers(path):
        try:
            it = iter(rd)
            first = next(it, None)
        except (ValueError, KeyError, IndexError):
            continue        # fixtures that are malformed on purpose
        k = 0
        v = first
        while v is not None and k < 60:
            assert v.to_text(native=True) == v.to_text(native=False)
            S = v.n_samples_hint()
            if S and v.genotype is not None and 'GT' in v.FORMAT:
                # what dumpSTR does to a record (dumpSTR.py:684, 721-746)
                mask = np.zeros(S, dtype=np.uint32)
                mask[rng.random(S) < 0.3] |= np.uint32(1)
                mask[rng.random(S) < 0.2] |= np.uint32(4)
                mask[np.any(np.asarray(v.genotype.array())[:, :-1] == -1, axis=1)] = np.uint32(0x80000000)
                filtered = (mask & np.uint32(0x7fffffff)) != 0
                vals = [np.round(rng.random(S) * 50, 2), None, rng.integers(0, 1000, S).astype(float) / 7]
                col = vcfio.CallFilterColumn(mask, ['MinDP', 'unused', 'Q0.9'], vals)
                if k % 2:
                    v.set_format('FILTER', col)                  # written from the mask by the native serialiser
                else:
                    v.set_format('FILTER', col.to_array())       # the text array the reference builds
                try:
                    D._null_filtered(v, filtered, v.ploidy)
                except ValueError:
                    pass
                assert v.to_text(native=True) == v.to_text(native=False)
                assert v._samples_text_native() is not None or any(
                    v.format(key).dtype.kind not in 'ifUS' for key in v.FORMAT if key != 'GT')
            n_checked += 1
            k += 1
            try:
                v = next(it, None)
            except (ValueError, KeyError, IndexError):
                break
    assert n_checked or os.path.getsize(path) < 4096


def test_value_forms():
    """Missing / vector-end / NaN conventions and unicode strings."""
    api = vcfio._serializer()
    assert api is not None
    lib, Column = api
    import ctypes
    gt = np.array([[0, 1, 1], [-1, -1, 0], [2, -2, 0], [-2, -2, 0], [-1, 3, 1]], dtype=np.int16)
    iv = np.array([[5, -2147483647, -2147483647], [-2147483648, 7, -2147483647], [-2147483647, 1, 1],
                   [1, 2, 3], [-2147483648, -2147483648, -2147483648]], dtype=np.int32)
    fv = np.array([[0.5, np.nan], [np.nan, np.nan], [1e-7, 3.0], [np.nan, 1.25], [123456789.0, -0.0]], dtype=np.float32)
    sv = np.array(['a', '', 'x|y', 'é', 'PASS'])
    bv = np.array([b'', b'q', b'longer', b'.', b'z'])
    cols = (Column * 5)(Column(0, 3, 0, 0, gt.ctypes.data), Column(1, 3, 0, 0, iv.ctypes.data),
                        Column(2, 2, 0, 0, fv.ctypes.data), Column(4, 1, sv.dtype.itemsize, 0, sv.ctypes.data),
                        Column(3, 1, bv.dtype.itemsize, 0, bv.ctypes.data))
    buf = ctypes.create_string_buffer(1024)
    n = lib.trk_vcf_format_samples(5, 5, cols, buf, 1024)
    assert n > 0
    want = ('\t0|1:5:0.5,.:a:.' '\t./.:.,7:.:.:q' '\t2:.:1e-07,3:x|y:longer' '\t.:1,2,3:.,1.25:é:.'
            '\t.|3:.,.,.:1.23457e+08,-0:PASS:z')
    assert buf.raw[:n].decode() == want
    for x in (0.1, 1e-5, 123456.0, 1234567.0, 0.30000001192092896, 2.5e-10, 1e16, 100.0, 0.95, 3.4028234663852886e38):
        f1 = np.array([[x]], dtype=np.float32)
        c1 = (Column * 1)(Column(2, 1, 0, 0, f1.ctypes.data))
        n1 = lib.trk_vcf_format_samples(1, 1, c1, buf, 1024)
        assert buf.raw[:n1].decode() == '\t' + '%g' % float(f1[0, 0]), x
    # too small a buffer: the size comes back negated, nothing is written past the end
    small = ctypes.create_string_buffer(8)
    assert lib.trk_vcf_format_samples(5, 5, cols, small, 8) == -n
    assert lib.trk_vcf_format_samples(-1, 5, cols, small, 8) < -10**15


@pytest.mark.parametrize('path', FILES, ids=[os.path.relpath(p, DATA) for p in FILES])
def test_native_format_decode_equals_python_decode(path, monkeypatch):
    """trk_vcf_decode_formats against the Python field decoder (Variant._numeric / np.array(col)), every FORMAT
    field of every record: same dtype, shape and values (NaN == NaN)."""
    assert vcfio._serializer() is not None, "libtrk.so is not built"

    def decode_all(limit=80):
        out = []
        try:
            for k, v in enumerate(vcfio.VCFReader(path)):
                if k >= limit:
                    break
                rec = {}
                for key in v.FORMAT:
                    if key == 'GT':
                        continue
                    try:
                        rec[key] = v.format(key)
                    except (ValueError, KeyError) as e:
                        rec[key] = type(e).__name__
                out.append((rec, v._cols is None))
        except (ValueError, KeyError, IndexError):
            pass
        return out

    nat = decode_all()
    monkeypatch.setattr(vcfio, '_SERIALIZER', None)
    py = decode_all()
    assert len(nat) == len(py)
    n_native = 0
    for (a, a_native), (b, _) in zip(nat, py):
        n_native += bool(a_native)
        assert a.keys() == b.keys()
        for key in a:
            x, y = a[key], b[key]
            if isinstance(y, str):
                assert x == y, key
                continue
            assert x.dtype == y.dtype and x.shape == y.shape, (key, x.dtype, y.dtype, x.shape, y.shape)
            if x.dtype.kind == 'f':
                assert np.array_equal(x, y, equal_nan=True), key
            else:
                assert np.array_equal(x, y), key
    if nat and any(len(r) for r, _ in nat):
        assert n_native > 0, "the native decoder was never used"


def test_large_records_are_formatted_in_parallel_with_the_same_text():
    """Records with many samples are split by sample range over a thread pool: the text must be the serial text."""
    api = vcfio._serializer()
    assert api is not None
    lib, Column = api
    import ctypes
    rng = np.random.default_rng(3)
    S = 6007
    gt = rng.integers(-2, 40, size=(S, 3)).astype(np.int16)
    gt[:, 2] = rng.integers(0, 2, size=S)
    iv = rng.integers(-5, 100000, size=(S, 2)).astype(np.int32)
    iv[rng.random(S) < 0.1, 0] = -2147483648
    iv[rng.random(S) < 0.3, 1] = -2147483647
    fv = (rng.random((S, 1)) * 10.0 ** rng.integers(-6, 7, size=(S, 1))).astype(np.float32)
    fv[rng.random(S) < 0.1] = np.nan
    sv = np.array(['x' * int(n) for n in rng.integers(0, 40, size=S)])
    mask = rng.integers(0, 8, size=S).astype(np.uint32)
    mask[rng.random(S) < 0.1] = 0x80000000
    cf = vcfio.CallFilterColumn(mask, ['a', 'bb', 'ccc'], [rng.random(S), None, rng.random(S) * 100])

    def text(lo, hi):
        st = vcfio.CallFilterColumn(mask[lo:hi], cf.names, [None if v is None else v[lo:hi] for v in cf.values])
        keep = [np.ascontiguousarray(gt[lo:hi]), np.ascontiguousarray(iv[lo:hi]), np.ascontiguousarray(fv[lo:hi]),
                np.ascontiguousarray(sv[lo:hi]), st.native_struct()]
        cols = (Column * 5)(Column(0, 3, 0, 0, keep[0].ctypes.data), Column(1, 2, 0, 0, keep[1].ctypes.data),
                            Column(2, 1, 0, 0, keep[2].ctypes.data),
                            Column(4, 1, sv.dtype.itemsize, 0, keep[3].ctypes.data),
                            Column(5, 1, 0, 0, ctypes.addressof(keep[4][0])))
        cap = (hi - lo) * 400
        buf = ctypes.create_string_buffer(cap)
        n = lib.trk_vcf_format_samples(hi - lo, 5, cols, buf, cap)
        assert n > 0
        return buf.raw[:n]

    whole = text(0, S)
    pieces = b''.join(text(lo, min(S, lo + 500)) for lo in range(0, S, 500))   # 500 x 5 fields: the serial path
    assert whole == pieces
    assert text(0, S) == whole


def test_decode_edge_cases(tmp_path):
    """Short sample columns (trailing fields dropped), vectors of unequal length, '.', CRLF line ends, unicode strings,
    a non-numeric token in an Integer field (-> the Python decoder's ValueError)."""
    p = tmp_path / 'edge.vcf'
    lines = ['##fileformat=VCFv4.2',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
             '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
             '##FORMAT=<ID=AD,Number=.,Type=Integer,Description="a">',
             '##FORMAT=<ID=PL,Number=.,Type=Float,Description="p">',
             '##FORMAT=<ID=TX,Number=1,Type=String,Description="t">',
             '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\tD',
             'c\t1\t.\tAC\tACAC\t.\t.\t.\tGT:DP:AD:PL:TX\t0/1:7:3,4:0.5,1e-3,nan:é\t1|1\t.:.:.:.:.\t0/0:-3:1,2,3:.,2:x y',
             'c\t2\t.\tAC\tACAC\t.\t.\t.\tGT:DP\t0/1:5\t1/1:6\t./.:.\t0/0:7',
             'c\t3\t.\tAC\tACAC\t.\t.\t.\tGT:DP\t0/1:5\t1/1:oops\t./.:.\t0/0:7']
    p.write_bytes(('\r\n'.join(lines) + '\r\n').encode())

    def decode(native):
        old = vcfio._SERIALIZER
        if not native:
            vcfio._SERIALIZER = None
        try:
            out = []
            for v in vcfio.VCFReader(str(p)):
                rec = {}
                for key in v.FORMAT[1:]:
                    try:
                        rec[key] = v.format(key)
                    except ValueError as e:
                        rec[key] = 'ValueError'
                bad = any(isinstance(x, str) for x in rec.values())
                out.append((rec, None if bad else v.to_text(native=native)))
            return out
        finally:
            vcfio._SERIALIZER = old
    assert vcfio._serializer() is not None
    nat, py = decode(True), decode(False)
    assert len(nat) == len(py) == 3
    for (a, ta), (b, tb) in zip(nat, py):
        assert ta == tb
        for key in b:
            if isinstance(b[key], str):
                assert a[key] == b[key]
            else:
                assert a[key].dtype == b[key].dtype and a[key].shape == b[key].shape, key
                assert np.array_equal(a[key], b[key], equal_nan=(a[key].dtype.kind == 'f')), key
    assert nat[0][0]['AD'].shape == (4, 3) and nat[0][0]['AD'][1, 0] == -2147483648 and nat[0][0]['AD'][0, 2] == -2147483647
    assert nat[0][0]['TX'][0] == 'é' and nat[0][0]['TX'][1] == '.'
    assert nat[2][0]['DP'] == 'ValueError'


# ---- randomised: native serialiser / decoder against the Python definitions --------------------------------------
from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402

_FLOATS = [0.0, -0.0, 1.0, 0.1, 0.95, 1e-5, 1e-4, 123456.0, 1234567.0, 1e16, 3.4e38, 1.17549435e-38, 1e-45, float('inf'),
           float('-inf'), 0.30000001192092896, 2.5, 99999.95, 999999.5, 1e-7]


@settings(max_examples=150, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), S=st.integers(1, 60), ploidy=st.integers(1, 3), ki=st.integers(1, 4), kf=st.integers(1, 3))
def test_random_arrays_native_text_equals_python_text(seed, S, ploidy, ki, kf):
    """Variant.to_text(native=True) == to_text(native=False) on records whose FORMAT arrays are random: genotype
    sentinels and phasing, INT_MIN / vector-end patterns, float specials and values at the %g format switches, ASCII
    and non-ASCII strings, the call-filter column."""
    rng = np.random.default_rng(seed)
    header = ['##fileformat=VCFv4.2', '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
              '##FORMAT=<ID=IV,Number=.,Type=Integer,Description="i">', '##FORMAT=<ID=FV,Number=.,Type=Float,Description="f">',
              '##FORMAT=<ID=SV,Number=1,Type=String,Description="s">',
              '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    line = 'c\t10\t.\tAC\tACAC\t.\t.\t.\tGT:IV:FV:SV\t' + '\t'.join(['0/0:1:1:x'] * S)
    import io
    import tempfile
    with tempfile.NamedTemporaryFile('w', suffix='.vcf', delete=False) as fh:
        fh.write('\n'.join(header + [line]) + '\n')
        path = fh.name
    try:
        v = next(iter(vcfio.VCFReader(path)))
    finally:
        os.remove(path)
    gt = rng.integers(-2, 12, size=(S, ploidy + 1)).astype(np.int16)
    gt[:, ploidy] = rng.integers(0, 2, size=S)
    v.set_gt_array(gt)
    iv = rng.integers(-1000, 100000, size=(S, ki)).astype(np.int32)
    iv[rng.random((S, ki)) < 0.2] = -2147483648
    for s in range(S):                       # vector ends only at the tail of a row
        n_end = int(rng.integers(0, ki + 1)) if rng.random() < 0.3 else 0
        if n_end:
            iv[s, ki - n_end:] = -2147483647
    fv = rng.choice(np.array(_FLOATS + list(rng.normal(size=8) * 10.0 ** rng.integers(-8, 9, size=8))), size=(S, kf)).astype(np.float32)
    fv[rng.random((S, kf)) < 0.2] = np.nan
    sv = np.array([rng.choice(['', '.', 'PASS', 'a|b', 'é', 'x' * int(rng.integers(1, 30)), '名前']) for _ in range(S)])
    v.set_format('IV', iv)
    v.set_format('FV', fv)
    v.set_format('SV', sv)
    mask = rng.integers(0, 16, size=S).astype(np.uint32)
    mask[rng.random(S) < 0.2] = 0x80000000
    mask[rng.random(S) < 0.3] = 0
    vals = [rng.choice(np.array(_FLOATS[:14]), size=S), None, rng.integers(0, 100, size=S).astype(float), rng.random(S)]
    v.set_format('FILTER', vcfio.CallFilterColumn(mask, ['A', 'unused', 'Cc', 'd_d'], vals))
    a = v.to_text(native=True)
    b = v.to_text(native=False)
    assert a == b


@settings(max_examples=150, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), S=st.integers(1, 50), n_rec=st.integers(1, 4))
def test_random_text_native_decode_equals_python_decode(seed, S, n_rec):
    """trk_vcf_decode_formats == Variant._numeric / np.array(col) on random sample columns."""
    rng = np.random.default_rng(seed)

    def tok_i():
        return '.' if rng.random() < 0.15 else str(int(rng.integers(-10**6, 10**6)))

    def tok_f():
        if rng.random() < 0.15:
            return '.'
        x = float(rng.choice(_FLOATS[:-1] + [float(rng.normal()) * 10.0 ** int(rng.integers(-30, 30))]))
        return rng.choice(['%g' % x, '%e' % x, repr(x), '%.3f' % x if abs(x) < 1e6 else '%g' % x])

    header = ['##fileformat=VCFv4.2', '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
              '##FORMAT=<ID=IV,Number=.,Type=Integer,Description="i">', '##FORMAT=<ID=FV,Number=.,Type=Float,Description="f">',
              '##FORMAT=<ID=SV,Number=1,Type=String,Description="s">', '##FORMAT=<ID=UN,Number=1,Type=Character,Description="u">',
              '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    lines = []
    for r in range(n_rec):
        cols = []
        for s in range(S):
            toks = ['0/1', ','.join(tok_i() for _ in range(int(rng.integers(1, 4)))),
                    ','.join(tok_f() for _ in range(int(rng.integers(1, 4)))),
                    rng.choice(['.', 'x', 'a|b;c', 'é', 'long' * int(rng.integers(1, 6))]), rng.choice(['.', 'q'])]
            if rng.random() < 0.2:
                toks = toks[:int(rng.integers(1, 5))]
            cols.append(':'.join(toks))
        lines.append('c\t%d\t.\tAC\tACAC\t.\t.\t.\tGT:IV:FV:SV:UN\t' % (10 + r) + '\t'.join(cols))
    import tempfile
    with tempfile.NamedTemporaryFile('w', suffix='.vcf', delete=False, encoding='utf-8') as fh:
        fh.write('\n'.join(header + lines) + '\n')
        path = fh.name

    def decode(native):
        old = vcfio._SERIALIZER
        if not native:
            vcfio._SERIALIZER = None
        try:
            return [{k: v.format(k) for k in ('IV', 'FV', 'SV', 'UN')} for v in vcfio.VCFReader(path)]
        finally:
            vcfio._SERIALIZER = old
    try:
        assert vcfio._serializer() is not None
        nat, py = decode(True), decode(False)
    finally:
        os.remove(path)
    for a, b in zip(nat, py):
        for k in b:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (k, a[k].dtype, b[k].dtype, a[k].shape, b[k].shape)
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == 'f')), k


def test_rewrite_info_equals_parse_assign_format():
    from helpers import GOLDEN
    """vcfio.rewrite_info (the dumpSTR batch path's INFO column: tokens that already read the way they would be
    written back pass through untouched) against the general path -- parse into typed values, assign, re-serialise --
    on every record of the fixture files and on strings that must NOT take the shortcut."""
    import glob
    from trtools_amd import vcfio, vcfnative
    files = glob.glob(os.path.join(GOLDEN, 'data', 'dumpSTR', '*.vcf.gz')) + \
        glob.glob(os.path.join(GOLDEN, 'dumpstr_synth', '*.vcf')) + [os.path.join(GOLDEN, 'data', 'many_samples.vcf.gz')]
    upd = [('HRUN', 3), ('HET', 0.123456789), ('HWEP', 1e-12), ('AC', "1,2"), ('REFAC', 5)]
    n = 0
    for p in files:
        r = vcfnative.NativeVCFReader(p)
        while True:
            rb = r.read_raw_batch(500)
            if rb.n == 0:
                break
            for l in range(0, rb.n, 3):
                f = rb.head_fields(l)
                if len(f) >= 8:
                    assert vcfio.rewrite_info(r, f[7], upd) == vcfio._rewrite_info_general(r, f[7], upd), (p, f[7])
                    n += 1
        r.close()
    assert n > 5000

    class Hdr:
        info_types = {'A': ('Integer', '1'), 'F': ('Float', '1'), 'G': ('Flag', '0'), 'S': ('String', '1'),
                      'HET': ('Float', '1'), 'AC': ('Integer', 'A')}
        _parse_info = vcfio.VCFReader._parse_info
    h = Hdr()
    for t in ['.', '', 'A=1', 'A=007', 'A=+5', 'A=-5', 'A=-05', 'A=1,2,.', 'A=.', 'F=0.123456789;A=3', 'G', 'G=1',
              'S=x;S=y', 'X;Y=2;A=0', 'HET=0.5;A=1;AC=3,4', 'A=1;;S=q', 'A=-0', 'A=1e3', 'A=', 'S=', 'A=1,,2']:
        for u in (upd, []):
            res = []
            for fn in (vcfio._rewrite_info_general, vcfio.rewrite_info):
                try:
                    res.append(fn(h, t, u))
                except Exception as e:      # noqa: BLE001 -- the same failure is the same behaviour
                    res.append(('raises', type(e).__name__))
            assert res[0] == res[1], (t, u, res)
