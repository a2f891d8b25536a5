"""trk_bgzf_compress (include/trk_vcf.h; round 6: dumpSTR --zip's members made by libtrk on its worker pool): every member a
gzip member of its own with the 'BC' field, 0xff00 bytes of text each, the text back byte for byte through zlib and through
the native reader; the bytes of the file depend on the text and the level alone -- not on the thread count, not on how the
writer was fed; libdeflate's members and zlib's (library option TRK_BGZF_ZLIB) both; stored members at level 0."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from test_vcfnative_hook import _synthetic


def _lib():
    from trtools_amd import bgzf
    bgzf._native = False
    lib = bgzf._native_lib()
    assert lib is not None
    return lib


def _compress(lib, text, level, threads):
    out = bytearray(lib.trk_bgzf_bound(len(text)))
    dst = (C.c_char * len(out)).from_buffer(out)
    got = C.c_size_t(0)
    rc = lib.trk_bgzf_compress(C.c_char_p(text), len(text), level, threads, dst, len(out), C.byref(got))
    del dst
    assert rc == 0
    return bytes(out[:got.value])


def _members(raw):
    pos, out = 0, []
    while pos < len(raw):
        assert raw[pos:pos + 4] == b'\x1f\x8b\x08\x04' and raw[pos + 10:pos + 16] == b'\x06\x00BC\x02\x00'
        bsize = struct.unpack_from('<H', raw, pos + 16)[0] + 1
        body = raw[pos + 18:pos + bsize - 8]
        crc, isize = struct.unpack_from('<II', raw, pos + bsize - 8)
        text = zlib.decompress(body, -15)
        assert len(text) == isize and zlib.crc32(text) & 0xffffffff == crc
        out.append(text)
        pos += bsize
    assert pos == len(raw)
    return out


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_members_hold_the_text(level):
    lib = _lib()
    rng = np.random.default_rng(level)
    text = _synthetic(120, 2000, seed=7) + bytes(rng.integers(0, 256, size=200000, dtype=np.uint8)) + b'\n' * 70000
    for n in (0, 1, 0xff00 - 1, 0xff00, 0xff00 + 1, 5 * 0xff00, len(text)):
        t = text[:n]
        raw = _compress(lib, t, level, 3)
        ms = _members(raw)
        assert b''.join(ms) == t and all(len(m) == 0xff00 for m in ms[:-1]) and (not ms or 0 < len(ms[-1]) <= 0xff00)
        assert raw == _compress(lib, t, level, 1) == _compress(lib, t, level, 16)      # the thread count changes nothing
        if level == 0 and n:
            assert len(raw) == n + 31 * len(ms)                                         # stored: 5 + 26 bytes per member
    assert len(_compress(lib, text, 6, 4)) < len(_compress(lib, text, 1, 4)) < len(_compress(lib, text, 0, 4))


def test_zlib_members_when_libdeflate_is_not_used():
    from trtools_amd import _lib as L
    lib = _lib()
    text = _synthetic(60, 3000, seed=3)
    a = _compress(lib, text, 6, 4)
    with L.options(TRK_BGZF_ZLIB=1):
        b = _compress(lib, text, 6, 4)
    assert b''.join(_members(a)) == b''.join(_members(b)) == text
    # zlib's members are the interpreter's zlib members: the Python writer's bytes
    from trtools_amd.bgzf import _compress_block, BLOCK
    assert b == b''.join(_compress_block((text[i:i + BLOCK], 6)) for i in range(0, len(text), BLOCK))


def test_bound_and_arguments():
    lib = _lib()
    got = C.c_size_t(7)
    small = bytearray(100)
    dst = (C.c_char * 100).from_buffer(small)
    assert lib.trk_bgzf_compress(C.c_char_p(b'x' * 1000), 1000, 6, 1, dst, 100, C.byref(got)) == 1 and got.value == 0
    assert lib.trk_bgzf_compress(C.c_char_p(b'x'), 1, 11, 1, dst, 100, C.byref(got)) == 2
    del dst
    lib.trk_bgzf_eof.argtypes = [C.c_void_p]
    lib.trk_bgzf_eof.restype = C.c_size_t
    e = bytearray(28)
    eb = (C.c_char * 28).from_buffer(e)
    assert lib.trk_bgzf_eof(eb) == 28
    del eb
    from trtools_amd.bgzf import _EOF
    assert bytes(e) == _EOF


def test_writer_feeds_change_nothing_and_the_reader_reads_it(tmp_path):
    """BgzfWriter over libtrk: one big write, many small ones, bytearrays -- the same file; the native reader (block-
    parallel inflate) and gzip read the text back."""
    from trtools_amd import bgzf, vcfnative
    bgzf._native = False
    text = _synthetic(400, 1500, seed=11)
    paths = []
    rng = np.random.default_rng(1)
    for k in range(3):
        p = str(tmp_path / ('f%d.vcf.gz' % k))
        with bgzf.BgzfWriter(p, threads=(None, 2, 5)[k]) as fh:
            assert fh._lib is not None
            if k == 0:
                fh.write(text)
            else:
                at = 0
                while at < len(text):
                    n = int(rng.choice([1, 300, 0xff00, 100000, 40 << 20]))
                    fh.write(text[at:at + n] if k == 1 else bytearray(text[at:at + n]))
                    at += n
        paths.append(p)
    raws = [open(p, 'rb').read() for p in paths]
    assert raws[0] == raws[1] == raws[2] and raws[0].endswith(bgzf._EOF)
    assert gzip.open(paths[0]).read() == text
    r = vcfnative.NativeVCFReader(paths[0])
    n = sum(1 for _ in r)
    r.close()
    assert n == 400


def test_a_failing_device_deflate_hands_over_to_the_host(tmp_path):
    """BgzfWriter(engine=...): blocks go to the engine's deflate_bgzf; when that raises, the host compressor makes the
    members from there on and the file is whole (no GPU needed: the engine is a stand-in)."""
    from trtools_amd import bgzf
    bgzf._native = False
    calls = []

    class Failing:
        def deflate_bgzf(self, data, address=None, nbytes=None, out=None):
            calls.append(nbytes)
            raise RuntimeError("no device")

    text = os.urandom(4000) + b'chr1\t100\tabc\n' * (3 << 20)
    path = str(tmp_path / 'f.gz')
    w = bgzf.BgzfWriter(path, engine=Failing())
    w.write(text)
    w.write(text[:100000])
    w.close()
    assert calls and w._engine is None
    assert gzip.open(path).read() == text + text[:100000]
    # the offsets the index is built from are those of the file that was written
    raw = open(path, 'rb').read()
    for k in (0, 1, len(w._coff) // 2, len(w._coff) - 2):
        assert raw[w._coff[k]:w._coff[k] + 4] == b'\x1f\x8b\x08\x04'
    assert w._coff[-1] == len(raw) - 28 and w._toff[-1] == len(text) + 100000


def test_member_offsets_follow_the_chain():
    """trk_bgzf_member_offsets: where the members of a buffer begin (what BgzfWriter's virtual offsets are made of), against a
    walk of the BSIZE fields here; a buffer that is not whole members is refused, a short table is said to be short."""
    lib = _lib()
    text = _synthetic(60, 700, seed=4)
    raw = _compress(lib, text, 1, 4) + bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
    want, pos = [], 0
    while pos < len(raw):
        want.append(pos)
        pos += struct.unpack_from('<H', raw, pos + 16)[0] + 1
    assert len(want) == (len(text) + 0xfeff) // 0xff00 + 1
    out = np.zeros(len(want) + 3, dtype=np.uint64)
    walk = lambda buf, cap: int(lib.trk_bgzf_member_offsets(C.c_char_p(buf), len(buf), out.ctypes.data, cap))
    assert walk(raw, len(out)) == len(want) and out[:len(want)].tolist() == want and out[len(want):].tolist() == [0, 0, 0]
    out[:] = 0
    assert walk(raw, 2) == len(want) and out[:3].tolist() == want[:2] + [0]          # (the count, the first two places)
    assert walk(raw[:-1], len(out)) == -1 and walk(raw[:want[1] + 5], len(out)) == -1
    assert walk(b'x' + raw, len(out)) == -1 and walk(b'', len(out)) == 0
