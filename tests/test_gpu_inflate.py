"""GPU parity: trk_inflate_blocks (BGZF members inflated on the device, include/trk.h) against zlib -- the text must be
zlib's byte for byte on every member of every fixture file, on raw DEFLATE streams of every level and strategy zlib
writes (stored, fixed and dynamic blocks, several blocks per member, matches at every distance up to 32 KiB, runs
whose period is shorter than the match), and a member zlib refuses (corrupt bits, truncated payload, a wrong ISIZE) must
come back FLAGGED, with nothing written outside its own span of the text buffer."""
import glob
import os
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0, reserve_pair_gb=0)
    yield e
    e.close()


def bgzf_members(raw):
    """[(payload offset, payload length, ISIZE)] of a BGZF file's members (header walk, as trk_vcf.cpp does it)."""
    out, p = [], 0
    while p + 18 <= len(raw):
        assert raw[p] == 0x1f and raw[p + 1] == 0x8b and raw[p + 2] == 8 and raw[p + 3] & 4
        xlen = raw[p + 10] | (raw[p + 11] << 8)
        q, bsize = p + 12, None
        while q + 4 <= p + 12 + xlen:
            slen = raw[q + 2] | (raw[q + 3] << 8)
            if raw[q] == 66 and raw[q + 1] == 67 and slen == 2:
                bsize = (raw[q + 4] | (raw[q + 5] << 8)) + 1
            q += 4 + slen
        assert bsize
        isize = struct.unpack('<I', raw[p + bsize - 4:p + bsize])[0]
        out.append((p + 12 + xlen, bsize - 12 - xlen - 8, isize))
        p += bsize
    return out


def run(eng, comp, members):
    """members: [(payload offset, payload length, text length)] -> (text bytes per member, flags)."""
    in_off = np.array([m[0] for m in members], np.int64)
    in_len = np.array([m[1] for m in members], np.int32)
    out_len = np.array([m[2] for m in members], np.int32)
    gap = 64                                        # guard bytes between the members' spans: must stay untouched
    out_off = np.zeros(len(members), np.int64)
    if len(members) > 1:
        out_off[1:] = np.cumsum(out_len[:-1] + gap)
    total = int(out_off[-1] + out_len[-1] + gap) if len(members) else gap
    text = eng.empty((total + 32,), np.uint8)
    text.set(np.full(total + 32, 0xA5, np.uint8))
    text, flags = eng.inflate_blocks(comp, in_off, in_len, out_off, out_len, text=text)
    host = text.get()
    text.free()
    parts = []
    for o, n in zip(out_off, out_len):
        parts.append(bytes(host[o:o + n]))
        assert np.all(host[o + n:o + n + gap] == 0xA5), "bytes written beyond a member's span"
    return parts, flags


FILES = sorted(glob.glob(os.path.join(GOLDEN, '**', '*.vcf.gz'), recursive=True))


def test_every_member_of_every_fixture(eng):
    n_files = n_members = 0
    for path in FILES:
        raw = open(path, 'rb').read()
        if len(raw) < 18 or not (raw[3] & 4 and raw[12:14] == b'BC'):
            continue                                 # (a plain gzip fixture)
        mem = bgzf_members(raw)
        parts, flags = run(eng, raw, mem)
        for (o, n, isz), got, fl in zip(mem, parts, flags):
            want = zlib.decompress(raw[o:o + n], -15)
            assert fl == 0 and got == want, (path, o, int(fl))
        n_files += 1
        n_members += len(mem)
    assert n_files >= 5 and n_members >= n_files


def _texts(rng):
    """Payloads of every kind: VCF-like lines, digits, random bytes, zeros, short periods, long far matches."""
    line = lambda: ('chr1\t%d\t.\tACAC\tAC\t.\t.\tSTART=1\tGT:DP:Q\t' % rng.integers(1, 10**7) +
                    '\t'.join('%d|%d:%d:0.%02d' % (rng.integers(0, 3), rng.integers(0, 3), rng.integers(0, 80), rng.integers(0, 100))
                              for _ in range(int(rng.integers(1, 400)))) + '\n').encode()
    vcf = b''.join(line() for _ in range(40))[:65280]
    yield 'vcf', vcf
    yield 'vcf-short', vcf[:777]
    yield 'one byte', b'x'
    yield 'random', rng.integers(0, 256, size=65536, dtype=np.uint8).tobytes()
    yield 'random-short', rng.integers(0, 256, size=300, dtype=np.uint8).tobytes()
    yield 'zeros', bytes(65536)
    yield 'period 3', (b'abc' * 22000)[:65536]
    yield 'period 70', (bytes(range(70)) * 1000)[:65000]
    far = rng.integers(0, 256, size=32768, dtype=np.uint8).tobytes()
    yield 'far matches', (far + far)[:65536]                 # distance 32768 exactly
    yield 'digits', ''.join(str(int(x)) for x in rng.integers(0, 10**9, size=8000)).encode()[:65536]
    yield 'two symbols', bytes(rng.integers(0, 2, size=60000, dtype=np.uint8) + 65)
    yield 'skewed', bytes(np.minimum(rng.geometric(0.3, size=65536), 255).astype(np.uint8))
    yield 'all bytes', bytes(range(256)) * 200


def test_raw_deflate_streams_of_every_level_and_strategy(eng):
    rng = np.random.default_rng(7)
    comp, members, wants, names = bytearray(), [], [], []
    for name, data in _texts(rng):
        for level in (0, 1, 2, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                if level in (4, 6) and strategy == zlib.Z_DEFAULT_STRATEGY and len(data) > 2000:
                    # several blocks in one member: a sync flush (an empty stored block) and a full flush in the middle
                    z = c.compress(data[:len(data) // 3]) + c.flush(zlib.Z_SYNC_FLUSH) + \
                        c.compress(data[len(data) // 3:2 * len(data) // 3]) + c.flush(zlib.Z_FULL_FLUSH) + \
                        c.compress(data[2 * len(data) // 3:]) + c.flush()
                else:
                    z = c.compress(data) + c.flush()
                pad = int(rng.integers(0, 4))          # payloads at every byte alignment
                comp += bytes(pad)
                members.append((len(comp), len(z), len(data)))
                comp += z + bytes(8)                    # (room for a trailer, as in a BGZF file)
                wants.append(data)
                names.append((name, level, strategy))
    parts, flags = run(eng, bytes(comp), members)
    for nm, got, want, fl in zip(names, parts, wants, flags):
        assert fl == 0, (nm, int(fl))
        assert got == want, (nm, next(i for i, (x, y) in enumerate(zip(got, want)) if x != y))
    assert len(members) > 300


def test_members_zlib_refuses_are_flagged(eng):
    from trtools_amd import _lib as L
    rng = np.random.default_rng(11)
    data = b''.join(b'chr1\t%d\tACGT\t0|1:30:0.9\n' % i for i in range(3000))[:60000]
    good = zlib.compressobj(6, zlib.DEFLATED, -15).compress(data) + zlib.compressobj(6, zlib.DEFLATED, -15).flush()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    good = c.compress(data) + c.flush()
    cases = [('good', good, len(data))]
    for k in range(60):                                   # flipped bits anywhere in the stream
        b = bytearray(good)
        i = int(rng.integers(0, len(b)))
        b[i] ^= 1 << int(rng.integers(0, 8))
        cases.append(('flip %d' % i, bytes(b), len(data)))
    cases.append(('truncated', good[:len(good) // 2], len(data)))
    cases.append(('isize too small', good, len(data) - 100))
    cases.append(('isize too large', good, len(data) + 100))
    cases.append(('reserved block type', bytes([0x07]) + good, len(data)))
    cases.append(('stored with a bad NLEN', bytes([0x01, 0x05, 0x00, 0x00, 0x00]) + b'hello', 5))
    cases.append(('empty payload', b'', 10))
    comp, members = bytearray(), []
    for _, z, n in cases:
        members.append((len(comp), len(z), n))
        comp += z + bytes(8)
    parts, flags = run(eng, bytes(comp), members)
    n_flagged = 0
    for (name, z, n), got, fl in zip(cases, parts, flags):
        try:
            d = zlib.decompressobj(-15)
            want = d.decompress(z)
            ok = d.eof and len(want) == n and not d.unused_data
        except zlib.error:
            ok = False
        if ok:
            assert fl == 0 and got == want, name
        else:
            assert fl != 0, name
            n_flagged += 1
    assert flags[0] == 0 and n_flagged >= 6


def test_a_thousand_members_at_once(eng):
    """More members than waves resident on the chip (the kernel walks its share), sizes from 0 to 64 KiB."""
    rng = np.random.default_rng(3)
    base = b''.join(b'%d\t%d|%d:%d\n' % (i, i % 3, i % 2, i % 61) for i in range(9000))
    comp, members, wants = bytearray(), [], []
    for k in range(1500):
        n = int(rng.integers(0, 65537)) if k % 7 else 0
        o = int(rng.integers(0, max(1, len(base) - n)))
        data = base[o:o + n]
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        z = c.compress(data) + c.flush()
        members.append((len(comp), len(z), len(data)))
        comp += z + bytes(8)
        wants.append(data)
    parts, flags = run(eng, bytes(comp), members)
    assert not flags.any()
    assert all(g == w for g, w in zip(parts, wants))
