"""trk_deflate_bgzf (include/trk.h; round 6): BGZF members DEFLATED on the device.  The requirement: every member is a gzip
member with the 'BC' field whose stream zlib inflates to the member's text (16 KB: TRK_DEFLATE_MEMBER), with the right CRC-32 and
ISIZE -- checked on VCF text, runs, random bytes (stored members), every length around the member size, and through the
native reader.  Beyond it: the payload equals tests/deflate_model.py's stream byte for byte (the kernel's line-by-line
model), so that a difference names the stage that went wrong."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import deflate_model as dm
from helpers import GOLDEN
from test_vcfnative_hook import _synthetic

pytestmark = pytest.mark.gpu
COUNTS = {'cases': 0, 'members': 0, 'bytes': 0, 'model': 0}


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0, reserve_pair_gb=0)
    yield e
    e.close()
    if COUNTS['cases']:
        print("\n[device deflate fuzz] %(cases)d texts, %(members)d members, %(bytes)d bytes: every member inflated by zlib, "
              "%(model)d compared with the model byte for byte" % COUNTS)


def _members(raw):
    pos, out = 0, []
    while pos < len(raw):
        assert raw[pos:pos + 4] == b'\x1f\x8b\x08\x04' and raw[pos + 10:pos + 16] == b'\x06\x00BC\x02\x00', pos
        bsize = struct.unpack_from('<H', raw, pos + 16)[0] + 1
        payload = bytes(raw[pos + 18:pos + bsize - 8])
        crc, isize = struct.unpack_from('<II', raw, pos + bsize - 8)
        text = zlib.decompress(payload, -15)
        assert len(text) == isize and zlib.crc32(text) & 0xffffffff == crc, pos
        out.append((text, payload))
        pos += bsize
    assert pos == len(raw)
    return out


def _texts():
    rng = np.random.default_rng(3)
    vcf = _synthetic(60, 2500, seed=5)
    fixture = open(os.path.join(GOLDEN, 'dumpstr_synth', 'hipstr_all.vcf'), 'rb').read()
    rand = bytes(rng.integers(0, 256, size=70000, dtype=np.uint8))
    return {'vcf': vcf, 'fixture': fixture * 3, 'random': rand, 'newlines': b'\n' * 140000, 'ab': b'ab' * 40000,
            'one byte': b'x', 'four': b'abcd', 'long runs': b'A' * 300 + b'C' * 70000 + b'ACGT' * 500,
            'text + noise': vcf[:100000] + rand[:3000] + vcf[:50000], 'every byte': bytes(range(256)) * 300,
            # positions of one token that share a place of a bucket (periods of 8 and 16), and buckets that overflow (three letters)
            'periods': b'ATCGATCC' * 3000 + bytes(rng.integers(65, 68, size=40000, dtype=np.uint8)) + b'0123456789abcdef' * 1500}


@pytest.mark.parametrize('name', sorted(_texts()))
def test_members_inflate_to_the_text(eng, name):
    text = _texts()[name]
    raw = bytes(eng.deflate_bgzf(text))
    ms = _members(raw)
    assert b''.join(t for t, _ in ms) == text
    assert all(len(t) == dm.MEMBER for t, _ in ms[:-1]) and 0 < len(ms[-1][0]) <= dm.MEMBER
    assert gzip.decompress(raw) == text
    # the kernel's model, member by member (the first few of a long text: the model is Python)
    for k, (t, payload) in enumerate(ms[:12]):
        assert payload == dm.deflate_member(t), (name, k)
    if name == 'random':
        assert all(len(p) == len(t) + 5 and p[0] == 1 for t, p in ms)          # stored
    if name in ('vcf', 'fixture', 'newlines', 'ab', 'long runs', 'periods'):
        assert len(raw) < 0.5 * len(text)


def test_every_length_around_a_member(eng):
    base = _synthetic(30, 900, seed=8)
    M = dm.MEMBER
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 258, 259, 260, 1023, M - 1, M, M + 1, 2 * M, 2 * M + 3, 0xff00, 5 * M - 2):
        text = base[:n]
        raw = bytes(eng.deflate_bgzf(text))
        assert b''.join(t for t, _ in _members(raw)) == text, n


def test_the_native_reader_reads_what_the_device_wrote(eng, tmp_path):
    from trtools_amd import bgzf, tabix, vcfnative
    text = _synthetic(500, 1200, seed=21)
    path = str(tmp_path / 'd.vcf.gz')
    with open(path, 'wb') as fh:
        fh.write(eng.deflate_bgzf(bytearray(text)))
        fh.write(bgzf._EOF)
    assert gzip.open(path).read() == text
    r = vcfnative.NativeVCFReader(path)
    assert sum(1 for _ in r) == 500
    r.close()
    idx = tabix.build(path)
    assert sum(idx.bins[k][tabix.META_BIN][1][0] for k in range(len(idx.names))) == 500


def test_repeatable_and_many_members(eng):
    text = _synthetic(700, 3000, seed=2)          # ~25 MB: several rounds of members per launch slot
    a = bytes(eng.deflate_bgzf(text))
    b = bytes(eng.deflate_bgzf(text))
    assert a == b and gzip.decompress(a) == text
    assert len(a) < 0.45 * len(text)


def test_dumpstr_zip_with_the_members_made_on_the_device(tmp_path):
    """dumpSTR --zip three ways -- the host compressor (the default), the device's members (TRK_DEVICE_DEFLATE=1) and no
    --zip at all: the same text in all three, and the writer-built index of the device's file equals a scan of it."""
    import sys
    from test_dumpstr_cli import make_args as dump_args
    from test_vcfnative_hook import _bgzip
    from trtools_amd import tabix
    from trtools_amd.dumpSTR import dumpSTR
    text = _synthetic(1500, 1200, seed=9)
    path = _bgzip(tmp_path, 'in.vcf.gz', text)
    outs = {}
    for tag, env, zipped in (('plain', {}, False), ('host', dict(TRK_DEVICE_DEFLATE='0'), True), ('device', dict(TRK_DEVICE_DEFLATE='1'), True)):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            o = str(tmp_path / tag)
            a = dump_args(o, path, vcftype='hipstr', hipstr_min_call_DP=20, hipstr_max_call_DP=70, hipstr_min_call_Q=0.3,
                          min_locus_callrate=0.2)
            a.zip = zipped
            assert dumpSTR.main(a) == 0
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        raw = gzip.open(o + '.vcf.gz').read() if zipped else open(o + '.vcf', 'rb').read()
        outs[tag] = b'\n'.join(ln for ln in raw.split(b'\n') if not ln.startswith(b'##command-DumpSTR'))
    assert outs['plain'] == outs['host'] == outs['device'] and len(outs['plain']) > (8 << 20)
    dev = str(tmp_path / 'device.vcf.gz')
    sizes = [len(t) for t, _ in _members(open(dev, 'rb').read()[:-28])]
    assert dm.MEMBER in sizes                                   # members of the device's size are in the file
    idx = tabix.TabixIndex.load(dev + '.tbi')
    scan = tabix.build(dev, str(tmp_path / 'scan.tbi'))
    assert (idx.names, idx.bins, idx.linear, idx.meta) == (scan.names, scan.bins, scan.linear, scan.meta)


# ---- random texts (hypothesis): whatever the text, zlib gives it back and the first members equal the model's -----------
from hypothesis import HealthCheck, given, settings, strategies as st

_SCALE = int(os.environ.get('TRK_PROPERTY_SCALE', '0'))       # k: a one-off campaign, k times the examples, fresh seeds


def _random_text(seed, pieces, alphabet, n_words, around):
    """Text made of a small vocabulary of random words (so that matches of every length and distance occur), runs of one
    byte, periodic stretches and noise; cut to a length around a multiple of the member size."""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(0, alphabet, size=int(rng.integers(1, 40)), dtype=np.uint8) + (48 if alphabet <= 64 else 0))
             for _ in range(n_words)]
    out = bytearray()
    want = around * dm.MEMBER + int(rng.integers(-70, 70))
    while len(out) < max(want, 1):
        kind = int(rng.integers(0, pieces))
        if kind == 0:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 700))                    # a run
        elif kind == 1:
            out += words[int(rng.integers(0, n_words))][:int(rng.integers(1, 17))] * int(rng.integers(2, 60))   # a period
        elif kind == 2:
            out += bytes(rng.integers(0, 256, size=int(rng.integers(1, 300)), dtype=np.uint8))       # noise
        else:
            for _ in range(int(rng.integers(1, 200))):
                out += words[int(rng.integers(0, n_words))]
    return bytes(out[:max(want, 1)])


@settings(max_examples=40 * max(_SCALE, 1), deadline=None, derandomize=_SCALE == 0, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2 ** 31), pieces=st.integers(3, 8), alphabet=st.sampled_from([2, 4, 10, 64, 256]),
       n_words=st.integers(1, 400), around=st.integers(0, 5))
def test_random_texts(eng, seed, pieces, alphabet, n_words, around):
    text = _random_text(seed, pieces, alphabet, n_words, around)
    raw = bytes(eng.deflate_bgzf(text))
    ms = _members(raw)                                      # (inflates every member, checks CRC-32 and ISIZE)
    assert b''.join(t for t, _ in ms) == text
    assert all(len(t) == dm.MEMBER for t, _ in ms[:-1])
    for k in {0, len(ms) - 1}:                              # the first and the last (short) member against the model
        assert ms[k][1] == dm.deflate_member(ms[k][0]), (seed, pieces, alphabet, n_words, around, k)
        COUNTS['model'] += 1
    COUNTS['cases'] += 1
    COUNTS['members'] += len(ms)
    COUNTS['bytes'] += len(text)
