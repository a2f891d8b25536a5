"""tests/deflate_model.py (the line-by-line model of k_deflate_bgzf the GPU tests compare the device's bytes with) on the CPU:
whatever it makes must inflate, under zlib, to the text it was given -- the model is only worth comparing with if it is a
DEFLATE producer itself.  (The kernel against the model, and against zlib: tests/test_gpu_deflate.py.)"""
import random
import zlib

import pytest

import deflate_model as dm


def _texts():
    rng = random.Random(11)
    yield b'a'
    yield b'ab' * 3
    yield b'\n' * 700                                   # one byte: a match at distance one, 258 bytes at a time
    yield b'ATCG' * 900 + b'ATCGATCC' * 40              # a period of four and of eight: positions that share a place in a bucket
    yield bytes(rng.randrange(256) for _ in range(6000))            # does not get smaller: stored
    yield bytes(rng.randrange(4) for _ in range(dm.MEMBER))         # a full member
    for alphabet, n in ((2, 300), (16, 5000), (64, 9000)):
        yield bytes(rng.randrange(alphabet) for _ in range(n))
    cells = [b'0|1:%d:0.%d:PASS' % (rng.randrange(60), rng.randrange(100)) for _ in range(900)]
    yield b'chr1\t1000\tSTR_1\tATATAT\tATATATAT\t.\tPASS\tEND=1005\tGT:DP:Q:FILTER\t' + b'\t'.join(cells) + b'\n'
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33):          # short texts: the last positions have no four bytes to hash
        yield bytes(rng.randrange(3) for _ in range(n))


@pytest.mark.parametrize('k', range(len(list(_texts()))))
def test_the_model_makes_deflate_streams(k):
    text = list(_texts())[k][:dm.MEMBER]
    assert zlib.decompress(dm.deflate_member(text), -15) == text


def test_tokens_spell_the_text():
    text = list(_texts())[3]
    out = bytearray()
    for a, b in dm.lz_tokens(text):
        if b:
            assert dm.MIN_MATCH <= a <= dm.MAX_MATCH and 1 <= b <= len(out)
            for _ in range(a):
                out.append(out[-b])
        else:
            out.append(a)
    assert bytes(out) == text


def test_code_lengths_are_complete_and_limited():
    rng = random.Random(5)
    for limit, n in ((15, 286), (15, 30), (7, 19)):
        for _ in range(40):
            freq = [rng.choice((0, 0, 1, 2, 5, 40, 3000)) for _ in range(n)]
            lens = dm.huffman_lengths(freq, limit)
            used = [l for l in lens if l]
            assert all((l > 0) == (f > 0) for l, f in zip(lens, freq))
            assert max(used, default=0) <= limit
            if len(used) > 1:
                assert sum(2.0 ** -l for l in used) == 1.0        # Kraft: a complete code
