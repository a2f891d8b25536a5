"""Contiguous shards of one VCF for N readers (trk_vcf_shard; SURVEY 8(e): contiguous locus ranges per rank): the
ranks' records, concatenated in rank order, are the records of the file in file order -- bgzipped fixtures of the
reference, a multi-block synthetic file with lines that straddle block boundaries, plain text -- and every rank
inflates its own share of the blocks plus at most two more."""
import gzip
import os

import numpy as np
import pytest

from helpers import GOLDEN

D = os.path.join(GOLDEN, 'data')
FILES = [os.path.join(D, 'many_samples.vcf.gz'), os.path.join(D, 'dumpSTR', 'trio_chr21_hipstr.sorted.vcf.gz'),
         os.path.join(D, 'dumpSTR', 'trio_chr21_gangstr.sorted.vcf.gz'),
         os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf')]


def _records(path, rank=0, world=1, batch=257):
    from trtools_amd.vcfnative import NativeVCFReader
    r = NativeVCFReader(path)
    assert r.shard(rank, world)
    out, gts = [], []
    while True:
        rb = r.read_raw_batch(batch)
        if rb.n == 0:
            break
        for l in range(rb.n):
            out.append('\t'.join(rb.head_fields(l)[:5]))
        gts.append(np.array(rb.gt, copy=True))
    c = r.counters()
    r.close()
    return out, gts, c


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(p) for p in FILES])
@pytest.mark.parametrize('world', [2, 3, 8])
def test_shards_concatenate_to_the_file(path, world):
    whole, gt_whole, _ = _records(path)
    got, gts, inflated = [], [], []
    for rank in range(world):
        recs, g, c = _records(path, rank, world)
        got += recs
        gts += g
        inflated.append(c)
    assert got == whole
    assert np.array_equal(np.concatenate(gts), np.concatenate(gt_whole))
    if path.endswith('.gz'):
        total = sum(c['blocks'] for c in inflated)
        n_blocks = _records(path, 0, 1)[2]['blocks'] or total
        for c in inflated:      # own share + the block(s) in which the last line ends
            assert c['blocks'] <= -(-n_blocks // world) + 3, (inflated, n_blocks)


def test_lines_that_straddle_blocks_and_ranks(tmp_path):
    """A bgzip file written in tiny blocks (lines longer than a block, block ends on and off line ends): for every
    world size up to the number of blocks each record is read exactly once, in order."""
    import struct
    import zlib
    src = os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf')
    text = open(src, 'rb').read()
    lines = text.split(b'\n')
    # 40 records are enough; make some lines end exactly at a block end
    head = [l for l in lines if l.startswith(b'#')]
    body = [l for l in lines if l and not l.startswith(b'#')][:40]
    data = b'\n'.join(head + body) + b'\n'
    path = str(tmp_path / 'tiny_blocks.vcf.gz')
    rng = np.random.default_rng(5)
    with open(path, 'wb') as fh:
        p = 0
        while p < len(data):
            n = int(rng.integers(200, 3000))
            nl = data.find(b'\n', p)
            if rng.random() < 0.3 and nl >= 0 and nl + 1 - p <= 60000:
                n = nl + 1 - p                  # this block ends exactly at a line end
            chunk = data[p:p + n]
            p += len(chunk)
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            bsize = len(comp) + 25
            fh.write(b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', bsize) + comp +
                     struct.pack('<II', zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        fh.write(bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000'))
    assert gzip.open(path, 'rb').read() == data
    whole, gt_whole, c = _records(path)
    assert len(whole) == 40
    for world in (2, 3, 5, 7, 16, 64):
        got, gts = [], []
        for rank in range(world):
            recs, g, _ = _records(path, rank, world, batch=7)
            got += recs
            gts += g
        assert got == whole, world
        assert np.array_equal(np.concatenate(gts), np.concatenate(gt_whole))


def test_plain_gzip_cannot_be_cut(tmp_path):
    from trtools_amd.vcfnative import NativeVCFReader
    src = os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf')
    path = str(tmp_path / 'plain.vcf.gz')
    with gzip.open(path, 'wb') as fh:
        fh.write(open(src, 'rb').read())
    r = NativeVCFReader(path)
    assert r.shard(1, 2) is False
    n = 0
    while True:
        rb = r.read_raw_batch(100)
        if rb.n == 0:
            break
        n += rb.n
    r.close()
    assert n == sum(1 for l in open(src) if l.strip() and not l.startswith('#'))
