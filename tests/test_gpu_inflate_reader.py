"""GPU: the native reader with its BGZF members inflated ON THE DEVICE (NativeVCFReader.device_inflate: trk_inflate_hook /
trk_inflate_text, include/trk.h) against the same reader inflating on the host: per batch the same lines, heads, FORMAT key
indices, genotypes and planes; the text a batch's parse kernel reads in HBM equals the file's text byte for byte; and
statSTR's / dumpSTR's command lines write the same files either way (TRK_DEVICE_INFLATE=0: the host inflates)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import lab_env
from test_vcfnative_hook import _synthetic, _bgzip, _bgzip_members, _crlf_cut_members, _blank_runs_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0, reserve_pair_gb=0)
    yield e
    e.close()


def _read(eng, path, batch_records, device_inflate, pipeline=True):
    from trtools_amd import vcfnative
    r = vcfnative.NativeVCFReader(path, batch_records=batch_records)
    r.select_format('DP')
    r.select_format('Q')
    assert r.device_parse(eng)
    if device_inflate:
        # (two runs in flight -- trk_inflate_hook_async, the default -- or the hook's one synchronous call per run)
        os.environ['TRK_INFLATE_PIPELINE'] = '1' if pipeline else '0'
        try:
            assert r.device_inflate(eng)
        finally:
            os.environ.pop('TRK_INFLATE_PIPELINE', None)
        assert bool(r._inflate_hook.submit) == pipeline
    out = []
    while True:
        rb = r._read_raw_batch(batch_records)
        if rb.n == 0:
            break
        b = rb.b
        heads = []
        for l in range(rb.n):
            f9 = int(b.field_off[l * 10 + 9])
            heads.append((C.string_at(b.text + b.line_off[l], f9), int(b.line_end[l] - b.line_off[l])))
        dev_text = None
        if rb.dev is not None and rb.dev.get('text') is not None:
            dev_text = bytes(rb.dev['text'].get()[:rb.dev['text_nbytes']]) if device_inflate else None
            base = rb.dev['text_base'] if device_inflate else None
        gt, ph, planes = rb.gt.copy(), rb.phased.copy(), {k: v.copy() for k, v in rb.planes.items()}     # (device -> host copies)
        lines = None
        if device_inflate:
            rb.fetch_text()
        lines = [C.string_at(b.text + b.line_off[l], int(b.line_end[l] - b.line_off[l])) for l in range(rb.n)]
        if dev_text is not None:
            # the span of the stream the parse kernel read: from the first line's 16-byte boundary to the last newline
            lo0 = int(b.line_off[0])
            assert dev_text[lo0 - base:lo0 - base + len(lines[0])] == lines[0]
        out.append((heads, lines, gt, ph, planes, rb.locus_ploidy.copy()))
        rb.release_device()
    fallbacks = getattr(r, 'device_fallbacks', 0)
    r.close()
    return out, fallbacks


@pytest.mark.parametrize("case", ["long rows", "short rows", "crlf", "no last newline", "crlf cut by the runs", "runs of blank lines"])
def test_device_inflated_batches_equal_host_inflated_ones(eng, tmp_path, case):
    sizes = ((64, 500000), (37, 3 << 20), (128, None))
    if case == "crlf cut by the runs":
        # every member ends with a line's '\r' and begins with its '\n': each run boundary cuts a pair, and the flag of
        # a newline at offset 0 of a run comes from the run before
        _, parts = _crlf_cut_members(n_rec=900, S=60)
        path = _bgzip_members(tmp_path, 'f.vcf.gz', parts)
        sizes = ((64, 3000), (37, 40000), (128, None))
    elif case == "runs of blank lines":
        # lines of fewer than 16 bytes on average: the line tables are sized from the count (they were
        # total / 16 + 1024 entries, overrun on the device before the host refused the read)
        path = _bgzip(tmp_path, 'f.vcf.gz', _blank_runs_text(n_rec=300, S=200, run=1500000))
    else:
        text = {"long rows": lambda: _synthetic(300, 6000, seed=1), "short rows": lambda: _synthetic(20000, 3, seed=2),
                "crlf": lambda: _synthetic(400, 900, crlf=True, seed=3), "no last newline": lambda: _synthetic(250, 700, last_newline=False, seed=4)}[case]()
        path = _bgzip(tmp_path, 'f.vcf.gz', text)
    from trtools_amd import _lib as L
    for br, read_bytes in sizes:
        opts = {} if read_bytes is None else dict(TRK_VCF_READ_BYTES=read_bytes)
        with L.options(**opts):
            host, _ = _read(eng, path, br, False)
            dev, fb = _read(eng, path, br, True)
            dev1, fb1 = _read(eng, path, br, True, pipeline=False)
        for dev, fb in ((dev, fb), (dev1, fb1)):
            assert len(host) == len(dev) and len(host) > 0 and fb == 0
            for (h1, l1, g1, p1, pl1, lp1), (h2, l2, g2, p2, pl2, lp2) in zip(host, dev):
                assert h1 == h2 and l1 == l2
                assert np.array_equal(g1, g2) and np.array_equal(p1, p2) and np.array_equal(lp1, lp2)
                for k in pl1:
                    assert np.array_equal(pl1[k].view(np.uint32), pl2[k].view(np.uint32)), k
    st = (C.c_uint64 * 5)()
    eng.lib.trk_inflate_stats(eng.ctx, st)
    assert st[0] > 0 and st[1] == 0 and st[2] > 0      # members inflated on the device, none left to zlib


def _args_stat(vcf, out):
    import argparse
    return argparse.Namespace(vcf=vcf, out=out, vcftype='hipstr', samples=None, sample_prefixes=None, plot_afreq=False,
                              region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True, mean=True,
                              mode=True, var=True, numcalled=True, use_length=False, precision=4, nalleles=True,
                              nalleles_thresh=0.01, only_passing=False)


def test_command_lines_with_and_without_the_device_inflate(tmp_path):
    from test_dumpstr_cli import make_args as dump_args
    from trtools_amd.statSTR import statSTR
    from trtools_amd.dumpSTR import dumpSTR
    from trtools_amd import _lib as L
    text = _synthetic(1500, 1200, seed=9)
    # (a record with a triploid call: the device flags it, the host parses that batch -- with the text fetched back)
    lines = text.split(b'\n')
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b'chr1')) + 700
    f = lines[k].split(b'\t')
    f[20] = b'0/1/1:30:0.95'
    lines[k] = b'\t'.join(f)
    path = _bgzip(tmp_path, 'in.vcf.gz', b'\n'.join(lines))
    outs = {}
    for tag, env in (('dev', dict(TRK_DEVICE_INFLATE='1')), ('host', dict(TRK_DEVICE_INFLATE='0'))):
        old = {k_: os.environ.get(k_) for k_ in env}
        os.environ.update(env)
        try:
            with L.options(TRK_VCF_READ_BYTES=2 << 20):
                so = str(tmp_path / ('s_' + tag))
                assert statSTR.main(_args_stat(path, so)) == 0
                assert statSTR.LAST_RUN['device_inflate'] == (tag == 'dev') and statSTR.LAST_RUN['device_parse']
                do = str(tmp_path / ('d_' + tag))
                assert dumpSTR.main(dump_args(do, path, vcftype='hipstr', hipstr_min_call_DP=20, hipstr_max_call_DP=70,
                                              hipstr_min_call_Q=0.3, min_locus_callrate=0.2)) == 0
                assert dumpSTR.LAST_RUN['device_inflate'] == (tag == 'dev')
            outs[tag] = [open(so + '.tab').read()] + ['\n'.join(x for x in open(do + ext).read().split('\n') if not x.startswith('##command-DumpSTR'))
                                                      for ext in ('.vcf', '.samplog.tab', '.loclog.tab')]
        finally:
            for k_, v in old.items():
                if v is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v
    assert outs['dev'] == outs['host']
    assert len(outs['dev'][1]) > 1000000


@pytest.mark.gpu
def test_a_member_nobody_can_inflate_fails_the_read_and_the_next_reader_works(eng, tmp_path):
    """A bgzip'ed file with one member's DEFLATE stream broken (its ISIZE and CRC left alone): the kernel flags the member,
    zlib refuses it too, and the read fails with an error -- with two runs in flight at that moment.  The context's hook is
    usable again afterwards: the next reader on the same engine reads a good file to its end."""
    from trtools_amd import vcfnative, _lib as L
    text = _synthetic(3000, 400, seed=11)
    good = _bgzip(tmp_path, 'good.vcf.gz', text)
    raw = bytearray(open(good, 'rb').read())
    # walk the members; break the payload of one in the middle of the file
    offs, p = [], 0
    while p + 18 <= len(raw):
        bsize = (raw[p + 16] | (raw[p + 17] << 8)) + 1
        offs.append((p, bsize))
        p += bsize
    assert len(offs) > 12
    p, bsize = offs[len(offs) // 2]
    for k in range(p + 18 + 40, p + 18 + 60):
        raw[k] ^= 0x5a
    bad = str(tmp_path / 'bad.vcf.gz')
    open(bad, 'wb').write(bytes(raw))
    with L.options(TRK_VCF_READ_BYTES=200000):       # (runs of a few members: the broken one is met with another run behind it)
        r = vcfnative.NativeVCFReader(bad, batch_records=64)
        r.select_format('DP')
        assert r.device_parse(eng) and r.device_inflate(eng) and r._inflate_hook.submit
        with pytest.raises(ValueError, match='inflate'):
            while True:
                rb = r._read_raw_batch(64)
                if rb.n == 0:
                    break
                rb.release_device()
        r.close()
        out, fb = _read(eng, good, 64, True)
    assert sum(len(h) for h, *_ in out) == 3000 and fb == 0
