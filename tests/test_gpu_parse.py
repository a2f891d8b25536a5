"""GPU parity: trk_parse_samples (the sample columns of a batch parsed on the device, include/trk.h) against the native
reader's arrays for the same records (trk_vcf_read_batch, which the hypothesis fuzz of tests/test_vcfnative_fuzz.py
pins to the Python decoder and through it to the reference's golden outputs).  Bit for bit on every record the device
takes (flag 0); a flagged record is one the host parses -- the flags must stay rare on regular files and must be set
wherever the text leaves the device grammar."""
import os
import sys

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
    e = Engine(0)
    yield e
    e.close()


def device_inputs(rb, keys):
    """Text and offsets of a raw batch as trk_parse_samples wants them, the FORMAT key indices found in Python."""
    import ctypes as C
    b, n = rb.b, rb.n
    lo = np.ctypeslib.as_array(b.line_off, shape=(n,)).astype(np.int64)
    le = np.ctypeslib.as_array(b.line_end, shape=(n,)).astype(np.int64)
    fo = np.ctypeslib.as_array(b.field_off, shape=(n * 10,)).reshape(n, 10).astype(np.int64)
    base = int(lo[0]) & ~15                      # the upload starts on a 16-byte boundary of the reader's buffer
    text = C.string_at(b.text + base, int(le[-1]) + 1 - base)
    gt_idx = np.full(n, -1, np.int8)
    pidx = [np.full(n, -1, np.int8) for _ in keys]
    for i in range(n):
        fmt = text[lo[i] - base + fo[i, 8]: lo[i] - base + fo[i, 9] - 1].decode().split(':') if fo[i, 9] > fo[i, 8] else []
        for k, name in enumerate(fmt[:127]):
            if name == 'GT':
                gt_idx[i] = k
            for j, key in enumerate(keys):
                if name == key and pidx[j][i] < 0:
                    pidx[j][i] = k
    return text, lo + fo[:, 9] - base, le - base, gt_idx, pidx


def compare_file(eng, path, max_ploidy=2, batch_records=64, min_taken=0.0):
    """Every batch of the file through both parsers; returns (records, records the device took)."""
    from trtools_amd import vcfnative, _lib as L
    r = vcfnative.NativeVCFReader(path, batch_records=batch_records, max_ploidy=max_ploidy)
    keys = [k for k, (t, nn) in r.format_types.items() if t in ('Integer', 'Float') and nn == '1' and k != 'GT'][:L.PARSE_MAX_PLANES]
    kinds = ['f' if r.format_types[k][0] == 'Float' else 'i' for k in keys]
    for k in keys:
        r.select_format(k)
    S = len(r.samples)
    n_rec = n_taken = 0
    while True:
        rb = r.read_raw_batch(batch_records)
        if rb.n == 0:
            break
        text, so, le, gi, pidx = device_inputs(rb, keys)
        out = eng.parse_samples(text, so, le, S, rb.gt.shape[2], gi, planes=list(zip(pidx, kinds)), want_phased=True)
        flags = out['flags'].get()
        gt, ph, lp = out['gt'].get(), out['phased'].get(), out['locus_ploidy'].get()
        pl = [p.get() for p in out['planes']]
        take = flags == 0
        assert np.array_equal(gt[take], rb.gt[take]), path
        assert np.array_equal(ph[take], rb.phased[take]), path
        assert np.array_equal(lp[take], rb.locus_ploidy[take]), path
        for k, a in zip(keys, pl):
            want = rb.planes[k][:, :, 0]
            if a.dtype.kind == 'f':
                assert np.array_equal(a[take].view(np.uint32), want[take].view(np.uint32)), (path, k)     # bits: -0.0, NaN
            else:
                assert np.array_equal(a[take], want[take]), (path, k)
        n_rec += rb.n
        n_taken += int(take.sum())
        for a in [out['gt'], out['phased'], out['locus_ploidy'], out['flags']] + out['planes']:
            a.free()
    assert n_taken >= min_taken * n_rec, (path, n_taken, n_rec)
    return n_rec, n_taken


FILES = [os.path.join(GOLDEN, 'dumpstr_synth', f) for f in ('synth_hipstr.vcf', 'synth_gangstr.vcf', 'synth_popstr.vcf', 'synth_eh.vcf')
         if os.path.exists(os.path.join(GOLDEN, 'dumpstr_synth', f))]


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f) for f in FILES])
def test_device_parse_equals_the_native_reader_on_the_fixtures(eng, path):
    n, taken = compare_file(eng, path, min_taken=0.9)
    assert n > 0


def test_device_parse_on_every_spelling(eng, tmp_path):
    """The text of tests/test_vcfnative.py's spelling test: the device takes what its grammar covers -- bit for bit the
    reader's values -- and flags the records that hold anything else."""
    ints = ['7', '007', '-3', '-0', '0', '123456789', '1234567890', '.', '', '+5', '1,2', '12x', '-', '2147483647']
    floats = ['0.97', '1', '.5', '5.', '-.5', '-0', '0.000123', '123456789012345', '1234567890123456', '0.1234567890123456789',
              '1e-3', '1E2', 'inf', '-inf', 'nan', '.', '', '0.5,0.6', '00.25', '-12.75', '3.', '1e', '0x10']
    gts = ['0|1', '1/0', '.', './.', '.|1', '0|', '|1', '', '10|2', '1234|0', '12345|0', '-1|0', '0', '1|1']
    rng = np.random.default_rng(4)
    S = 300
    lines = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 x', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
             '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=GB,Number=1,Type=String,Description="b">',
             '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">', '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">',
             '##FORMAT=<ID=XX,Number=1,Type=String,Description="x">',
             '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    for r in range(60):
        fmt = [['GT', 'GB', 'DP', 'Q', 'XX'], ['GT', 'DP', 'Q'], ['GT', 'Q', 'XX', 'DP'], ['DP', 'GT', 'Q']][r % 4]
        odd = r % 3 == 0                               # two records in three hold regular tokens only
        cols = []
        for s in range(S):
            t = []
            for k in fmt:
                pool = {'GT': gts, 'DP': ints, 'Q': floats, 'GB': ['0|0', '.', 'a'], 'XX': ['x', 'y|z', '.']}[k]
                easy = {'GT': ['0|1', '1|1', '.|.', '12|3'], 'DP': ['12', '30', '.', '-4', '007'],
                        'Q': ['0.9', '1', '0.55', '.5', '123456.789', '-0']}.get(k, pool)
                t.append(str(rng.choice(pool if odd and rng.random() < 0.02 else easy)))
            if rng.random() < 0.1:
                t = t[:int(rng.integers(1, len(t) + 1))]
            tok = ':'.join(t)
            cols.append(tok if tok else '.')
        lines.append('\t'.join(['chr1', str(100 + 50 * r), '.', 'ACAC', 'ACACAC,AC', '.', '.', 'START=%d;END=%d;PERIOD=2' % (100 + 50 * r, 103 + 50 * r),
                                ':'.join(fmt)] + cols))
    path = str(tmp_path / 'spell.vcf')
    open(path, 'w').write('\n'.join(lines) + '\n')
    n, taken = compare_file(eng, path, batch_records=16)
    assert n == 60 and 30 <= taken < 60, (n, taken)


def test_device_parse_long_rows_and_tile_boundaries(eng, tmp_path):
    """Rows of 9000 samples (eight 16 KB tiles and a part), tokens of every length so that starts fall on every byte of a
    chunk, CRLF line ends, a short and a long record; three selected planes."""
    rng = np.random.default_rng(9)
    S = 9000
    hdr = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 x', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
           '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">', '##FORMAT=<ID=ST,Number=1,Type=Integer,Description="s">',
           '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    lines = list(hdr)
    for r in range(12):
        g0 = rng.integers(0, 3, size=S)
        g1 = rng.integers(0, 1200 if r % 4 == 0 else 3, size=S)
        dp = rng.integers(-5, 100000 if r % 3 == 0 else 60, size=S)
        q = rng.random(S)
        cols = []
        for s in range(S):
            if rng.random() < 0.03:
                cols.append('.')
                continue
            cols.append('%d|%d:%d:%s:%d' % (g0[s], g1[s], dp[s], ('%.*f' % (int(rng.integers(0, 9)), q[s])), s % 7))
        lines.append('\t'.join(['chr1', str(100 + 50 * r), '.', 'ACAC', 'ACACAC,AC', '.', '.', 'START=%d;END=%d;PERIOD=2' % (100 + 50 * r, 103 + 50 * r),
                                'GT:DP:Q:ST'] + cols))
    for nl, name in (('\n', 'lf.vcf'), ('\r\n', 'crlf.vcf')):
        path = str(tmp_path / name)
        open(path, 'wb').write((nl.join(lines) + nl).encode())
        n, taken = compare_file(eng, path, batch_records=5, min_taken=1.0)
        assert n == 12
    # a record with a column too few and one with a column too many: flagged, the others untouched
    from trtools_amd import _lib as L
    bad = list(lines)
    bad[len(hdr) + 2] = '\t'.join(bad[len(hdr) + 2].split('\t')[:-1])
    path = str(tmp_path / 'short.vcf')
    open(path, 'w').write('\n'.join(bad) + '\n')
    from trtools_amd import vcfnative
    r = vcfnative.NativeVCFReader(path, batch_records=12, max_ploidy=2)
    r.select_format('DP')
    try:
        rb = r.read_raw_batch(12)
        host_error = None
    except Exception as e:          # the host reader refuses the file (fewer sample columns than the header announces)
        host_error = e
    assert host_error is not None
    # the device says the same about THAT record and parses the rest
    text = ('\n'.join(bad[len(hdr):]) + '\n').encode()
    starts = np.cumsum([0] + [len(x) + 1 for x in bad[len(hdr):]])
    so, le = [], []
    for i, ln in enumerate(bad[len(hdr):]):
        f = ln.split('\t')
        so.append(starts[i] + sum(len(x) + 1 for x in f[:9]))
        le.append(starts[i] + len(ln))
    out = eng.parse_samples(text, np.array(so), np.array(le), S, 2, np.zeros(12, np.int8), planes=[(np.ones(12, np.int8), 'i')])
    flags = out['flags'].get()
    assert flags[2] & L.PARSE_COLUMNS and not flags[[0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]].any()


def _statstr(vcf, out, env):
    import argparse
    from trtools_amd.statSTR import statSTR
    ns = argparse.Namespace(vcf=vcf, out=out, vcftype='hipstr', samples=None, sample_prefixes=None, plot_afreq=False,
                            region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True, mean=True,
                            mode=True, var=True, numcalled=True, use_length=False, precision=4, nalleles=True,
                            nalleles_thresh=0.01, only_passing=False)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        assert statSTR.main(ns) == 0
        return open(out + '.tab').read(), dict(statSTR.LAST_RUN)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_statstr_command_line_with_the_sample_columns_parsed_on_the_device(tmp_path):
    """statSTR's batch pipeline with TRK_DEVICE_PARSE=1 (reader stops at the FORMAT keys, trk_parse_samples builds the
    tensor in HBM): the same table, byte for byte -- on a regular file (no batch falls back) and on one with a triploid
    call and an exponent-free but odd token in every third record (those batches are parsed by the host after all)."""
    src = os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf')
    a, ra = _statstr(src, str(tmp_path / 'host'), {'TRK_DEVICE_PARSE': '0'})
    b, rb = _statstr(src, str(tmp_path / 'dev'), {'TRK_DEVICE_PARSE': '1'})
    assert a == b and len(a.splitlines()) > 20
    assert rb.get('device_parse') is True and not ra.get('device_parse')
    odd = str(tmp_path / 'odd.vcf')
    with open(src) as fin, open(odd, 'w') as fout:
        k = 0
        for line in fin:
            if not line.startswith('#'):
                k += 1
                f = line.rstrip('\n').split('\t')
                if k % 3 == 0:
                    t = f[12].split(':')
                    t[0] = '0|1|1' if k % 6 == 0 else '+1|0'
                    f[12] = ':'.join(t)
                line = '\t'.join(f) + '\n'
            fout.write(line)
    try:
        a, _ = _statstr(odd, str(tmp_path / 'host2'), {'TRK_DEVICE_PARSE': '0'})
        err_a = None
    except Exception as e:
        a, err_a = None, (type(e), str(e))
    try:
        b, _ = _statstr(odd, str(tmp_path / 'dev2'), {'TRK_DEVICE_PARSE': '1'})
        err_b = None
    except Exception as e:
        b, err_b = None, (type(e), str(e))
    assert err_a == err_b and a == b


def test_dumpstr_command_line_with_the_sample_columns_parsed_on_the_device(tmp_path):
    """dumpSTR's batch pipeline with TRK_DEVICE_PARSE=1 on the HipSTR fixture (three and five call filters: two and
    four scalar planes parsed on the device, handed to the call-filter pass without an upload, host copies back by DMA for
    the record writer): output VCF and both logs byte for byte the host parse's."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_dumpstr_cli import make_args as dump_args
    from trtools_amd.dumpSTR import dumpSTR
    src = os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf')
    sets = [dict(hipstr_min_call_DP=20, hipstr_max_call_DP=50, hipstr_min_call_Q=0.9, min_locus_callrate=0.2),
            dict(hipstr_min_call_DP=10, hipstr_max_call_DP=60, hipstr_min_call_Q=0.8, hipstr_max_call_flank_indel=0.15,
                 hipstr_max_call_stutter=0.15, min_locus_hwep=0.0001)]
    for i, kw in enumerate(sets):
        outs = []
        for dev in ('0', '1'):
            os.environ['TRK_DEVICE_PARSE'] = dev
            try:
                out = str(tmp_path / ('d%d_%s' % (i, dev)))
                assert dumpSTR.main(dump_args(out, src, vcftype='hipstr', **kw)) == 0
                assert dumpSTR.LAST_RUN['path'] == 'batch' and bool(dumpSTR.LAST_RUN.get('device_parse')) == (dev == '1')
                outs.append([open(out + suf).read() for suf in ('.vcf', '.samplog.tab', '.loclog.tab')])
            finally:
                os.environ.pop('TRK_DEVICE_PARSE', None)
        a, b = outs
        a[0] = '\n'.join(x for x in a[0].split('\n') if not x.startswith('##command-DumpSTR'))
        b[0] = '\n'.join(x for x in b[0].split('\n') if not x.startswith('##command-DumpSTR'))
        assert a == b and a[0].count('\n') > 30


@pytest.mark.parametrize('seed', [5, 6, 7, 8])
def test_dumpstr_sample_columns_written_on_the_device(tmp_path, seed):
    """trk_format_samples behind dumpSTR's command line (on by default with the device parse; TRK_DEVICE_FORMAT=0: the host writer): on the
    HipSTR fixture and on its rewritten-text variants (numbers spelled every way, tokens that stop early, a lone '.', a
    field too many -- tests/test_batch_pipelines.py's generator) the output VCF is byte for byte the host writer's; the
    device takes the regular records and leaves the others."""
    from test_dumpstr_cli import make_args as dump_args
    from test_batch_pipelines import _mutated_hipstr
    from trtools_amd import vcfnative
    from trtools_amd.dumpSTR import dumpSTR
    src = os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf') if seed == 5 else _mutated_hipstr(tmp_path, seed, scalar=True)
    if seed != 5:
        # (exponents are outside the device PARSE grammar and would send every batch to the host reader: spelled out
        # here, as numbers that parse on the device and are not canonical for the writer)
        text = open(src).read().replace(':1e-3', ':0.001000').replace(':1E2', ':100.0')
        open(src, 'w').write(text)
    kw = dict(hipstr_min_call_DP=20, hipstr_max_call_DP=50, hipstr_min_call_Q=0.9, min_locus_callrate=0.2)
    outs, took, emitted = [], [], []
    # (host writer; device columns put in place in the output block -- whole-record emit, the default; device columns
    # downloaded and gathered behind their heads by the writer, round 4's form)
    for dev, emit in (('0', '1'), ('1', '1'), ('1', '0')):
        os.environ['TRK_DEVICE_FORMAT'] = dev
        os.environ['TRK_FMT_EMIT'] = emit
        before = dict(vcfnative.DEVICE_FORMAT)
        try:
            out = str(tmp_path / ('f%s%s' % (dev, emit)))
            assert dumpSTR.main(dump_args(out, src, vcftype='hipstr', **kw)) == 0
            assert dumpSTR.LAST_RUN['path'] == 'batch'
            outs.append('\n'.join(x for x in open(out + '.vcf').read().split('\n') if not x.startswith('##command-DumpSTR')))
            took.append(vcfnative.DEVICE_FORMAT['records'] - before['records'])
            emitted.append(vcfnative.DEVICE_FORMAT.get('emitted', 0) - before.get('emitted', 0))
        finally:
            os.environ.pop('TRK_DEVICE_FORMAT', None)
            os.environ.pop('TRK_FMT_EMIT', None)
    assert outs[2] == outs[1] and took[2] == took[1] and emitted == [0, took[1], 0], (took, emitted)
    if outs[0] != outs[1]:
        la, lb = outs[0].split('\n'), outs[1].split('\n')
        i = next(i for i, (p, q) in enumerate(zip(la, lb)) if p != q)
        fa, fb = la[i].split('\t'), lb[i].split('\t')
        j = next((j for j, (p, q) in enumerate(zip(fa, fb)) if p != q), -1)
        raise AssertionError("line %d column %d: host %r device %r" % (i, j, fa[j][:80] if j >= 0 else len(fa), fb[j][:80] if j >= 0 else len(fb)))
    assert took[0] == 0 and took[1] > (20 if seed == 5 else 0), took


def test_dumpstr_device_format_long_rows_and_tile_boundaries(tmp_path):
    """k_format_samples beyond one tile: rows of 3000 samples (~45 KB of sample text: six 8 KB tiles), tokens of every
    length so that starts and ends fall on every byte of a chunk and across tile ends, calls every filter fires on, a
    record whose tokens are longer than the text staged beyond a tile (the host writer's) and CRLF line ends.  dumpSTR's output with the device writer is byte for byte the host writer's."""
    from test_dumpstr_cli import make_args as dump_args
    from trtools_amd import vcfnative
    from trtools_amd.dumpSTR import dumpSTR
    rng = np.random.default_rng(31)
    S = 3000
    hdr = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 x', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
           '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">', '##FORMAT=<ID=TAG,Number=1,Type=String,Description="t">',
           '##contig=<ID=chr1,length=100000>',
           '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    # (depths of seven digits and more among the values: the writer prints those with an exponent.  No Q below 1e-4: spelled
    # without an exponent such a token is not canonical -- decode -> format would rewrite it -- and the record is the host's)
    qs = ['0.99', '1', '0.5', '0.93', '0.912345', '0.0001', '0.85', '.', '0.001', '0.00012']
    for nl, name in (('\n', 'lf'), ('\r\n', 'crlf')):
        lines = list(hdr)
        for r in range(14):
            cols = []
            for s in range(S):
                u = rng.random()
                if u < 0.03:
                    cols.append('.')
                    continue
                gt = '%d|%d' % (rng.integers(0, 3), rng.integers(0, 3))
                tag = 'x' * (400 if r == 9 else int(rng.integers(1, 24)))
                dp = int(rng.integers(0, 90)) if rng.random() < 0.97 else int(rng.choice([1234567, 20000000, 999999, 1000000, 214748364]))
                t = [gt, str(dp), qs[int(rng.integers(0, len(qs)))], tag]
                cols.append(':'.join(t[:int(rng.integers(2, 5))] if u < 0.1 else t))
            lines.append('\t'.join(['chr1', str(100 + 50 * r), '.', 'ACAC', 'ACACAC,AC', '.', '.',
                                    'START=%d;END=%d;PERIOD=2' % (100 + 50 * r, 103 + 50 * r), 'GT:DP:Q:TAG'] + cols))
        src = str(tmp_path / (name + '.vcf'))
        open(src, 'wb').write((nl.join(lines) + nl).encode())
        kw = dict(hipstr_min_call_DP=20, hipstr_max_call_DP=70, hipstr_min_call_Q=0.9)
        outs, took = [], []
        for dev in ('0', '1'):
            os.environ['TRK_DEVICE_FORMAT'] = dev
            before = dict(vcfnative.DEVICE_FORMAT)
            try:
                out = str(tmp_path / ('%s_f%s' % (name, dev)))
                assert dumpSTR.main(dump_args(out, src, vcftype='hipstr', **kw)) == 0
                assert dumpSTR.LAST_RUN['path'] == 'batch'
                outs.append([x for x in open(out + '.vcf', 'rb').read().split(b'\n') if not x.startswith(b'##command-DumpSTR')])
                took.append((vcfnative.DEVICE_FORMAT['records'] - before['records'],
                             vcfnative.DEVICE_FORMAT['left_to_host'] - before['left_to_host']))
            finally:
                os.environ.pop('TRK_DEVICE_FORMAT', None)
        assert len(outs[0]) == len(outs[1])
        for i, (p, q) in enumerate(zip(outs[0], outs[1])):
            if p != q:
                fa, fb = p.split(b'\t'), q.split(b'\t')
                j = next((j for j, (x, y) in enumerate(zip(fa, fb)) if x != y), -1)
                raise AssertionError("%s line %d column %d: host %r device %r" % (name, i, j, fa[j][:80], fb[j][:80]))
        if name == 'lf':
            assert took[0] == (0, 0) and took[1][0] >= 12 and took[1][1] >= 1, took     # (record 9: tokens beyond the staged text)


def test_device_parse_tokens_longer_than_the_staged_text(eng, tmp_path):
    """k_parse_samples walks its tokens in an LDS copy of the tile that reaches 256 bytes beyond it.  FORMAT GT:TAG:DP:Q with
    string fields of 20 ... 900 bytes IN FRONT of the numbers: a token that starts near a tile's end and needs bytes beyond
    the staged text flags its record (the host's); every record the device takes equals the host reader's arrays, and the
    records with short strings are all taken."""
    rng = np.random.default_rng(123)
    S = 1500
    hdr = ['##fileformat=VCFv4.2', '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
           '##FORMAT=<ID=TAG,Number=1,Type=String,Description="t">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
           '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">',
           '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    lines = list(hdr)
    long_rec = []
    for r in range(10):
        big = r % 2 == 1
        long_rec.append(big)
        cols = []
        for s in range(S):
            ln = int(rng.integers(300, 900)) if big else int(rng.integers(1, 20))
            cols.append('%d/%d:%s:%d:%.3f' % (rng.integers(0, 3), rng.integers(0, 3), 'y' * ln, rng.integers(0, 99), rng.random()))
        lines.append('\t'.join(['chr1', str(100 + 10 * r), '.', 'ACAC', 'AC,ACACAC', '.', '.', '.', 'GT:TAG:DP:Q'] + cols))
    path = str(tmp_path / 'longtags.vcf')
    open(path, 'w').write('\n'.join(lines) + '\n')
    from trtools_amd import vcfnative
    r = vcfnative.NativeVCFReader(path, batch_records=10, max_ploidy=2)
    for k in ('DP', 'Q'):
        r.select_format(k)
    rb = r.read_raw_batch(10)
    text, so, le, gi, pidx = device_inputs(rb, ['DP', 'Q'])
    out = eng.parse_samples(text, so, le, S, 2, gi, planes=list(zip(pidx, ['i', 'f'])), want_phased=True)
    flags = out['flags'].get()
    take = flags == 0
    assert take[[i for i, b in enumerate(long_rec) if not b]].all()            # short strings: all the device's
    assert not take[[i for i, b in enumerate(long_rec) if b]].any()            # 1500 tokens of 300+ bytes: some tile end hits one
    assert np.array_equal(out['gt'].get()[take], rb.gt[take])
    assert np.array_equal(out['planes'][0].get()[take], rb.planes['DP'][:, :, 0][take])
    assert np.array_equal(out['planes'][1].get()[take].view(np.uint32), rb.planes['Q'][:, :, 0][take].view(np.uint32))
    # and through the reader's own device mode: the batch falls back to the host parser, same arrays
    n, taken = compare_file(eng, path, batch_records=10)
    assert n == 10 and taken == 5
