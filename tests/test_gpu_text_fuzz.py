"""The DEVICE text kernels against the definition (hypothesis fuzz).

tests/test_gpu_parse.py pins k_parse_samples / k_format_samples to the native host reader / writer, which the CPU
fuzz pins to the Python decoder (vcfio.py), which the reference's golden outputs pin: three hops, and the C++ scanner
and the kernels were written from the same "one forward scan" rules.  Here the device is compared with the Python
side DIRECTLY:

* ``trk_parse_samples`` on random record text -- every spelling tests/test_vcfnative_fuzz.py renders (mixed ploidy,
  phasing, partial and missing calls, '.', ragged vectors, exponents, trailing fields dropped, CRLF) plus the spelling
  pools of tests/test_gpu_parse.py -- against ``vcfio.VCFReader``'s arrays.  A record the device takes (flag 0) must
  hold vcfio's genotypes, phasing and values bit for bit (floats by their bits); a record it flags is parsed by the
  host reader in the product, so THAT result must equal vcfio's too, and the flagged records are counted: text made
  of in-grammar spellings only must never be flagged.
* ``trk_format_samples`` behind dumpSTR's command line (device parse + device format, the defaults) against the
  per-record Python loop (vcfio reader, vcfio writer: the path the reference's golden VCFs pin) on random HipSTR-shape
  files with filters that fire: VCF, sample log and locus log byte for byte.

TRK_PROPERTY_SCALE=k: k times the examples with fresh seeds (a one-off campaign; profiles/r05_text_fuzz_campaign.txt)."""
import os
import sys

import numpy as np
import pytest
from hypothesis import example, HealthCheck, given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import lab_env

pytestmark = pytest.mark.gpu
_SCALE = int(os.environ.get('TRK_PROPERTY_SCALE', '0'))
COUNTS = {'records': 0, 'taken': 0, 'flagged': 0, 'cases': 0}


def _cfg(n):
    return settings(max_examples=n * max(_SCALE, 1), deadline=None, derandomize=_SCALE == 0, database=None,
                    suppress_health_check=list(HealthCheck))


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    if COUNTS['cases']:
        print("\n[device text fuzz] %(cases)d cases, %(records)d records: %(taken)d taken by the device, %(flagged)d flagged "
              "(host reader checked against vcfio on those)" % COUNTS)


HDR = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 fuzz',
       '##INFO=<ID=START,Number=1,Type=Integer,Description="s">', '##INFO=<ID=END,Number=1,Type=Integer,Description="e">',
       '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
       '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
       '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">', '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="a">',
       '##FORMAT=<ID=ST,Number=1,Type=Integer,Description="s">', '##FORMAT=<ID=GB,Number=1,Type=String,Description="b">',
       '##FORMAT=<ID=W,Number=1,Type=Float,Description="w">']

# spellings: `easy` pools are inside the device grammar (plain digits, a sign, decimals with a digit on one side at
# least); `hard` pools hold everything the host reader takes (and a few things nobody takes as a number)
GT_EASY = ['0|1', '1/0', '.', './.', '.|1', '1|.', '0/0', '10|2', '123|0', '7', '1|1']
GT_HARD = GT_EASY + ['1234|0', '12345|0', '0/1/2', '1|2|3', '.|.|.', '0/1|1']
INT_EASY = ['7', '30', '-3', '0', '-0', '.', '123456789', '007', '-999999999']
# (tokens the Python decoder refuses outright -- '', '12x', '1e3' in an Integer field, integers beyond int32 -- are no
# use here: the definition says nothing about them)
INT_HARD = INT_EASY + ['2147483647', '-2147483647', '1234567890', '+5', '1,2', '-2147483648', '+0', '0000000012']
FLT_EASY = ['0.97', '1', '.5', '5.', '-.5', '-0', '0.000123', '123456.789', '00.25', '-12.75', '3.', '.', '0', '0.1',
            '16777217', '0.3333333', '1234567.125', '123456789012345']
FLT_HARD = FLT_EASY + ['0.30000000000000004', '1234567890123456', '0.1234567890123456789', '1e-3', '1E2', '2.5e+4', 'inf', '-inf',
                       'nan', 'NaN', '0.5,0.6', '+.5', '4e400', '1e-400']


def _render(rng, n_rec, S, hard_p, crlf, ploidy_max):
    """Record lines (no header) and whether every token came from the easy pools."""
    lines, easy_only = [], True
    pos = 100
    for r in range(n_rec):
        pos += int(rng.integers(1, 500))
        keys_all = ['DP', 'Q', 'AD', 'ST', 'GB', 'W']
        keys = [k for k in keys_all if rng.random() < 0.65]
        rng.shuffle(keys)
        gt_at = int(rng.integers(0, len(keys) + 1)) if rng.random() < 0.2 else 0     # GT is not always the first key
        fmt = keys[:gt_at] + ['GT'] + keys[gt_at:]
        if rng.random() < 0.05:
            fmt = [k for k in fmt if k != 'GT'] or ['DP']                              # a record without genotypes
        cols = []
        for s in range(S):
            toks = []
            for k in fmt:
                hard = rng.random() < hard_p
                if k == 'GT':
                    if rng.random() < 0.7:
                        p = int(rng.integers(1, ploidy_max + 1))
                        t = ('|' if rng.random() < 0.5 else '/').join(
                            '.' if rng.random() < 0.12 else str(int(rng.integers(0, 14))) for _ in range(p))
                    else:
                        t = str(rng.choice(GT_HARD if hard else GT_EASY))
                        easy_only &= not hard
                elif k in ('DP', 'ST'):
                    t = str(rng.choice(INT_HARD if hard else INT_EASY)) if rng.random() < 0.5 else str(int(rng.integers(-40, 90000)))
                    easy_only &= not hard
                elif k in ('Q', 'W'):
                    if rng.random() < 0.5:
                        t = str(rng.choice(FLT_HARD if hard else FLT_EASY))
                        easy_only &= not hard
                    else:
                        t = '%.*f' % (int(rng.integers(0, 9)), float(rng.random()) * 10 ** int(rng.integers(-3, 5)))
                elif k == 'AD':
                    t = ','.join(str(int(rng.integers(0, 50))) for _ in range(int(rng.integers(1, 4))))
                else:
                    t = str(rng.choice(['0|0', '.', '-2|4', 'x', 'a;b|c', 'long_' * int(rng.integers(1, 30))]))
                    easy_only &= len(t) <= 40      # (a token whose needed fields lie beyond the text staged behind a tile is the host's)
                toks.append(t)
            if rng.random() < 0.12 and len(toks) > 1:
                toks = toks[:int(rng.integers(1, len(toks)))]        # trailing fields dropped
            tok = ':'.join(toks)
            cols.append(tok if tok else '.')
        lines.append('\t'.join(['chr1', str(pos), '.', 'ACAC', 'ACACAC,AC,ACACACAC', '.', '.', 'START=%d;END=%d;PERIOD=2' % (pos, pos + 3),
                                ':'.join(fmt)] + cols))
    return lines, easy_only


def _device_parse(eng, rec_lines, nl, S, P, keys, kinds):
    """trk_parse_samples over the records' text: offsets found in Python (nothing of the native reader involved)."""
    text = (nl.join(rec_lines) + nl).encode()
    so, le, gi, pidx = [], [], [], [[] for _ in keys]
    at = 0
    for ln in rec_lines:
        f = ln.split('\t')
        so.append(at + sum(len(x.encode()) + 1 for x in f[:9]))
        le.append(at + len(ln.encode()))                               # the line's end: its '\r' or '\n'
        fmt = f[8].split(':')
        gi.append(fmt.index('GT') if 'GT' in fmt else -1)
        for j, k in enumerate(keys):
            pidx[j].append(fmt.index(k) if k in fmt else -1)
        at += len(ln.encode()) + len(nl)
    return eng.parse_samples(text, np.array(so, np.int64), np.array(le, np.int64), S, P, np.array(gi, np.int8),
                             planes=[(np.array(p, np.int8), kd) for p, kd in zip(pidx, kinds)], want_phased=True)


def _vcfio_arrays(x, S, P, keys, kinds):
    """What the Python decoder says about one record, laid out as the device lays it out."""
    gt = np.full((S, P), -2, np.int16)
    ph = np.zeros(S, np.uint8)
    if x.genotype is not None:
        a = x.genotype.array()
        w = a.shape[1] - 1
        if w > P:
            return None                                         # more alleles than the tensor holds: the reader's error
        gt[:, :w] = a[:, :w]
        ph = a[:, -1].astype(np.uint8)
    else:
        gt[:] = -1 if False else gt                               # (no GT key: see below)
    planes = []
    for k, kd in zip(keys, kinds):
        if k in x.FORMAT:
            v = x.format(k)[:, 0]
            planes.append(v.astype(np.float32) if kd == 'f' else v.astype(np.int32))
        else:
            planes.append(np.full(S, np.nan, np.float32) if kd == 'f' else np.full(S, -2147483648, np.int32))
    return gt, ph, planes


@_cfg(2000)
@given(seed=st.integers(0, 2**31 - 1), n_rec=st.integers(1, 10), S=st.integers(1, 48), P=st.integers(1, 3),
       hard=st.sampled_from([0.0, 0.0, 0.03, 0.3]), crlf=st.booleans())
@example(seed=9611089, n_rec=2, S=5, P=2, hard=0.0, crlf=False)      # (round-6 campaign: a FORMAT column with none of the asked keys)
def test_device_parse_against_the_python_decoder(eng, tmp_path_factory, seed, n_rec, S, P, hard, crlf):
    from trtools_amd import vcfio, vcfnative, _lib as L
    rng = np.random.default_rng(seed)
    nl = '\r\n' if crlf else '\n'
    rec_lines, easy_only = _render(rng, n_rec, S, hard, crlf, P)
    keys, kinds = ['DP', 'Q', 'ST', 'W'], ['i', 'f', 'i', 'f']
    d = tmp_path_factory.mktemp('dfz')
    path = str(d / 'f.vcf')
    head = HDR + ['#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    with open(path, 'wb') as fh:
        fh.write((nl.join(head + rec_lines) + nl).encode())
    try:
        py = list(vcfio.VCFReader(path))
    except (ValueError, OverflowError):
        return                      # (text the Python decoder refuses outright: the definition says nothing)
    assert len(py) == n_rec
    out = _device_parse(eng, rec_lines, nl, S, P, keys, kinds)
    flags = out['flags'].get()
    gt, ph, lp = out['gt'].get(), out['phased'].get(), out['locus_ploidy'].get()
    pl = [p.get() for p in out['planes']]
    for a in [out['gt'], out['phased'], out['locus_ploidy'], out['flags']] + out['planes']:
        a.free()
    host = None
    for i, x in enumerate(py):
        try:
            want = _vcfio_arrays(x, S, P, keys, kinds)
        except (ValueError, OverflowError):
            # a token the Python decoder refuses outright ('' or '1e3' in an Integer field): the device must not take it
            assert flags[i] != 0, ("the device took a record the Python decoder refuses", seed, i, rec_lines[i][:300])
            COUNTS['records'] += 1
            COUNTS['flagged'] += 1
            continue
        COUNTS['records'] += 1
        if flags[i] == 0:
            COUNTS['taken'] += 1
            assert want is not None, "the device took a record with more alleles per call than the tensor holds"
            wg, wp, wpl = want
            if x.genotype is None:
                continue                      # (no GT key: the genotype rows are whatever the caller pre-set; planes below)
            assert np.array_equal(gt[i], wg), (seed, i, 'gt', [(c, a_.tolist(), b_.tolist()) for c, a_, b_ in zip(rec_lines[i].split('\t')[9:], gt[i], wg) if not np.array_equal(a_, b_)][:5], rec_lines[i].split('\t')[8])
            assert np.array_equal(ph[i], wp), (seed, i, 'phased', [(c, int(a_), int(b_)) for c, a_, b_ in zip(rec_lines[i].split('\t')[9:], ph[i], wp) if a_ != b_][:5], rec_lines[i].split('\t')[8])
            assert lp[i] == max(1, x.ploidy) or x.ploidy == 0, (seed, i, lp[i], x.ploidy)
            for k, kd, a, w in zip(keys, kinds, pl, wpl):
                if kd == 'f':
                    assert np.array_equal(a[i].view(np.uint32), w.view(np.uint32)), (seed, i, k, rec_lines[i][:200])
                else:
                    # (cyvcf2's missing / end-of-vector markers are both "not a value" to every caller)
                    miss_a, miss_w = a[i] <= -2147483647, w <= -2147483647
                    assert np.array_equal(miss_a, miss_w) and np.array_equal(a[i][~miss_a], w[~miss_w]), (
                        seed, i, k, [(c, int(a_), int(b_)) for c, a_, b_ in zip(rec_lines[i].split('\t')[9:], a[i], w) if a_ != b_][:5], rec_lines[i].split('\t')[8])
        else:
            COUNTS['flagged'] += 1
            if want is None:                  # a call with more alleles than the tensor holds
                assert flags[i] & (L.PARSE_PLOIDY | L.PARSE_HOST), (seed, i, int(flags[i]))
                continue
            # (a record whose FORMAT holds neither GT nor any of the asked keys is the host's by the kernel's own rule -- nothing
            # for the device to parse, "the host has the last word": trk_parse.hip; met by the round-6 campaign, seed 9611089)
            nothing_asked = 'GT' not in x.FORMAT and not any(k in x.FORMAT for k in keys)
            assert not easy_only or nothing_asked, ("a record of in-grammar spellings was flagged", seed, i, int(flags[i]), rec_lines[i][:300])
            # the product parses this record on the host: that result must be vcfio's
            if host is None:
                try:
                    r = vcfnative.NativeVCFReader(path, batch_records=n_rec, max_ploidy=P)
                    for k in keys:
                        r.select_format(k)
                    host = list(r)
                except Exception as e:                      # (a call wider than the tensor somewhere in the file)
                    host = e
            if isinstance(host, Exception):
                continue
            y = host[i]
            if x.genotype is not None:
                assert np.array_equal(x.genotype.array(), y.genotype.array()), (seed, i)
            for k in keys:
                if k in x.FORMAT:
                    xa, ya = x.format(k)[:, :1], y.format(k)[:, :1]
                    same = (np.array_equal(xa.view(np.uint32), ya.view(np.uint32)) if xa.dtype.kind == 'f' else
                            np.array_equal(xa, ya))
                    assert same, (seed, i, k)
    COUNTS['cases'] += 1
    os.remove(path)


def _hipstr_file(rng, path, n_rec, S, hard_p, crlf):
    """A HipSTR-shape file dumpSTR runs on: GT first, DP and Q among the keys, depths and qualities on both sides of the
    thresholds, spellings the writers must keep or re-serialise."""
    nl = '\r\n' if crlf else '\n'
    head = HDR + ['##contig=<ID=chr1,length=10000000>',
                  '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    qs_easy = ['0.99', '1', '0.5', '0.93', '0.912345', '0.85', '.', '0.001', '0.9', '0.90', '1.0', '0.899999', '.95']
    qs_hard = qs_easy + ['1e-3', '9E-1', '0.0000001', '00.95', '0.950000000000000001', 'nan']
    lines = []
    pos = 1000
    for r in range(n_rec):
        pos += int(rng.integers(10, 400))
        extra = [k for k in ('ST', 'GB', 'W') if rng.random() < 0.4]
        fmt = ['GT', 'DP', 'Q'] + extra
        if rng.random() < 0.15:
            fmt = ['GT', 'Q'] + extra + ['DP']
        cols = []
        for s in range(S):
            toks = []
            for k in fmt:
                hard = rng.random() < hard_p
                if k == 'GT':
                    t = str(rng.choice(['0|1', '1|1', '0|0', '.', '.|.', '2|1', '0/1', '1|.'] + (['0/1/1', '1'] if hard else [])))
                elif k == 'DP':
                    t = str(rng.choice(['15', '25', '30', '45', '55', '70', '.', '007', '1234567', '20', '50'] + (['+25', '0000030'] if hard else [])))
                elif k == 'Q':
                    t = str(rng.choice(qs_hard if hard else qs_easy))
                elif k == 'ST':
                    t = str(int(rng.integers(0, 9)))
                elif k == 'W':
                    t = str(rng.choice(['0.5', '1', '.', '2.50', '1e2' if hard else '3']))
                else:
                    t = str(rng.choice(['0|0', '.', '-2|4', 'x' * int(rng.integers(1, 40))]))
                toks.append(t)
            if rng.random() < 0.08:
                toks = toks[:int(rng.integers(1, len(toks) + 1))]
            cols.append(':'.join(toks) or '.')
        lines.append('\t'.join(['chr1', str(pos), 'id%d' % r, 'ACAC', 'ACACAC,AC', '.', '.', 'START=%d;END=%d;PERIOD=2' % (pos, pos + 3),
                                ':'.join(fmt)] + cols))
    with open(path, 'wb') as fh:
        fh.write((nl.join(head + lines) + nl).encode())


@_cfg(150)
@given(seed=st.integers(0, 2**31 - 1), n_rec=st.integers(1, 30), S=st.integers(1, 60), hard=st.sampled_from([0.0, 0.0, 0.02, 0.2]),
       crlf=st.booleans())
def test_device_format_against_the_per_record_python_writer(eng, tmp_path_factory, seed, n_rec, S, hard, crlf):
    from test_dumpstr_cli import make_args as dump_args
    from trtools_amd import vcfnative
    from trtools_amd.dumpSTR import dumpSTR
    rng = np.random.default_rng(seed)
    d = tmp_path_factory.mktemp('ffz')
    src = str(d / 'in.vcf')
    _hipstr_file(rng, src, n_rec, S, hard, crlf)
    kw = dict(hipstr_min_call_DP=20, hipstr_max_call_DP=50, hipstr_min_call_Q=0.9, min_locus_callrate=0.2)
    res = []
    for tag, env in (('dev', {}), ('py', dict(TRK_DUMPSTR_BATCH='0', TRK_NATIVE_VCF='0', TRK_NATIVE_WRITER='0'))):
        with lab_env(**env):
            out = str(d / tag)
            before = dict(vcfnative.DEVICE_FORMAT)
            try:
                rc = dumpSTR.main(dump_args(out, src, vcftype='hipstr', **kw))
                err = None
            except Exception as e:           # both paths must refuse the same files
                rc, err = None, type(e).__name__
            took = vcfnative.DEVICE_FORMAT['records'] - before['records']
            if rc == 0:
                files = tuple('\n'.join(x for x in open(out + ext).read().split('\n') if not x.startswith('##command-DumpSTR'))
                              for ext in ('.vcf', '.samplog.tab', '.loclog.tab'))
            else:
                files = None
            res.append((rc, err, files, took, dumpSTR.LAST_RUN.get('path') if rc == 0 else None))
    (rc_a, err_a, fa, took_a, path_a), (rc_b, err_b, fb, took_b, path_b) = res
    assert (rc_a, err_a) == (rc_b, err_b), (seed, res[0][:2], res[1][:2])
    if rc_a == 0:
        assert path_a in ('batch', 'mixed') and path_b != 'batch' and took_b == 0, (path_a, path_b)
        if fa != fb:
            for name, x, y in zip(('vcf', 'samplog', 'loclog'), fa, fb):
                if x != y:
                    la, lb = x.split('\n'), y.split('\n')
                    i = next((i for i, (p, q) in enumerate(zip(la, lb)) if p != q), min(len(la), len(lb)))
                    ca, cb = (la[i].split('\t') if i < len(la) else []), (lb[i].split('\t') if i < len(lb) else [])
                    j = next((j for j, (p, q) in enumerate(zip(ca, cb)) if p != q), -1)
                    raise AssertionError("seed %d %s line %d column %d: device %r python %r" %
                                         (seed, name, i, j, ca[j][:80] if 0 <= j < len(ca) else len(ca), cb[j][:80] if 0 <= j < len(cb) else len(cb)))
        COUNTS['cases'] += 1
