"""Randomised VCF text through both decoders: the native reader (include/trk_vcf.h) must return what the Python decoder
(vcfio.py, pinned through the reference's golden outputs) returns -- genotype arrays with mixed ploidy / phasing /
partial and missing calls, Integer and Float fields with '.', ragged vectors, negative and exponent notation, String
fields, sample columns with trailing fields dropped, CRLF, plain / gzip / bgzip containers.  CPU only."""
import gzip
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from test_vcfnative import _compare


def _gt(rng, max_ploidy):
    p = int(rng.integers(1, max_ploidy + 1))
    toks = ['.' if rng.random() < 0.15 else str(int(rng.integers(0, 12))) for _ in range(p)]
    return ('|' if rng.random() < 0.5 else '/').join(toks)


def _int(rng):
    r = rng.random()
    if r < 0.12:
        return '.'
    v = int(rng.integers(-50, 5000))
    return str(v)


def _float(rng):
    r = rng.random()
    if r < 0.12:
        return '.'
    v = float(rng.normal()) * 10 ** int(rng.integers(-6, 6))
    return rng.choice(['%g' % v, '%.3f' % v, '%e' % v, '%.1f' % v, repr(round(v, 2))])


def _vec(rng, f, nmax):
    n = int(rng.integers(1, nmax + 1))
    return ','.join(f(rng) for _ in range(n))


def _write(path, text, container):
    if container == 'plain':
        with open(path, 'wb') as fh:
            fh.write(text)
    elif container == 'gzip':
        with gzip.open(path, 'wb') as fh:
            fh.write(text)
    else:
        from trtools_amd.bgzf import BgzfWriter
        with BgzfWriter(path, threads=1) as fh:
            fh.write(text)


@settings(max_examples=120, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n_rec=st.integers(1, 12), S=st.integers(1, 40), max_ploidy=st.integers(1, 3),
       container=st.sampled_from(['plain', 'gzip', 'bgzip']), crlf=st.booleans())
def test_random_vcf_text(tmp_path_factory, seed, n_rec, S, max_ploidy, container, crlf):
    rng = np.random.default_rng(seed)
    nl = '\r\n' if crlf else '\n'
    lines = ['##fileformat=VCFv4.2', '##command=HipSTR-v0.6.2 fuzz',
             '##INFO=<ID=START,Number=1,Type=Integer,Description="s">', '##INFO=<ID=END,Number=1,Type=Integer,Description="e">',
             '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">', '##INFO=<ID=AF,Number=A,Type=Float,Description="f">',
             '##INFO=<ID=FLAG,Number=0,Type=Flag,Description="f">', '##INFO=<ID=NOTE,Number=1,Type=String,Description="n">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
             '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">', '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="a">',
             '##FORMAT=<ID=PL,Number=3,Type=Float,Description="p">', '##FORMAT=<ID=GB,Number=1,Type=String,Description="b">',
             '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    keys_all = ['DP', 'Q', 'AD', 'PL', 'GB']
    pos = 100
    for r in range(n_rec):
        pos += int(rng.integers(1, 500))
        n_alt = int(rng.integers(0, 5))
        ref = 'AC' * int(rng.integers(1, 6))
        alts = ','.join('AC' * int(rng.integers(1, 9)) for _ in range(n_alt)) or '.'
        keys = [k for k in keys_all if rng.random() < 0.7]
        info = ['START=%d' % pos, 'END=%d' % (pos + len(ref) - 1), 'PERIOD=2']
        if rng.random() < 0.5 and n_alt:
            info.append('AF=' + ','.join('%g' % rng.random() for _ in range(n_alt)))
        if rng.random() < 0.3:
            info.append('FLAG')
        if rng.random() < 0.3:
            info.append('NOTE=x_%d' % r)
        cols = []
        for s in range(S):
            toks = [_gt(rng, max_ploidy)]
            for k in keys:
                toks.append({'DP': lambda: _int(rng), 'Q': lambda: _float(rng), 'AD': lambda: _vec(rng, _int, 4),
                             'PL': lambda: ('.' if rng.random() < 0.1 else ','.join(_float(rng) for _ in range(3))),
                             'GB': lambda: rng.choice(['.', '0|0', '-2|4', 'x', 'a;b|c'])}[k]())
            if rng.random() < 0.15 and len(toks) > 1:
                toks = toks[:int(rng.integers(1, len(toks)))]       # trailing fields dropped
            cols.append(':'.join(toks))
        filt = rng.choice(['.', 'PASS', 'q10', 'q10;s50'])
        qual = rng.choice(['.', '30', '12.5'])
        lines.append('\t'.join(['chr1', str(pos), rng.choice(['.', 'id%d' % r]), ref, alts, qual, filt, ';'.join(info),
                                ':'.join(['GT'] + keys)] + cols))
    text = (nl.join(lines) + nl).encode()
    d = tmp_path_factory.mktemp('fuzz')
    path = str(d / ('f.vcf' if container == 'plain' else 'f.vcf.gz'))
    _write(path, text, container)
    # (a Float plane wider than a record's own vectors cannot tell padding from '.': PL is written with its
    # declared three values or a single '.', as the fixed-Number fields the CLIs select are)
    assert _compare(path, batch_records=int(rng.integers(1, 8)), max_ploidy=max_ploidy) == n_rec
    # and with the scalar planes alone: the one-scan form of a sample (parse_record, round 4)
    assert _compare(path, batch_records=int(rng.integers(1, 8)), max_ploidy=max_ploidy, only=('DP', 'Q')) == n_rec
    os.remove(path)


@settings(max_examples=80, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n_rec=st.integers(1, 8), S=st.integers(1, 30), caller=st.sampled_from(['hipstr', 'gangstr']))
def test_random_string_fields_preparse(tmp_path_factory, seed, n_rec, S, caller):
    """The native pre-parsers of HipSTR's ALLREADS/GB (minimum supporting reads) and GangSTR's RC / REPCI against the
    Python pre-parsers of dumpSTR/filters.py on random field text: read counts for lengths the call does not have,
    absent lengths, negative base-pair differences, missing fields, unphased and phased GB, no-calls."""
    from trtools_amd import vcfio, vcfnative
    from trtools_amd.dumpSTR import filters
    from trtools_amd.utils import tr_harmonizer as trh
    rng = np.random.default_rng(seed)
    hdr = ['##fileformat=VCFv4.2']
    if caller == 'hipstr':
        hdr += ['##command=HipSTR-v0.6.2 fuzz', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
                '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
                '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=GB,Number=1,Type=String,Description="b">',
                '##FORMAT=<ID=ALLREADS,Number=1,Type=String,Description="a">']
        fkeys = 'GT:GB:ALLREADS'
    else:
        hdr += ['##command=GangSTR-2.4 fuzz', '##INFO=<ID=RU,Number=1,Type=String,Description="m">',
                '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=RC,Number=1,Type=String,Description="r">',
                '##FORMAT=<ID=REPCI,Number=1,Type=String,Description="c">', '##FORMAT=<ID=REPCN,Number=2,Type=Integer,Description="n">']
        fkeys = 'GT:REPCN:REPCI:RC'
    hdr.append('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S)))
    lines = []
    for r in range(n_rec):
        pos = 1000 + 100 * r
        cols = []
        sep = rng.choice(['|', '/'])
        for s in range(S):
            nocall = rng.random() < 0.15
            a0, a1 = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            gt = '.' if nocall and rng.random() < 0.5 else ('./.' if nocall else '%d%s%d' % (a0, sep, a1))
            if caller == 'hipstr':
                d0, d1 = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
                gb = '.' if nocall else '%d%s%d' % (d0, sep, d1)
                if rng.random() < 0.15:
                    ar = '.'
                else:
                    ks = sorted(set(int(x) for x in rng.integers(-8, 9, size=int(rng.integers(1, 5)))) |
                                ({d0} if rng.random() < 0.8 else set()) | ({d1} if rng.random() < 0.8 else set()))
                    ar = ';'.join('%d|%d' % (k, int(rng.integers(1, 40))) for k in ks)
                cols.append(':'.join([gt, gb, ar]))
            else:
                if nocall:
                    cols.append(':'.join([gt, '.', '.', '.']))
                    continue
                n0, n1 = int(rng.integers(2, 30)), int(rng.integers(2, 30))
                # (bounds stay non-negative: 'lo-hi' has no room for a sign and the reference's own split fails on one)
                ci = '%d-%d,%d-%d' % (max(0, n0 - int(rng.integers(0, 4))), n0 + int(rng.integers(0, 4)),
                                      max(0, n1 - int(rng.integers(0, 4))), n1 + int(rng.integers(0, 4)))
                rc = ','.join(str(int(x)) for x in rng.integers(0, 60, size=4))
                cols.append(':'.join([gt, '%d,%d' % (n0, n1), ci, rc]))
        if caller == 'hipstr':
            info = 'START=%d;END=%d;PERIOD=2' % (pos, pos + 11)
        else:
            info = 'RU=ac'
        lines.append('\t'.join(['chr1', str(pos), '.', 'ACACACACACAC', 'ACACACAC,ACACACACACACACAC', '.', '.', info, fkeys] + cols))
    d = tmp_path_factory.mktemp('pre')
    path = str(d / 'p.vcf')
    with open(path, 'w') as fh:
        fh.write('\n'.join(hdr + lines) + '\n')
    r = vcfnative.NativeVCFReader(path)
    if caller == 'hipstr':
        r.select_format('ALLREADS', vcfnative.KIND_MINSUPP, 1, alias='__minsupp')
    else:
        r.select_format('RC', vcfnative.KIND_INT, 4, alias='__rc')
        r.select_format('REPCI', vcfnative.KIND_INT_RANGES, 4, alias='__repci')
    n = 0
    for vpy, vnat in zip(vcfio.VCFReader(path), r):
        rec = trh.HarmonizeRecord(caller, vpy)
        called = rec.GetCalledSamples()
        if caller == 'hipstr':
            assert np.array_equal(filters._min_supp_reads(rec)[:, 0][called], vnat.format('__minsupp')[:, 0][called]), vpy.POS
        elif called.any():
            assert np.array_equal(filters._rc_plane(rec)[called], vnat.format('__rc')[called]), vpy.POS
            assert np.array_equal(filters._repci_plane(rec)[called], vnat.format('__repci')[called]), vpy.POS
        n += 1
    assert n == n_rec
    os.remove(path)
