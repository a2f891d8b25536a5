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
    os.remove(path)
