"""VCF comparator for the tests (own code; same tolerance rules the reference's tests apply in
trtools/testsupport/utils.py:40-196): header lines as a set (##command lines only counted),
identical sample line, CHROM..FILTER and FORMAT keys exact, INFO as a key -> values mapping
(numeric values compared numerically, ``info_ignore`` keys skipped), per-sample FORMAT values
numerically (approx) with '.' == '.,.' and ``format_ignore`` fields skipped."""
import gzip

import numpy as np
import pytest


def _vals(text):
    v = np.array(text.split(','))
    try:
        return v.astype(float)
    except ValueError:
        return v


def _info(text):
    d = {}
    for pair in text.split(';'):
        if '=' not in pair:
            d[pair] = None
        else:
            k, v = pair.split('=', 1)
            d[k] = _vals(v)
    return d


def _open(path):
    return gzip.open(path, 'rt') if path.endswith('.gz') else open(path, 'rt')


def compare_vcfs(out_path, control_path, info_ignore=(), format_ignore=(), header=True):
    """Returns a list of human-readable differences (empty == same)."""
    problems = []
    with _open(out_path) as f1, _open(control_path) as f2:
        l1, l2 = f1.read().split('\n'), f2.read().split('\n')
    h1 = [x for x in l1 if x.startswith('##')]
    h2 = [x for x in l2 if x.startswith('##')]
    if header:
        c1 = sum('##command' in x for x in h1)
        c2 = sum('##command' in x for x in h2)
        if c1 != c2:
            problems.append("##command lines: %d vs %d" % (c1, c2))
        s1 = {x for x in h1 if '##command' not in x}
        s2 = {x for x in h2 if '##command' not in x}
        for x in sorted(s1 - s2):
            problems.append("header only in output: " + x)
        for x in sorted(s2 - s1):
            problems.append("header only in control: " + x)
    b1 = [x for x in l1 if x and not x.startswith('##')]
    b2 = [x for x in l2 if x and not x.startswith('##')]
    if b1[0] != b2[0]:
        problems.append("sample lines differ")
    if len(b1) != len(b2):
        problems.append("record counts differ: %d vs %d" % (len(b1) - 1, len(b2) - 1))
    for n, (r1, r2) in enumerate(zip(b1[1:], b2[1:])):
        a, b = r1.split('\t'), r2.split('\t')
        where = "record %d (%s:%s)" % (n, a[0], a[1])
        if len(a) != len(b):
            problems.append(where + " column counts differ")
            continue
        for i in (0, 1, 2, 3, 4, 5, 6, 8):
            if i < len(a) and a[i] != b[i]:
                problems.append("%s column %d: %r vs %r" % (where, i, a[i][:80], b[i][:80]))
        i1, i2 = _info(a[7]), _info(b[7])
        if i1.keys() != i2.keys():
            problems.append("%s INFO keys %s vs %s" % (where, sorted(i1), sorted(i2)))
        else:
            for k in i1:
                if k in info_ignore or i1[k] is None:
                    continue
                same = (i1[k].shape == i2[k].shape) and (
                    np.allclose(i1[k], i2[k], rtol=1e-6, atol=0) if i1[k].dtype.kind == 'f' and i2[k].dtype.kind == 'f'
                    else np.array_equal(i1[k], i2[k]))
                if not same:
                    problems.append("%s INFO %s: %s vs %s" % (where, k, i1[k], i2[k]))
        if len(a) > 9:
            keys = a[8].split(':')
            skip = {keys.index(k) for k in format_ignore if k in keys}
            for s in range(9, len(a)):
                fa, fb = a[s].split(':'), b[s].split(':')
                if len(fa) != len(fb):
                    problems.append("%s sample %d: %d vs %d FORMAT values" % (where, s - 8, len(fa), len(fb)))
                    continue
                for c, (x, y) in enumerate(zip(fa, fb)):
                    if c in skip or x == y:
                        continue
                    vx, vy = _vals(x), _vals(y)
                    if vx.dtype.kind == 'U' and np.all(vx == '.') and vy.dtype.kind == 'U' and np.all(vy == '.'):
                        continue
                    if vx.dtype.kind == 'f' and vy.dtype.kind == 'f' and vx.shape == vy.shape \
                            and pytest.approx(vx) == vy:
                        continue
                    problems.append("%s sample %d field %s: %r vs %r" % (where, s - 8, keys[c], x, y))
        if len(problems) > 40:
            problems.append("... (truncated)")
            break
    return problems
