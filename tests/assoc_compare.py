"""Comparators for associaTR tables.

``compare_to_plink`` applies the acceptance rules of the reference's own test-suite
(associaTR/tests/test_associaTR.py:39-84: same sign, same exponent, third significant digit
within 2) between an associaTR table and a plink2 ``.glm.linear`` fixture — restated here on
plain lists (no pandas).  ``compare_tables`` is the tight check between two associaTR tables:
text columns equal, float columns within a relative tolerance.
"""
import math

import numpy as np


def _fmt(x):
    return np.format_float_scientific(x, precision=2, unique=False)


def comp_floats(f1, f2, slack=2):
    assert np.sign(f1) == np.sign(f2), (f1, f2)
    a, b = _fmt(abs(f1)), _fmt(abs(f2))
    assert a[:2] == b[:2], (f1, f2)
    assert abs(int(a[2:4]) - int(b[2:4])) <= slack, (f1, f2)
    assert a[5:] == b[5:], (f1, f2)


def read_table(path):
    with open(path) as fh:
        header = fh.readline().rstrip('\n').split('\t')
        rows = [line.rstrip('\n').split('\t') for line in fh]
    return header, rows


def compare_to_plink(assoc_file, plink_file, pheno, skip_filtered=False):
    h, rows = read_table(assoc_file)
    ph, prows = read_table(plink_file)
    col = {k: i for i, k in enumerate(h)}
    pcol = {k: i for i, k in enumerate(ph)}
    if skip_filtered:
        rows = [r for r in rows if r[col['locus_filtered']] == 'False']
        prows = [r for r in prows if r[pcol['ERRCODE']] == '.' and len(r[pcol['REF']]) != len(r[pcol['ALT']])]
    assert len(rows) == len(prows), (len(rows), len(prows))
    n = 0
    for r, p in zip(rows, prows):
        out_p = float(r[col['p_' + pheno]])
        if not skip_filtered and math.isnan(out_p):
            if ',' in r[col['alleles']]:
                assert p[pcol['ERRCODE']] != '.'
            continue
        comp_floats(out_p, float(p[pcol['P']]))
        ref_len = float(r[col['ref_len']])
        alleles = [float(x) for x in r[col['alleles']].split(',')]
        assert len(alleles) == 2
        diff = abs(alleles[0] - alleles[1])
        sign = 1 if ref_len == min(alleles) else -1
        comp_floats(float(r[col['coeff_' + pheno]]) * diff * sign, float(p[pcol['BETA']]))
        comp_floats(float(r[col['se_' + pheno]]) * diff, float(p[pcol['SE']]))
        n += 1
    return n


FLOAT_COLS = ('p_', 'coeff_', 'se_', 'regression_R^2')


def compare_tables(got_file, want_file, rtol=1e-9, p_rtol=None):
    """Rows of two associaTR tables: identical text, float columns within rtol (nan == nan).
    Returns the number of compared numeric cells."""
    gh, grows = read_table(got_file)
    wh, wrows = read_table(want_file)
    assert gh == wh, (gh, wh)
    assert len(grows) == len(wrows), (len(grows), len(wrows))
    isf = [any(c.startswith(p) for p in FLOAT_COLS) for c in gh]
    cells = 0
    for ln, (g, w) in enumerate(zip(grows, wrows)):
        assert len(g) == len(w), (ln, g, w)
        for j, (a, b) in enumerate(zip(g, w)):
            if not isf[j] or j >= len(gh):
                assert a == b, (ln, gh[j] if j < len(gh) else j, a, b)
                continue
            fa, fb = float(a), float(b)
            if math.isnan(fb):
                assert math.isnan(fa), (ln, gh[j], a, b)
                continue
            tol = p_rtol if (p_rtol is not None and gh[j].startswith('p_')) else rtol
            assert abs(fa - fb) <= tol * max(abs(fb), 1e-300), (ln, gh[j], a, b)
            cells += 1
    return cells
