"""An independent pin of the OLS restatement behind SURVEY row f3 (associaTR).

statsmodels -- the third-party call at associaTR.py:281-291 -- is absent from the reference checkout and from this
image, so both ``oracle/associatr_oracle.ols_pinv`` and the stand-in that let the real reference code write the
golden tables (``tools/refshim/statsmodels``) restate its published ``OLS.fit(method='pinv')``.  The only numbers the
reference itself holds are the plink2 fixtures (six significant digits: tests/test_assoc_oracle.py pins to ~1e-5).
This test closes the gap with what the image does have: on EVERY design matrix the 17 golden cases produce (all loci,
real covariates, sample subsets, dosage regressors) both restatements must agree to 1e-10 with

  * ``numpy.linalg.lstsq`` (LAPACK gelsd) for the coefficients, residual sum of squares and rank,
    ``(X^T X)^-1`` by ``numpy.linalg.inv`` for the standard errors, ``scipy.stats.t.sf`` for the p-value, and
  * ``scipy.stats.linregress`` (closed-form simple regression) whenever the design is genotype + intercept.
"""
import os
import sys

import numpy as np
import pytest
import scipy.stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import associatr_oracle as ao      # noqa: E402
import assoc_cases                              # noqa: E402
from test_assoc_oracle import run_oracle        # noqa: E402

RTOL = 1e-10


def _independent(y, x):
    beta, _, rank, _ = np.linalg.lstsq(x, y, rcond=None)
    resid = y - x @ beta
    ssr = float(resid @ resid)
    df = x.shape[0] - rank
    cov = np.linalg.inv(x.T @ x) * (ssr / df)
    bse = np.sqrt(np.diag(cov))
    p = 2 * scipy.stats.t.sf(np.abs(beta / bse), df)
    yc = y - y.mean()
    return beta, bse, p, 1 - ssr / float(yc @ yc), df, rank


def _close(a, b, what):
    a, b = float(a), float(b)
    assert abs(a - b) <= RTOL * max(abs(b), 1e-300) + 1e-13 * (what != 'p'), (what, a, b)


@pytest.mark.parametrize('name', sorted(assoc_cases.CASES))
def test_ols_restatements_agree_with_lapack_and_linregress(name, tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'refshim'))
    try:
        from statsmodels.regression.linear_model import OLS as ShimOLS
    finally:
        sys.path.pop(0)
    seen = {'n': 0, 'full_rank': 0, 'simple': 0}
    real = ao.ols_pinv

    def checked(y, x):
        out = real(y, x)
        params, bse, pvalues, rsq, df = out
        y_, x_ = np.asarray(y, dtype=float), np.asarray(x, dtype=float)
        seen['n'] += 1
        shim = ShimOLS(y_, x_).fit()                       # what produced the golden tables
        _close(shim.params[0], params[0], 'coef')
        _close(shim.bse[0], bse[0], 'se')
        _close(shim.pvalues[0], pvalues[0], 'p')
        _close(shim.rsquared, rsq, 'r2')
        if np.linalg.matrix_rank(x_) == x_.shape[1] and np.linalg.cond(x_) < 1e6:
            b, se, p, r2, dfi, rank = _independent(y_, x_)
            seen['full_rank'] += 1
            assert dfi == df
            _close(params[0], b[0], 'coef')
            _close(bse[0], se[0], 'se')
            _close(pvalues[0], p[0], 'p')
            _close(rsq, r2, 'r2')
            if x_.shape[1] == 2 and np.all(x_[:, 1] == 1.0):
                lr = scipy.stats.linregress(x_[:, 0], y_)
                seen['simple'] += 1
                _close(params[0], lr.slope, 'coef')
                _close(bse[0], lr.stderr, 'se')
                _close(pvalues[0], lr.pvalue, 'p')
                _close(rsq, lr.rvalue ** 2, 'r2')
        return out

    monkeypatch.setattr(ao, 'ols_pinv', checked)
    kw, _, _ = assoc_cases.CASES[name]
    run_oracle(str(tmp_path / 'o.tsv'), kw, 10, 15)
    assert seen['n'] > 0 or 'cutoff' in name
    assert seen['full_rank'] >= 0.9 * seen['n']
