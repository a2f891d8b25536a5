"""Build-container stand-in for the absent ``statsmodels`` wheel (golden generation only).

Only what trtools/associaTR/associaTR.py touches on its non-plotting path is provided:
``OLS(endog, exog, missing='drop').fit()`` with ``params / bse / pvalues / rsquared``.
The arithmetic follows statsmodels' published algorithm (0.13/0.14,
regression/linear_model.py: RegressionModel.fit(method='pinv'), RegressionResults):
Moore-Penrose pseudo-inverse through an SVD with rcond 1e-15, rank from the singular values,
``scale = ssr / (nobs - rank)``, ``bse = sqrt(diag(pinv pinv') * scale)``,
``pvalues = 2 * t.sf(|params / bse|, nobs - rank)``, centred R^2 when exactly one column is
constant.  The stand-in is pinned by the plink2 fixtures of the reference's own test suite
(tools/gen_golden_associatr.py runs the reference's comparator on every case).
Never shipped, never imported by the product.
"""
__version__ = "0.14.shim"
