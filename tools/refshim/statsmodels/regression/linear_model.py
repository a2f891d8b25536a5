import numpy as np
import scipy.stats


def _pinv_extended(x, rcond=1e-15):
    x = np.asarray(x)
    u, s, vt = np.linalg.svd(x, False)
    s_orig = np.copy(s)
    cutoff = rcond * np.maximum.reduce(s) if s.size else 0.0
    for i in range(s.shape[0]):
        s[i] = 1. / s[i] if s_orig[i] > cutoff else 0.
    res = np.dot(np.transpose(vt), np.multiply(s[:, np.newaxis], np.transpose(u)))
    return res, s_orig


class _Results:
    pass


class OLS:
    def __init__(self, endog, exog, missing='none', hasconst=None):
        y = np.asarray(endog, dtype=float)
        x = np.asarray(exog, dtype=float)
        if x.ndim == 1:
            x = x[:, None]
        if missing == 'drop':
            keep = ~(np.isnan(y) | np.isnan(x).any(axis=1))
            y, x = y[keep], x[keep]
        self.endog, self.exog = y, x
        # constant detection (statsmodels base/data.py:_handle_constant)
        if x.shape[0] == 0:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")
        xmax, xmin = np.max(x, axis=0), np.min(x, axis=0)
        if not np.isfinite(xmax).all():
            raise ValueError("exog contains inf or nans")
        const_idx = np.where(xmax == xmin)[0]
        self.k_constant = 0
        if const_idx.size == 1 and x[:, const_idx[0]].mean() != 0:
            self.k_constant = 1
        elif const_idx.size > 1:
            self.k_constant = int(any(x[:, i].mean() != 0 for i in const_idx))
        elif const_idx.size == 0:
            # an implicit constant (columns spanning 1) raises the rank test in statsmodels
            aug = np.column_stack((np.ones(x.shape[0]), x))
            self.k_constant = int(np.linalg.matrix_rank(aug) == np.linalg.matrix_rank(x))

    def fit(self):
        y, x = self.endog, self.exog
        pinv, sv = _pinv_extended(x)
        ncp = np.dot(pinv, np.transpose(pinv))
        rank = np.linalg.matrix_rank(np.diag(sv))
        params = np.dot(pinv, y)
        nobs = float(x.shape[0])
        df_resid = nobs - rank
        resid = y - np.dot(x, params)
        ssr = np.dot(resid, resid)
        scale = ssr / df_resid
        r = _Results()
        r.params = params
        r.bse = np.sqrt(np.diag(ncp * scale))
        r.tvalues = params / r.bse
        r.pvalues = scipy.stats.t.sf(np.abs(r.tvalues), df_resid) * 2
        if self.k_constant:
            yc = y - np.mean(y)
            tss = np.dot(yc, yc)
        else:
            tss = np.dot(y, y)
        r.rsquared = 1 - ssr / tss
        r.df_resid = df_resid
        r.fittedvalues = np.dot(x, params)
        return r
