"""predict() / coef() for SgdnetFit objects: host mirror of the reference's R/predict.sgdnet.R.

Array conventions (numpy, 0-based) for what R returns:
  single-response families: (n_new, n_lambda)
  multinomial / mgaussian:  (n_new, n_classes, n_lambda)          (R: aperm(dp, c(3, 1, 2)))
"""
import numpy as np


def lambda_interpolate(lam, s):
    """R/predict.sgdnet.R:144-168 (from glmnet): indices (0-based) and mixing fractions."""
    lam = np.asarray(lam, dtype=np.float64)
    s = np.atleast_1d(np.asarray(s, dtype=np.float64)).copy()
    if lam.size == 1:
        z = np.zeros(s.size, dtype=np.int64)
        return z, z.copy(), np.ones(s.size)
    s = np.clip(s, lam.min(), lam.max())
    k = lam.size
    sfrac = (lam[0] - s) / (lam[0] - lam[k - 1])
    lamn = (lam[0] - lam) / (lam[0] - lam[k - 1])
    coord = np.interp(sfrac, lamn, np.arange(1, k + 1, dtype=np.float64))     # stats::approx
    left = np.floor(coord).astype(np.int64) - 1
    right = np.ceil(coord).astype(np.int64) - 1
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = (sfrac - lamn[right]) / (lamn[left] - lamn[right])
    frac[left == right] = 1.0
    frac[np.abs(lamn[left] - lamn[right]) < np.finfo(np.float64).eps] = 1.0
    return left, right, frac


def _bind(fit):
    """bind_intercept (R/predict.sgdnet.R:228-238): list over classes of (p + 1, n_lambda)."""
    if isinstance(fit.beta, list):
        return [np.vstack([fit.a0[k:k + 1, :], b]) for k, b in enumerate(fit.beta)]
    return [np.vstack([np.atleast_2d(fit.a0), fit.beta])]


def coef(fit, s=None):
    """coef.sgdnet: (p + 1, n_lambda or len(s)) with the intercept in row 0; a list per class for
    multinomial / mgaussian."""
    nb = _bind(fit)
    if s is not None:
        s = np.atleast_1d(np.asarray(s, dtype=np.float64))
        if np.any(s < 0):
            raise ValueError("s (lambda penalty) cannot be negative")
        left, right, frac = lambda_interpolate(fit.lambda_, s)
        nb = [b[:, left] * frac + b[:, right] * (1.0 - frac) for b in nb]       # :251-266
    return nb if isinstance(fit.beta, list) else nb[0]


def _stacked_coefficients(nb):
    """(a0, beta) as the C ABI holds them: a0 K x L and beta K x p x L, K fastest."""
    K, L = len(nb), nb[0].shape[1]
    p = nb[0].shape[0] - 1
    a0 = np.empty((L, K))
    beta = np.empty((L, p, K))
    for k, m in enumerate(nb):
        a0[:, k] = m[0, :]
        beta[:, :, k] = m[1:, :].T
    return np.ascontiguousarray(a0), np.ascontiguousarray(beta), K, p, L


def device_link(nb, newx, device):
    """cbind2(1, newx) %*% beta for every lambda on the GPU (sgdnet_predict_*, score.hip):
    list over classes of (n, n_lambda) arrays."""
    import ctypes as C
    import scipy.sparse as sp
    from . import _lib
    from ._lib import check, dptr
    a0, beta, K, p, L = _stacked_coefficients(nb)
    Lh = _lib.load()
    if sp.issparse(newx):
        X = sp.csr_matrix(newx, dtype=np.float64)
        X.sort_indices()
        n = X.shape[0]
        link = np.empty((n, L, K))
        ptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(X.indices, dtype=np.int32)
        val = np.ascontiguousarray(X.data, dtype=np.float64)
        check(Lh.sgdnet_predict_sparse(C.c_int64(n), C.c_int64(p), ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                       idx.ctypes.data_as(C.POINTER(C.c_int32)), dptr(val), C.c_int(K), dptr(a0),
                                       dptr(beta), C.c_int(L), C.c_int(device), dptr(link)))
    else:
        X = np.ascontiguousarray(newx, dtype=np.float64)
        n = X.shape[0]
        link = np.empty((n, L, K))
        check(Lh.sgdnet_predict_dense(dptr(X), C.c_int64(n), C.c_int64(p), C.c_int(K), dptr(a0), dptr(beta),
                                      C.c_int(L), C.c_int(device), dptr(link)))
    return [link[:, :, k] for k in range(K)]


def predict(fit, newx=None, s=None, type="link", exact=False, device=None):
    """predict.sgdnet_<family> (R/predict.sgdnet.R:347-583).  `exact=TRUE` (refit at s) is a
    front-end loop around sgdnet() and is not mirrored: call sgdnet() with the wanted lambdas.
    device: compute the linear predictors on that GPU (sgdnet_predict_*) instead of in numpy."""
    if exact:
        raise NotImplementedError("exact = TRUE refits the model: call sgdnet() with lambda_ = s instead")
    fam = fit.family
    allowed = ["link", "response", "coefficients", "nonzero"] + (["class"] if fam in ("binomial", "multinomial") else [])
    if type not in allowed:
        raise ValueError("'arg' should be one of " + ", ".join(f"'{a}'" for a in allowed))
    multi = isinstance(fit.beta, list)
    nb = coef(fit, s)
    if type == "coefficients":
        return nb
    if type == "nonzero":                                       # nonzero_coefs(beta[-1, , drop = FALSE]) of the
        b = [m[1:, :] for m in nb] if multi else nb[1:, :]      # coefficients AT s (R/predict.sgdnet.R), 0-based
        if multi and fit.grouped:
            b = b[0]
        mats = b if isinstance(b, list) else [b]
        out = [[np.flatnonzero(np.abs(m[:, i]) > 0) for i in range(m.shape[1])] for m in mats]
        return out if isinstance(b, list) else out[0]
    if newx is None:
        raise ValueError(f"you need to supply a value for 'newx' for type = '{type}'")
    import scipy.sparse as sp
    X = sp.csr_matrix(newx) if sp.issparse(newx) else np.asarray(newx, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(1, -1)
    mats = nb if multi else [nb]
    if device is not None:
        if X.shape[1] != mats[0].shape[0] - 1:
            raise ValueError("newx has the wrong number of features")
        link = device_link(mats, X, device)
    else:
        link = [np.asarray(X @ m[1:, :]) + m[0:1, :] for m in mats]            # cbind2(1, newx) %*% beta
    if not multi:
        f = link[0]
        if fam == "binomial":
            if type == "response":
                return 1.0 / (1.0 + np.exp(-f))
            if type == "class":
                names = np.asarray(fit.classnames, dtype=object)
                return names[(f > 0).astype(int)]
        return f
    dp = np.stack(link, axis=1)                                   # (n, K, n_lambda)
    if fam == "mgaussian" or type == "link":
        return dp
    if type == "response":
        pp = np.exp(dp)
        return pp / pp.sum(axis=1, keepdims=True)
    cls = np.argmax(dp, axis=1)                                   # softmax(): first maximum wins
    return np.asarray(fit.classnames, dtype=object)[cls] if fit.classnames is not None else cls + 1
