"""score(): prediction error along the lambda path; host mirror of the reference's R/score.R."""
import numpy as np

from .predict import predict, coef, _stacked_coefficients

_MEASURES = {"gaussian": ("deviance", "mse", "mae"), "mgaussian": ("deviance", "mse", "mae"),
             "binomial": ("deviance", "mse", "mae", "class", "auc"),
             "multinomial": ("deviance", "mse", "mae", "class")}
PROB_MIN = 1e-05                                                  # R/score.R:88, 133


def auc(y01, prob, weights=None, tie_break=None):
    """R/score.R:220-250.  y01: (n, 2) indicator matrix or a 0/1 vector.  The reference breaks
    ties between equal probabilities with stats::runif(); pass tie_break (same length as the
    stacked vector) to reproduce a given draw, default: stable order."""
    y01 = np.asarray(y01, dtype=np.float64)
    if y01.ndim == 2:
        ny = y01.shape[0]
        w = np.ones(ny) if weights is None else np.asarray(weights, dtype=np.float64)
        return auc(np.r_[np.zeros(ny), np.ones(ny)], np.r_[prob, prob], (w[:, None] * y01).ravel("F"), tie_break)
    prob = np.asarray(prob, dtype=np.float64)
    if weights is None:
        from scipy.stats import rankdata
        r = rankdata(prob)
        n1 = y01.sum()
        n0 = y01.size - n1
        u = r[y01 == 1].sum() - n1 * (n1 + 1) / 2
        return float(np.exp(np.log(u) - np.log(n1) - np.log(n0)))
    tb = np.arange(prob.size) if tie_break is None else np.asarray(tie_break)
    op = np.lexsort((tb, prob))
    y, w = y01[op], np.asarray(weights, dtype=np.float64)[op]
    cw = np.cumsum(w)
    w1 = w[y == 1]
    cw1 = np.cumsum(w1)
    wauc = np.log(np.sum(w1 * (cw[y == 1] - cw1)))
    sumw1 = cw1[-1]
    sumw2 = cw[-1] - sumw1
    return float(np.exp(wauc - np.log(sumw1) - np.log(sumw2)))


def _score_device(fit, x, y, type_measure, s, device, tie_break=None, rng=None):
    """sgdnet_score_* (score.hip): linear predictors, per-sample losses and their means in one
    kernel on the GPU; x never leaves sample-major form and no (n, n_lambda) array is built.
    "auc": sgdnet_auc_* -- probabilities, two stable radix sorts (tie breaker, probability), a scan of
    the class-0 flags and an integer sum per lambda, all on the device."""
    import ctypes as C
    import scipy.sparse as sp
    from . import _lib
    from ._lib import FAMILIES, MEASURES, check, dptr
    fam = fit.family
    nb = coef(fit, s)
    nb = nb if isinstance(nb, list) else [nb]
    a0, beta, K, p, L = _stacked_coefficients(nb)
    y = np.asarray(y)
    if fam in ("binomial", "multinomial"):
        # diag(K)[as.numeric(y), ] with the factor levels of the fit (a held-out fold may miss one)
        uniq, inv = np.unique(y.ravel(), return_inverse=True)
        names = [str(v) for v in uniq]
        if fit.classnames is not None and all(nm in fit.classnames for nm in names):
            pos = np.array([list(fit.classnames).index(nm) for nm in names])
        else:
            pos = np.arange(uniq.size)
        yy = pos[inv].astype(np.float64).reshape(1, -1)
    elif fam == "mgaussian":
        yy = np.ascontiguousarray(np.asarray(y, dtype=np.float64))            # (n, K): K fastest per sample
    else:
        yy = np.asarray(y, dtype=np.float64).reshape(1, -1)
    yy = np.ascontiguousarray(yy)
    y_rows = K if fam == "mgaussian" else 1
    out = np.empty(L)
    Lh = _lib.load()
    common = (dptr(yy), C.c_int(y_rows), C.c_int(FAMILIES[fam]), C.c_int(K), dptr(a0), dptr(beta), C.c_int(L),
              C.c_int(MEASURES[type_measure]), C.c_int(device), dptr(out))
    tie = None
    if type_measure == "auc":
        n_new = yy.shape[1]
        if tie_break is not None:
            tie = np.ascontiguousarray(np.asarray(tie_break, dtype=np.float64).T.reshape(-1)
                                       if np.ndim(tie_break) == 2 else np.tile(np.asarray(tie_break, dtype=np.float64), L))
            if tie.size != 2 * n_new * L:
                raise ValueError("tie_break needs 2 n entries (or a (2 n, n_lambda) array)")
        if rng is not None and tie is not None:
            raise ValueError("pass tie_break or rng, not both")
        common = (dptr(yy), dptr(a0), dptr(beta), C.c_int(L),
                  C.byref(rng.state) if rng is not None else (dptr(tie) if tie is not None else None), C.c_int(device),
                  dptr(out))
    if type_measure == "auc" and sp.issparse(x):
        X = sp.csr_matrix(x, dtype=np.float64)
        X.sort_indices()
        ptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(X.indices, dtype=np.int32)
        val = np.ascontiguousarray(X.data, dtype=np.float64)
        fn = Lh.sgdnet_auc_sparse_rng if rng is not None else Lh.sgdnet_auc_sparse
        check(fn(C.c_int64(X.shape[0]), C.c_int64(p), ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                 idx.ctypes.data_as(C.POINTER(C.c_int32)), dptr(val), *common))
    elif type_measure == "auc":
        X = np.ascontiguousarray(x, dtype=np.float64)
        fn = Lh.sgdnet_auc_dense_rng if rng is not None else Lh.sgdnet_auc_dense
        check(fn(dptr(X), C.c_int64(X.shape[0]), C.c_int64(p), *common))
    elif sp.issparse(x):
        X = sp.csr_matrix(x, dtype=np.float64)
        X.sort_indices()
        if X.shape[1] != p:
            raise ValueError("x has the wrong number of features")
        ptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(X.indices, dtype=np.int32)
        val = np.ascontiguousarray(X.data, dtype=np.float64)
        check(Lh.sgdnet_score_sparse(C.c_int64(X.shape[0]), C.c_int64(p), ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                     idx.ctypes.data_as(C.POINTER(C.c_int32)), dptr(val), *common))
    else:
        X = np.ascontiguousarray(x, dtype=np.float64)
        if X.shape[1] != p:
            raise ValueError("x has the wrong number of features")
        check(Lh.sgdnet_score_dense(dptr(X), C.c_int64(X.shape[0]), C.c_int64(p), *common))
    return out


def score(fit, x, y, type_measure="deviance", s=None, device=None, tie_break=None, rng=None):
    """score.sgdnet_<family>: one value per lambda (or per entry of s).
    device: evaluate on that GPU.  tie_break ("auc" only): the reference orders equal probabilities by
    stats::runif(2n) drawn per lambda; pass those draws (2n values used for every lambda, or a (2n, n_lambda)
    array) to reproduce it, default: sample order.  rng (an RRng; "auc" on a device only): draw them on the device
    from that generator instead -- 2n per lambda, in lambda order -- and leave it where R's would be."""
    fam = fit.family
    if type_measure not in _MEASURES[fam]:
        raise ValueError("'arg' should be one of " + ", ".join(f"'{m}'" for m in _MEASURES[fam]))
    s = fit.lambda_ if s is None else s
    y = np.asarray(y)
    if rng is not None and type_measure == "auc" and device is None:
        if tie_break is not None:
            raise ValueError("pass tie_break or rng, not both")
        m, L = y.shape[0], np.size(s)
        tie_break = rng.unif(2 * m * L).reshape(L, 2 * m).T          # the host mirror of sgdnet_auc_*_rng
    if device is not None:
        if type_measure == "deviance" and fam in ("gaussian", "mgaussian"):
            type_measure = "mse"                                  # R/score.R:63, 180: the same number
        return _score_device(fit, x, y, type_measure, s, device, tie_break, rng if type_measure == "auc" else None)
    if fam == "gaussian":                                         # R/score.R:55-70
        yh = predict(fit, x, s)
        d = yh - y.reshape(-1, 1)
        return np.mean(np.abs(d), axis=0) if type_measure == "mae" else np.mean(d ** 2, axis=0)
    if fam == "mgaussian":                                        # :172-186 (colSums over samples, mean over responses)
        yh = predict(fit, x, s)
        d = yh - np.asarray(y, dtype=np.float64)[:, :, None]
        e = np.abs(d) if type_measure == "mae" else d ** 2
        return e.sum(axis=0).mean(axis=0)
    levels = np.unique(y)
    Y = (y.reshape(-1, 1) == levels.reshape(1, -1)).astype(np.float64)     # diag(K)[as.numeric(y), ]
    if fam == "binomial":                                         # :74-115
        ph = predict(fit, x, s, type="response")
        y1, y2 = Y[:, 0:1], Y[:, 1:2]
        if type_measure == "auc":
            tb = None if tie_break is None else np.asarray(tie_break, dtype=np.float64)
            return np.array([auc(Y, ph[:, i], tie_break=None if tb is None else (tb[:, i] if tb.ndim == 2 else tb))
                             for i in range(ph.shape[1])])
        if type_measure == "mse":
            return np.mean((ph + y1 - 1) ** 2 + (ph - y2) ** 2, axis=0)
        if type_measure == "mae":
            return np.mean(np.abs(ph + y1 - 1) + np.abs(ph - y2), axis=0)
        if type_measure == "class":
            return np.mean(y1 * (ph > 0.5) + y2 * (ph <= 0.5), axis=0)
        ph = np.clip(ph, PROB_MIN, 1 - PROB_MIN)
        lp = y1 * np.log(1 - ph) + y2 * np.log(ph)
        return np.mean(2 * (0.0 - lp), axis=0)                    # ly = sum y log y = 0 for 0/1 responses
    ph = predict(fit, x, s, type="response")                      # multinomial, :119-168: (n, K, L)
    Y3 = Y[:, :, None]
    if type_measure == "mse":
        return ((Y3 - ph) ** 2).sum(axis=1).mean(axis=0)
    if type_measure == "mae":
        return np.abs(Y3 - ph).sum(axis=1).mean(axis=0)
    if type_measure == "class":
        cls = np.argmax(ph, axis=1)                                # (n, L)
        return 1.0 - np.take_along_axis(Y, cls, axis=1).mean(axis=0)
    ph = np.clip(ph, PROB_MIN, 1 - PROB_MIN)
    return (2 * (0.0 - Y3 * np.log(ph))).sum(axis=1).mean(axis=0)
