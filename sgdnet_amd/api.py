"""sgdnet(): host-side mirror of the reference's R front-end for the fit path.

Follows sgdnet.default() of the reference (R/sgdnet.R:183-433): same argument
names and defaults, same validation messages, same response encoding, the same
14-field control list handed to the native backend (R/sgdnet.R:346-359) and the
same post-processing of the returned list (R/sgdnet.R:368-431).  The native
backend is libsgdnet_hip.so (include/sgdnet_hip.h); nothing here computes.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import FAMILIES, MODES, check, dptr

_FAMILY_CHOICES = ("gaussian", "binomial", "multinomial", "mgaussian")


@dataclass
class SgdnetFit:
    """The S3 object built at R/sgdnet.R:412-431 (class c("sgdnet_<family>", "sgdnet"))."""
    a0: np.ndarray                 # (n_lambda,) or (n_classes, n_lambda)
    beta: object                   # (p, n_lambda) array, or list of n_classes such arrays
    lambda_: np.ndarray
    dev_ratio: np.ndarray
    df: np.ndarray
    nulldev: float
    npasses: float
    alpha: float
    offset: bool
    classnames: object
    grouped: bool
    nobs: int
    family: str
    dfmat: object = None
    diagnostics: dict = field(default_factory=dict)
    return_codes: np.ndarray = None
    draws_used: int = 0


def _stop(msg):
    raise ValueError(msg)


def _levels(y):
    """(levels, counts, codes) of a class vector (R: factor levels, table(), as.numeric(y) - 1)."""
    y = np.asarray(y).reshape(-1)
    levels, counts = np.unique(y, return_counts=True)
    return levels, counts, np.searchsorted(levels, y).astype(np.float64)


def _has_nan(a):
    """any(is.na(.)): the sum is NaN (or +-inf) iff something is off; only then look closer."""
    a = np.asarray(a)
    if a.dtype.kind != "f" or a.size == 0:
        return False
    with np.errstate(invalid="ignore", over="ignore"):
        t = a.sum()
    return bool(np.isnan(t) and np.isnan(a).any())


def sgdnet(x, y, family="gaussian", alpha=1, nlambda=100, lambda_min_ratio=None, lambda_=None,
           maxit=1000, standardize=True, intercept=True, thresh=0.001,
           standardize_response=False, *, debug=False, seed=0, rng=None, sample_stream=None,
           unif=None, mode="exact", batch=0, device=0, devices=None):
    """Fit an elastic-net GLM path with SAGA on one MI355X.

    Positional/keyword arguments up to `standardize_response` are those of the reference's
    sgdnet.default (lambda.min.ratio -> lambda_min_ratio, lambda -> lambda_).  Keyword-only
    extensions: debug (options(sgdnet.debug)), seed (set.seed) or rng (an RRng whose state is
    advanced like R's .Random.seed), sample_stream / unif (explicit sample order), mode /
    batch / device (backend).
    """
    import scipy.sparse as sp

    n_samples = x.shape[0]
    n_features = x.shape[1] if len(x.shape) > 1 else 1
    y_arr = np.asarray(y)
    n_targets = y_arr.shape[1] if y_arr.ndim > 1 else 1

    if not all(isinstance(v, (bool, np.bool_)) for v in (intercept, standardize, debug)):
        _stop("intercept, standardize and debug must be logical")
    if y_arr.shape[0] != n_samples:
        _stop("the number of samples in 'x' and 'y' must match")
    if y_arr.shape[0] == 0:
        _stop("the response (y) is empty.")
    if n_samples == 0:
        _stop("the predictor matrix (x) is empty.")

    is_sparse = sp.issparse(x)
    if is_sparse:
        x = sp.csc_matrix(x, dtype=np.float64)
        x.sort_indices()
    else:
        x = np.asfortranarray(np.asarray(x, dtype=np.float64).reshape(n_samples, -1))

    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n_samples < n_features else 0.0001
    if lambda_ is None or lambda_ is False:
        lam = np.zeros(0)
    else:
        lam = np.atleast_1d(np.asarray(lambda_, dtype=np.float64))
        nlambda = lam.size
    if nlambda == 0:
        _stop("lambda path cannot be of zero length.")
    if alpha < 0 or alpha > 1:
        _stop("elastic net mixing parameter (alpha) must be in [0, 1].")
    if np.any(lam < 0):
        _stop("penalty strengths (lambdas) must be positive.")
    xvals = x.data if is_sparse else x
    if _has_nan(xvals) or _has_nan(y_arr):
        _stop("NA values are not allowed.")
    if thresh < 0:
        _stop("threshold for stopping criteria cannot be negative.")
    if maxit <= 0:
        _stop("maximum number of iterations cannot be negative or zero.")

    type_multinomial = "ungrouped"
    grouped = False
    if family not in _FAMILY_CHOICES:
        _stop("'arg' should be one of " + ", ".join(f"'{f}'" for f in _FAMILY_CHOICES))

    class_names = None
    if family == "gaussian":
        if n_targets > 1:
            _stop("response for Gaussian regression must be one-dimensional.")
        if y_arr.dtype.kind not in "fiub":
            _stop("non-numeric response.")
        n_classes = 1
        y_enc = y_arr.astype(np.float64).reshape(-1)
    elif family == "binomial":
        levels, counts, codes = _levels(y_arr)
        if levels.size > 2:
            _stop("more than two classes in response. Are you looking for family = 'multinomial'?")
        if levels.size == 1:
            _stop("only one class in response.")
        n_classes = 1
        if counts.min() <= 1:
            _stop(f"one class only has {counts.min()} observations.")
        class_names = [str(v) for v in levels]
        y_enc = codes
    elif family == "multinomial":
        levels, counts, codes = _levels(y_arr)
        class_names = [str(v) for v in levels]
        n_classes = levels.size
        if n_classes == 2:
            _stop("only two classes in response. Are you looking for family = 'binomial'?")
        if n_classes == 1:
            _stop("only one class in response.")
        if counts.min() <= 1:
            _stop(f"one class only has {counts.min()} observations.")
        y_enc = codes
    else:
        if n_targets == 1:
            _stop("response for multivariate Gaussian regression must not be one-dimensional; "
                  "try family = 'gaussian'.")
        if y_arr.dtype.kind not in "fiub":
            _stop("non-numeric response.")
        grouped = True
        n_classes = n_targets
        y_enc = y_arr.astype(np.float64)

    y_mat = np.asfortranarray(y_enc.reshape(n_samples, -1))

    # ---- control list, R/sgdnet.R:346-359 ----
    ctl = _lib.Control()
    ctl.debug = int(debug)
    ctl.elasticnet_mix = float(alpha)
    ctl.family = FAMILIES[family]
    ctl.intercept = int(intercept)
    ctl.is_sparse = int(is_sparse)
    ctl.lambda_ = dptr(lam) if lam.size else None
    ctl.n_lambda_user = lam.size
    ctl.lambda_min_ratio = float(lambda_min_ratio)
    ctl.max_iter = int(maxit)
    ctl.n_lambda = int(nlambda)
    ctl.n_classes = int(n_classes)
    ctl.standardize = int(standardize)
    ctl.standardize_response = int(standardize_response)
    ctl.tol = float(thresh)
    ctl.type_multinomial = 0 if type_multinomial == "ungrouped" else 1
    keep = []
    if sample_stream is not None:
        ss = np.ascontiguousarray(sample_stream, dtype=np.uint32)
        ctl.sample_stream = ss.ctypes.data_as(C.POINTER(C.c_uint32))
        ctl.sample_stream_len = ss.size
        keep.append(ss)
    elif unif is not None:
        cb = _lib.UNIF_FN(lambda _ctx: float(unif()))
        ctl.unif = cb
        keep.append(cb)
    ctl.seed = int(seed) & 0xFFFFFFFF
    if rng is not None:
        ctl.rng_state = C.pointer(rng.state)
    ctl.mode = MODES[mode]
    ctl.batch = int(batch)
    ctl.device = int(device)
    if devices is not None and len(devices) > 1:          # the fit sharded over several GPUs (control.n_gpus, ABI 4)
        dev_arr = (C.c_int * len(devices))(*[int(v) for v in devices])
        keep.append(dev_arr)
        ctl.n_gpus = len(devices)
        ctl.devices = dev_arr

    K, p, nl = n_classes, n_features, nlambda
    a0 = np.zeros((K, nl), order="F")
    beta = np.zeros((K, p, nl), order="F")
    lam_out = np.zeros(nl)
    dev_ratio = np.zeros(nl)
    rcodes = np.zeros(nl)
    losses = np.zeros((maxit, nl), order="F") if debug else None
    llen = np.zeros(nl, dtype=np.int32)
    res = _lib.Result()
    res.a0, res.beta, res.lambda_ = dptr(a0), dptr(beta), dptr(lam_out)
    res.dev_ratio, res.return_codes = dptr(dev_ratio), dptr(rcodes)
    if debug:
        res.losses = dptr(losses)
        res.losses_len = llen.ctypes.data_as(C.POINTER(C.c_int32))

    L = _lib.load()
    # ---- the two native call sites, R/sgdnet.R:362-366 ----
    if is_sparse:
        csc = _lib.Csc()
        colptr = np.ascontiguousarray(x.indptr, dtype=np.int32)
        rowidx = np.ascontiguousarray(x.indices, dtype=np.int32)
        vals = np.ascontiguousarray(x.data, dtype=np.float64)
        csc.n_rows, csc.n_cols = n_samples, n_features
        csc.colptr = colptr.ctypes.data_as(C.POINTER(C.c_int32))
        csc.rowidx = rowidx.ctypes.data_as(C.POINTER(C.c_int32))
        csc.values = dptr(vals)
        check(L.sgdnet_fit_sparse(C.byref(csc), dptr(y_mat), y_mat.shape[1], C.byref(ctl),
                                  C.byref(res)))
    else:
        check(L.sgdnet_fit_dense(dptr(x), n_samples, n_features, dptr(y_mat), y_mat.shape[1],
                                 C.byref(ctl), C.byref(res)))

    # ---- post-processing, R/sgdnet.R:368-431 ----
    dfmat = None
    if family in ("gaussian", "binomial"):
        a0_out = a0[0, :].copy()
        beta_out = beta[0, :, :].copy()                       # (p, n_lambda)
        df = (beta_out != 0).sum(axis=0)
    else:
        a0_out = a0.copy()
        beta_out = [beta[k, :, :].copy() for k in range(K)]
        df = (sum(beta_out) != 0).sum(axis=0)
        dfmat = np.vstack([(np.abs(bk) > 0).sum(axis=0) for bk in beta_out])
    if family == "multinomial":                               # R/sgdnet.R:409-410
        a0_out = a0_out - a0_out.mean(axis=0, keepdims=True)

    fit = SgdnetFit(a0=a0_out, beta=beta_out, lambda_=lam_out, dev_ratio=dev_ratio, df=df,
                    nulldev=res.nulldev, npasses=res.npasses, alpha=alpha, offset=False,
                    classnames=class_names, grouped=grouped, nobs=n_samples, family=family,
                    dfmat=dfmat, return_codes=rcodes, draws_used=res.draws_used)
    if debug:
        fit.diagnostics = {"loss": [losses[:llen[i], i].copy() for i in range(nl)]}
    return fit
