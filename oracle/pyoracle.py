"""ctypes loader for the CPU oracle (oracle/sgdnet_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; sgdnet_amd/ never does.
See oracle/sgdnet_oracle.h for the parity status (pinned to the reference's printed outputs).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBS = {}
_DET = False


def use_det_math(flag):
    """Switch to liboracle_det.so: the same restatement with include/sgdnet_detmath.h's exp/log in the family
    gradients (what the HIP exact kernels compute with), instead of libm."""
    global _DET, _LIB
    _DET = bool(flag)
    _LIB = None

FAMILIES = {"gaussian": 0, "binomial": 1, "multinomial": 2, "mgaussian": 3}
PENALTIES = {"ridge": 0, "elasticnet": 1, "grouplasso": 2}


def build(force=False):
    if os.environ.get("SGDNET_ORACLE_ASAN") == "1":     # sanitizer run of the CPU test-suite
        subprocess.check_call(["make", "-C", _HERE, "liboracle_asan.so"], stdout=subprocess.DEVNULL)
        return os.path.join(_HERE, "liboracle_asan.so")
    name = "liboracle_det.so" if _DET else "liboracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "sgdnet_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    return so


class _Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int)]


class _Draws(C.Structure):
    _fields_ = [("stream", C.POINTER(C.c_uint32)), ("pos", C.c_int64), ("len", C.c_int64),
                ("rng", C.POINTER(_Rng))]


class _SagaParams(C.Structure):
    _fields_ = [("family", C.c_int), ("penalty", C.c_int), ("n_classes", C.c_int),
                ("n_samples", C.c_int64), ("n_features", C.c_int64),
                ("fit_intercept", C.c_int), ("standardize", C.c_int),
                ("gamma", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("max_iter", C.c_uint), ("tol", C.c_double), ("debug", C.c_int),
                ("n_total", C.c_int64), ("dense_intercept", C.c_int)]


class _Control(C.Structure):
    _fields_ = [("family", C.c_int), ("elasticnet_mix", C.c_double), ("fit_intercept", C.c_int),
                ("standardize", C.c_int), ("standardize_response", C.c_int),
                ("n_classes", C.c_int), ("n_lambda", C.c_int),
                ("lambda_", C.POINTER(C.c_double)), ("lambda_min_ratio", C.c_double),
                ("max_iter", C.c_uint), ("tol", C.c_double), ("debug", C.c_int),
                ("batch", C.c_int64)]


class _Result(C.Structure):
    _fields_ = [("a0", C.POINTER(C.c_double)), ("beta", C.POINTER(C.c_double)),
                ("lambda_", C.POINTER(C.c_double)), ("dev_ratio", C.POINTER(C.c_double)),
                ("return_codes", C.POINTER(C.c_double)), ("losses", C.POINTER(C.c_double)),
                ("losses_len", C.POINTER(C.c_int)), ("nulldev", C.c_double),
                ("npasses", C.c_double), ("step_size", C.POINTER(C.c_double)),
                ("alpha_l2", C.POINTER(C.c_double)), ("beta_l1", C.POINTER(C.c_double))]


def lib():
    global _LIB
    if _LIB is None:
        path = build()
        if path not in _LIBS:
            _LIBS[path] = C.CDLL(path)
        _LIB = _LIBS[path]
        _LIB.orc_unif_rand.restype = C.c_double
        _LIB.orc_draw.restype = C.c_uint32
        for f in ("orc_saga_sparse", "orc_saga_dense", "orc_saga_sparse_batched"):
            getattr(_LIB, f).restype = C.c_uint
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Rng:
    """R-compatible Mersenne-Twister: Rng(seed) == set.seed(seed)."""

    def __init__(self, seed):
        self.state = _Rng()
        lib().orc_rng_seed(C.byref(self.state), C.c_uint32(seed))

    def unif(self, count=1):
        L = lib()
        return np.array([L.orc_unif_rand(C.byref(self.state)) for _ in range(count)])

    def stream(self, n_samples, count):
        out = np.empty(count, dtype=np.uint32)
        lib().orc_fill_stream(C.byref(self.state), C.c_uint32(n_samples),
                              out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(count))
        return out


def _mk_draws(stream=None, rng=None):
    d = _Draws()
    keep = None
    if stream is not None:
        keep = np.ascontiguousarray(stream, dtype=np.uint32)
        d.stream = keep.ctypes.data_as(C.POINTER(C.c_uint32))
        d.len = keep.size
    else:
        d.stream = None
        d.rng = C.pointer(rng.state)
    d.pos = 0
    return d, keep


def batch_factors(alpha, gamma, m):
    r, ls = C.c_double(), C.c_double()
    lib().orc_batch_factors(C.c_double(alpha), C.c_double(gamma), C.c_int64(m), C.byref(r),
                            C.byref(ls))
    return r.value, ls.value


def saga(x, y, state, *, family, penalty, gamma, alpha, beta, fit_intercept=True,
         standardize=False, x_center_scaled=None, max_iter=1, tol=0.0, stream=None, rng=None,
         batch=0, debug=False, n_total=0, epoch_len=None, dense_intercept=False):
    """Run the SAGA loop for one (gamma, alpha, beta) on sample-major data.

    dense_intercept (batched restatement of a DENSE matrix handed over as CSC with every entry stored):
    the intercept step of saga-dense.h:170-173 instead of the sparse one with its 0.01 decay.

    epoch_len (timing only, bench.py's bounded cpu_baseline on config 5): the inner loop runs
    epoch_len iterations per epoch instead of n_samples, drawing from the whole data set through an
    explicit `stream`; the gradient average is then normalised by epoch_len, so the result is not a
    SAGA iterate of the full problem -- the work per iteration is.

    x: scipy.sparse CSC of shape (p, n) (column i = sample i) or dense ndarray (p, n)
       in Fortran order (sample i contiguous).
    y: (Ky, n) Fortran-ordered response.
    state: dict with w (K,p) F-order, intercept (K), g_memory (K,n) F, g_sum (K,p) F,
       g_sum_intercept (K); updated in place.
    Returns (epochs, return_code, losses).
    """
    import scipy.sparse as sp

    L = lib()
    K = state["w"].shape[0]
    sparse = sp.issparse(x)
    p, n = x.shape
    if epoch_len is not None:
        assert stream is not None and not batch
    P = _SagaParams(FAMILIES[family], PENALTIES[penalty], K, n if epoch_len is None else int(epoch_len), p,
                    int(fit_intercept), int(standardize), gamma, alpha, beta, max_iter, tol, int(debug), n_total,
                    int(dense_intercept))
    y = np.asfortranarray(y, dtype=np.float64)
    if y.ndim == 1:
        y = y.reshape(1, -1)
    Ky = y.shape[0]
    for k in ("w", "g_memory", "g_sum"):
        assert state[k].flags.f_contiguous and state[k].dtype == np.float64
    draws, _keep = _mk_draws(stream, rng)
    rc = C.c_uint(0)
    losses = np.zeros(max_iter)
    if sparse:
        x = x.tocsc()
        ptr = np.ascontiguousarray(x.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(x.indices, dtype=np.int32)
        val = np.ascontiguousarray(x.data, dtype=np.float64)
        c = np.zeros(p) if x_center_scaled is None else np.ascontiguousarray(x_center_scaled)
        if batch and batch >= 1:                                   # batch = 1 is the batched iteration with one draw per batch, not the exact one
            ep = L.orc_saga_sparse_batched(
                C.byref(P), C.c_int64(batch), ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(val), _dp(c), _dp(y), Ky,
                _dp(state["intercept"]), _dp(state["w"]), _dp(state["g_memory"]),
                _dp(state["g_sum"]), _dp(state["g_sum_intercept"]), C.byref(draws),
                C.byref(rc), _dp(losses))
        else:
            ep = L.orc_saga_sparse(
                C.byref(P), ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(val), _dp(c), _dp(y), Ky,
                _dp(state["intercept"]), _dp(state["w"]), _dp(state["g_memory"]),
                _dp(state["g_sum"]), _dp(state["g_sum_intercept"]), C.byref(draws),
                C.byref(rc), _dp(losses))
    else:
        xd = np.asfortranarray(x, dtype=np.float64)
        ep = L.orc_saga_dense(
            C.byref(P), _dp(xd), _dp(y), Ky, _dp(state["intercept"]), _dp(state["w"]),
            _dp(state["g_memory"]), _dp(state["g_sum"]), _dp(state["g_sum_intercept"]),
            C.byref(draws), C.byref(rc), _dp(losses))
    return ep, rc.value, losses[:ep] if debug else None


def new_state(K, p, n):
    return dict(w=np.zeros((K, p), order="F"), intercept=np.zeros(K),
                g_memory=np.zeros((K, n), order="F"), g_sum=np.zeros((K, p), order="F"),
                g_sum_intercept=np.zeros(K))


def fit(x, y, *, family="gaussian", alpha=1.0, nlambda=100, lambda_min_ratio=None, lambda_=None,
        maxit=1000, standardize=True, intercept=True, thresh=1e-3, standardize_response=False,
        n_classes=None, seed=None, stream=None, rng=None, batch=0, debug=False):
    """Oracle counterpart of the reference's SgdnetDense/SgdnetSparse (src/sgdnet.cpp:358-375).

    x: (n, p) dense ndarray or scipy sparse; y: (n,) or (n, Ky) already encoded the way
    R/sgdnet.R:277-339 encodes it (binomial {0,1}, multinomial 0..K-1).
    """
    import scipy.sparse as sp

    L = lib()
    n, p = x.shape
    y = np.asarray(y, dtype=np.float64)
    y2 = np.asfortranarray(y.reshape(n, -1))
    Ky = y2.shape[1]
    if n_classes is None:
        n_classes = {"gaussian": 1, "binomial": 1, "mgaussian": Ky}.get(family)
        if n_classes is None:
            n_classes = int(y.max()) + 1
    K = n_classes
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n < p else 1e-4
    lam = None
    if lambda_ is not None:
        lam = np.ascontiguousarray(np.atleast_1d(lambda_), dtype=np.float64)
        nlambda = lam.size
    ctl = _Control(FAMILIES[family], alpha, int(intercept), int(standardize),
                   int(standardize_response), K, nlambda, _dp(lam) if lam is not None else None,
                   lambda_min_ratio, maxit, thresh, int(debug), batch)
    a0 = np.zeros((K, nlambda), order="F")
    beta = np.zeros((K, p, nlambda), order="F")
    lam_out = np.zeros(nlambda)
    dev = np.zeros(nlambda)
    rcodes = np.zeros(nlambda)
    step = np.zeros(nlambda)
    al2 = np.zeros(nlambda)
    bl1 = np.zeros(nlambda)
    losses = np.zeros((maxit, nlambda), order="F") if debug else None
    llen = np.zeros(nlambda, dtype=np.int32)
    res = _Result(_dp(a0), _dp(beta), _dp(lam_out), _dp(dev), _dp(rcodes),
                  _dp(losses) if debug else None, llen.ctypes.data_as(C.POINTER(C.c_int)), 0.0, 0.0,
                  _dp(step), _dp(al2), _dp(bl1))
    if stream is None and rng is None:
        rng = Rng(0 if seed is None else seed)
    draws, _keep = _mk_draws(stream, rng)
    if sp.issparse(x):
        xc = sp.csc_matrix(x)
        xc.sort_indices()
        colptr = np.ascontiguousarray(xc.indptr, dtype=np.int32)
        rowidx = np.ascontiguousarray(xc.indices, dtype=np.int32)
        val = np.ascontiguousarray(xc.data, dtype=np.float64)
        rc = L.orc_fit_sparse(C.c_int64(n), C.c_int64(p), colptr.ctypes.data_as(C.POINTER(C.c_int32)),
                              rowidx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(val), _dp(y2), Ky,
                              C.byref(ctl), C.byref(draws), C.byref(res))
    else:
        xd = np.asfortranarray(x, dtype=np.float64)
        rc = L.orc_fit_dense(C.c_int64(n), C.c_int64(p), _dp(xd), _dp(y2), Ky, C.byref(ctl),
                             C.byref(draws), C.byref(res))
    assert rc == 0
    out = dict(a0=a0, beta=beta, **{"lambda": lam_out}, dev_ratio=dev, return_codes=rcodes,
               nulldev=res.nulldev, npasses=res.npasses, step_size=step, alpha_l2=al2,
               beta_l1=bl1, draws_used=draws.pos)
    if debug:
        out["losses"] = [losses[:llen[i], i].copy() for i in range(nlambda)]
    return out


def _params(state, x, *, family, penalty, gamma, alpha, beta, fit_intercept, standardize, n_total, dense_intercept=False):
    K = state["w"].shape[0]
    p, n = x.shape
    return _SagaParams(FAMILIES[family], PENALTIES[penalty], K, n, p, int(fit_intercept),
                       int(standardize), gamma, alpha, beta, 1, 0.0, 0, n_total, int(dense_intercept))


def batch_gather(x, y, state, draws, D, d0, *, family, penalty, gamma, alpha, beta,
                 fit_intercept=True, x_center_scaled=None, n_total=0, dense_intercept=False):
    """Gather half of one batch (orc_batch_gather): adds into D (K,p) F-order and d0 (K)."""
    x = x.tocsc()
    P = _params(state, x, family=family, penalty=penalty, gamma=gamma, alpha=alpha, beta=beta,
                fit_intercept=fit_intercept, standardize=x_center_scaled is not None, n_total=n_total,
                dense_intercept=dense_intercept)
    ptr = np.ascontiguousarray(x.indptr, dtype=np.int64)
    idx = np.ascontiguousarray(x.indices, dtype=np.int32)
    val = np.ascontiguousarray(x.data, dtype=np.float64)
    y = np.asfortranarray(y, dtype=np.float64)
    dr = np.ascontiguousarray(draws, dtype=np.uint32)
    c = None if x_center_scaled is None else np.ascontiguousarray(x_center_scaled, dtype=np.float64)
    lib().orc_batch_gather(C.byref(P), ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                           idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(val),
                           _dp(c) if c is not None else None, _dp(y), y.shape[0],
                           _dp(state["intercept"]), _dp(state["w"]), _dp(state["g_memory"]),
                           dr.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(dr.size), _dp(D), _dp(d0))


def batch_sweep(x_shape, state, m_global, D, d0, *, family, penalty, gamma, alpha, beta,
                fit_intercept=True, x_center_scaled=None, n_total=0, dense_intercept=False):
    """Sweep half of one batch (orc_batch_sweep): consumes and zeroes D, d0."""
    K = state["w"].shape[0]
    p, n = x_shape
    P = _SagaParams(FAMILIES[family], PENALTIES[penalty], K, n, p, int(fit_intercept),
                    int(x_center_scaled is not None), gamma, alpha, beta, 1, 0.0, 0, n_total, int(dense_intercept))
    c = None if x_center_scaled is None else np.ascontiguousarray(x_center_scaled, dtype=np.float64)
    lib().orc_batch_sweep(C.byref(P), C.c_int64(m_global), _dp(c) if c is not None else None, _dp(D),
                          _dp(d0), _dp(state["intercept"]), _dp(state["w"]), _dp(state["g_sum"]),
                          _dp(state["g_sum_intercept"]))
