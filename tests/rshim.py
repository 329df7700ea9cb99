"""Python handle on the shim compiled against the R-API mock (tests/rmock; TEST INFRASTRUCTURE).

Builds the objects R would pass to `.Call(_sgdnet_SgdnetDense|Sparse, x, y, control)` -- the
control list of R/sgdnet.R:346-359 with R's own types, a numeric matrix or the slots of a
dgCMatrix -- calls the real shim code (shim/sgdnet_shim.c) and decodes the returned R list."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SGDNET_SHIM_MOCK_SO: the sanitizer build of the same two sources (tests/test_shim_sanitizers.py)
SO = os.environ.get("SGDNET_SHIM_MOCK_SO") or os.path.join(HERE, "rmock", "libsgdnet_shim_mock.so")
REALSXP, INTSXP, LGLSXP, STRSXP, VECSXP = 14, 13, 10, 16, 19

_L = None


def lib():
    global _L
    if _L is None:
        L = C.CDLL(SO)
        vp = C.c_void_p
        for name, res, args in [
            ("rmock_real", vp, [C.POINTER(C.c_double), C.c_ssize_t]),
            ("rmock_int", vp, [C.POINTER(C.c_int), C.c_ssize_t, C.c_int]),
            ("rmock_str", vp, [C.c_char_p]), ("rmock_list", vp, [C.c_ssize_t]),
            ("rmock_list_set", None, [vp, C.c_ssize_t, C.c_char_p, vp]),
            ("rmock_set_dim", None, [vp, C.c_int, C.c_int]), ("rmock_s4", vp, []),
            ("rmock_set_slot", None, [vp, C.c_char_p, vp]), ("rmock_set_option", None, [C.c_char_p, vp]),
            ("rmock_call3", vp, [vp, vp, vp, vp]), ("rmock_last_error", C.c_char_p, []),
            ("rmock_typeof", C.c_int, [vp]), ("rmock_length", C.c_longlong, [vp]),
            ("rmock_real_ptr", C.POINTER(C.c_double), [vp]), ("rmock_elt", vp, [vp, C.c_longlong]),
            ("rmock_name", C.c_char_p, [vp, C.c_longlong]), ("rmock_dim", C.c_int, [vp, C.c_int]),
            ("rmock_set_seed", None, [C.c_uint32]), ("rmock_unif_count", C.c_longlong, []),
            ("rmock_rng_scope_calls", C.c_int, []), ("rmock_protect_depth", C.c_int, []),
            ("rmock_reset", None, []), ("rmock_registered", C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
            ("rmock_registered_fn", vp, [C.c_char_p]), ("rmock_dynamic_symbols", C.c_int, []),
            ("rmock_rng_state", None, [C.POINTER(C.c_uint32)]), ("R_init_sgdnet", None, [vp]),
            ("unif_rand", C.c_double, []), ("GetRNGstate", None, []), ("PutRNGstate", None, []),
            ("rmock_set_rng_kind", None, [C.c_int]),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _L = L
    return _L


class RError(RuntimeError):
    pass


def r_real(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))
    return lib().rmock_real(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)


def r_int(a, logical=False):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel(order="F"))
    return lib().rmock_int(a.ctypes.data_as(C.POINTER(C.c_int)), a.size, int(logical))


def r_matrix(a, integer=False):
    a = np.asarray(a)
    x = r_int(a) if integer else r_real(a)
    lib().rmock_set_dim(x, a.shape[0], a.shape[1])
    return x


def r_dgcmatrix(m):
    """S4 object with the slots the shim reads (Matrix::dgCMatrix): i, p, x, Dim."""
    import scipy.sparse as sp
    m = sp.csc_matrix(m, dtype=np.float64)
    m.sort_indices()
    L = lib()
    obj = L.rmock_s4()
    L.rmock_set_slot(obj, b"i", r_int(m.indices))
    L.rmock_set_slot(obj, b"p", r_int(m.indptr))
    L.rmock_set_slot(obj, b"x", r_real(m.data))
    L.rmock_set_slot(obj, b"Dim", r_int(m.shape))
    return obj


def control_list(*, family, alpha=1.0, n_classes=1, nlambda=100, lambda_=None, lambda_min_ratio=1e-4, maxit=1000,
                 standardize=True, intercept=True, thresh=1e-3, standardize_response=False, is_sparse=False,
                 debug=False, drop=None):
    """list(debug=, elasticnet_mix=, family=, intercept=, is_sparse=, lambda=, lambda_min_ratio=, max_iter=,
    n_lambda=, n_classes=, standardize=, standardize_response=, tol=, type_multinomial=) with the types
    sgdnet.default() produces: logicals, doubles (maxit = 1000 and nlambda = 100 are doubles in R),
    n_classes from length()/ncol() = integer, strings."""
    L = lib()
    lam = np.zeros(0) if lambda_ is None else np.atleast_1d(np.asarray(lambda_, dtype=np.float64))
    fields = [
        ("debug", r_int([debug], logical=True)), ("elasticnet_mix", r_real([alpha])), ("family", L.rmock_str(family.encode())),
        ("intercept", r_int([intercept], logical=True)), ("is_sparse", r_int([is_sparse], logical=True)),
        ("lambda", r_real(lam)), ("lambda_min_ratio", r_real([lambda_min_ratio])), ("max_iter", r_real([maxit])),
        ("n_lambda", r_real([lam.size if lam.size else nlambda])), ("n_classes", r_int([n_classes])),
        ("standardize", r_int([standardize], logical=True)),
        ("standardize_response", r_int([standardize_response], logical=True)), ("tol", r_real([thresh])),
        ("type_multinomial", L.rmock_str(b"ungrouped")),
    ]
    fields = [f for f in fields if f[0] != drop]
    lst = L.rmock_list(len(fields))
    for i, (k, v) in enumerate(fields):
        L.rmock_list_set(lst, i, k.encode(), v)
    return lst


def set_option(name, value):
    L = lib()
    v = L.rmock_str(value.encode()) if isinstance(value, str) else r_real([value])
    L.rmock_set_option(name.encode(), v)


def call(symbol, x, y, control):
    """.Call(symbol, x, y, control) through the routine table R_init_sgdnet registered."""
    L = lib()
    fn = L.rmock_registered_fn(symbol.encode())
    if not fn:
        raise KeyError(f"{symbol} is not a registered .Call routine")
    out = L.rmock_call3(fn, x, y, control)
    if not out:
        raise RError(L.rmock_last_error().decode())
    return out


def _real(x):
    L = lib()
    n = L.rmock_length(x)
    assert L.rmock_typeof(x) == REALSXP
    return np.ctypeslib.as_array(L.rmock_real_ptr(x), shape=(n,)).copy() if n else np.zeros(0)


def decode_result(res):
    """The named list of src/sgdnet.cpp:275-284 -> dict of numpy arrays, plus what R code would see
    of its structure: `names` in order, and unlist(beta) exactly as R/sgdnet.R:377 consumes it."""
    L = lib()
    assert L.rmock_typeof(res) == VECSXP
    names = [L.rmock_name(res, i).decode() for i in range(L.rmock_length(res))]
    el = {nm: L.rmock_elt(res, i) for i, nm in enumerate(names)}
    nl = L.rmock_length(el["a0"])
    a0 = [_real(L.rmock_elt(el["a0"], i)) for i in range(nl)]
    beta_dims = [(L.rmock_dim(L.rmock_elt(el["beta"], i), 0), L.rmock_dim(L.rmock_elt(el["beta"], i), 1))
                 for i in range(nl)]
    beta = [_real(L.rmock_elt(el["beta"], i)) for i in range(nl)]
    losses = [_real(L.rmock_elt(el["losses"], i)) for i in range(L.rmock_length(el["losses"]))]
    return dict(names=names, a0=np.array(a0).T, unlist_beta=np.concatenate(beta), beta_dims=beta_dims, losses=losses,
                npasses=float(_real(el["npasses"])[0]), nulldev=float(_real(el["nulldev"])[0]),
                dev_ratio=_real(el["dev.ratio"]), lambda_=_real(el["lambda"]), return_codes=_real(el["return_codes"]))
