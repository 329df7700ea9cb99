"""C-ABI boundary behaviour added in round 2: every source of the sample order reaches the
virtual-shard kernels in the layout they read, and indices that would be used as addresses are
rejected on the host (include/sgdnet_hip.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import torch  # noqa: F401
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the backend has no CPU fallback")
    return sgdnet_amd


@pytest.fixture(scope="module")
def big():
    from sgdnet_amd import data as D
    n, p = 200_000, 100                      # large enough for the fit driver to pick 8 virtual shards
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=21)
    return D.as_scipy(pr).T.tocsc(), pr["y"].ravel()


def test_virtual_shards_through_the_unif_callback_equal_the_builtin_generator(sa, big, monkeypatch):
    """The R shim hands the backend unif_rand() (shim/sgdnet_shim.c); with virtual shards the
    driver must lay those draws out per shard exactly as the device generator does."""
    x, y = big
    kw = dict(family="binomial", alpha=0.5, lambda_=[2e-3, 1e-3], standardize=False, thresh=1e-7, maxit=60,
              mode="batched")
    r = sa.RRng(3)
    with sa.option("rng_generators", 1):                   # one R stream on the device as well
        ref = sa.sgdnet(x, y, seed=3, **kw)
        cb = sa.sgdnet(x, y, unif=lambda: float(r.unif()[0]), **kw)
    assert cb.npasses == ref.npasses
    assert np.abs(cb.beta - ref.beta).max() <= 1e-12 * np.abs(ref.beta).max()
    assert np.abs(cb.a0 - ref.a0).max() <= 1e-12


def test_explicit_sample_stream_runs_unsharded_and_reaches_the_same_optimum(sa, big):
    x, y = big
    n = x.shape[0]
    kw = dict(family="binomial", alpha=0.5, lambda_=[1e-3], standardize=False, thresh=1e-9, maxit=200, mode="batched")
    ref = sa.sgdnet(x, y, seed=3, **kw)
    stream = sa.RRng(11).stream(n, n * 200)               # indices over the whole data set
    ex = sa.sgdnet(x, y, sample_stream=stream, **kw)
    assert ex.return_codes[0] == 0 and ref.return_codes[0] == 0
    assert np.abs(ex.beta - ref.beta).max() <= 1e-6 * np.abs(ref.beta).max()


def test_out_of_range_indices_are_rejected_before_they_reach_the_device(sa, big):
    x, y = big
    n = x.shape[0]
    bad = np.full(n, n, dtype=np.uint32)                   # n is not a sample index
    with pytest.raises(Exception, match="not a sample index"):
        sa.sgdnet(x, y, family="binomial", lambda_=[1e-3], sample_stream=bad, maxit=1)
    S = sa.SagaSolver(x.T.tocsc(), y.reshape(1, -1), family="binomial", n_classes=1)
    with pytest.raises(Exception, match="not a sample index"):
        S.upload_stream(bad)
    S.close()
    # class codes index arrays on host and device: the C ABI checks them itself
    import ctypes as C
    from sgdnet_amd import _lib
    L = sa.load()
    xs = x[:1000].tocsc()
    ctl = _lib.Control()
    ctl.family, ctl.n_classes, ctl.n_lambda, ctl.max_iter, ctl.tol = _lib.FAMILIES["multinomial"], 3, 2, 5, 1e-3
    ctl.elasticnet_mix, ctl.lambda_min_ratio, ctl.intercept = 1.0, 1e-2, 1
    yy = np.asfortranarray(np.full((1000, 1), 3.0))        # classes are 0..2
    res = _lib.Result()
    bufs = [np.zeros(3 * 2), np.zeros(3 * xs.shape[1] * 2), np.zeros(2), np.zeros(2), np.zeros(2)]
    res.a0, res.beta, res.lambda_, res.dev_ratio, res.return_codes = (_lib.dptr(b) for b in bufs)
    csc = _lib.Csc()
    cp, ri, va = xs.indptr.astype(np.int32), xs.indices.astype(np.int32), xs.data.astype(np.float64)
    csc.n_rows, csc.n_cols = xs.shape
    csc.colptr, csc.rowidx = cp.ctypes.data_as(C.POINTER(C.c_int32)), ri.ctypes.data_as(C.POINTER(C.c_int32))
    csc.values = _lib.dptr(va)
    assert L.sgdnet_fit_sparse(C.byref(csc), _lib.dptr(yy), 1, C.byref(ctl), C.byref(res)) == -1  # SGDNET_EINVAL
    assert b"class code" in L.sgdnet_last_error()
    cp[5] = cp[6] + 1                                       # colptr going backwards
    yy[:] = 1.0
    assert L.sgdnet_fit_sparse(C.byref(csc), _lib.dptr(yy), 1, C.byref(ctl), C.byref(res)) == -1  # SGDNET_EINVAL
    assert b"non-decreasing" in L.sgdnet_last_error()


def test_exact_mode_does_not_report_a_blown_up_run_as_converged(sa):
    """fmax() drops NaN: a non-finite coefficient must keep ConvergenceCheck false
    (src/utils.h:240-262 compares with NaN, which is false) so the fit ends with return_code 1."""
    rng = np.random.default_rng(0)
    n, p = 300, 5
    x = sp_csc(rng.standard_normal((p, n)))
    y = rng.standard_normal((1, n))
    S = sa.SagaSolver(x, y, family="gaussian", n_classes=1)
    S.set_penalty("ridge", 10.0, 1e-3, 0.0)                # a step size far beyond 1/L: diverges
    S.upload_stream(sa.RRng(1).stream(n, n * 40))
    ep, conv = S.run(mode="exact", max_epochs=40, tol=1e-3)
    w = S.get("w")
    S.close()
    assert not np.isfinite(w).all()
    assert not conv and ep == 40


def sp_csc(a):
    import scipy.sparse as sp
    return sp.csc_matrix(a)


def test_parallel_generators_stay_on_rs_single_stream(sa, big, monkeypatch):
    """The fit driver runs 8-32 Mersenne-Twisters side by side for large batched fits; with the
    jump-ahead of sgdnet_amd/csrc/mt_jump.cpp they all sit on the ONE stream set.seed() defines, so
    the fit is the single-generator fit and R's generator ends where the reference leaves it."""
    import ctypes as C
    x, y = big
    n = x.shape[0]
    kw = dict(family="binomial", alpha=0.5, lambda_=[2e-3, 1e-3], standardize=False, thresh=1e-7, maxit=60,
              mode="batched")
    r_par = sa.RRng(3)
    par = sa.sgdnet(x, y, rng=r_par, **kw)                     # default: several generators
    r_one = sa.RRng(3)
    with sa.option("rng_generators", 1):
        one = sa.sgdnet(x, y, rng=r_one, **kw)
    assert par.npasses == one.npasses
    assert np.abs(par.beta - one.beta).max() <= 1e-12 * np.abs(one.beta).max()
    assert par.draws_used == one.draws_used and par.draws_used >= int(par.npasses) * n
    for r, fit in ((r_par, par), (r_one, one)):
        host = sa.RRng(3)
        host.stream(n, fit.draws_used)                         # every draw the fit took, one by one
        assert np.array_equal(host.unif(64), r.unif(64))       # same position on the same stream


def test_device_jump_equals_the_sequential_generator(sa):
    """sgdnet_rng_jump_poly + the host jump against plain stepping (the device kernel is covered by
    the fit above): the state J draws ahead produces the draws the stepped generator produces."""
    import ctypes as C
    from sgdnet_amd import _lib
    L = sa.load()
    for J in (1, 624, 99_991, 10_000_000):
        poly = (C.c_uint32 * 624)()
        assert L.sgdnet_rng_jump_poly(C.c_uint64(J), poly) == 0
        a, b = sa.RRng(11), sa.RRng(11)
        a.unif(1000), b.unif(1000)
        out = sa.RRng(0)
        L.sgdnet_rng_jump(C.byref(a.state), poly, C.byref(out.state))
        b.stream(7, J)
        assert np.array_equal(out.unif(1500), b.unif(1500))
