"""The reference's front-end tests that reach the fit path through predict / score / cv_sgdnet
(tests/testthat/test-assertions.R, test-cross-validation.R, test-multinomial.R,
test-predictions.R), restated against the host mirrors with the HIP backend doing the fits.
Data sets that are base-R objects in the reference (trees, mtcars) are replaced by the
committed abalone / iris fixtures and seeded synthetic data."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def sa():
    import torch  # noqa: F401  -- before libsgdnet_hip.so (sgdnet_amd/_lib.py)
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the backend has no CPU fallback")
    return sgdnet_amd


def _random_data(rng, n, p, family):
    """tests/testthat/setup.R random_data(): dense design, family-appropriate response."""
    x = rng.standard_normal((n, p))
    if family == "gaussian":
        return x, x @ rng.standard_normal(p) + rng.standard_normal(n)
    if family == "binomial":
        return x, (rng.random(n) < 1 / (1 + np.exp(-x @ rng.standard_normal(p)))).astype(int)
    if family == "multinomial":
        return x, np.argmax(x @ rng.standard_normal((p, 3)) + rng.gumbel(size=(n, 3)), axis=1)
    return x, x @ rng.standard_normal((p, 2)) + rng.standard_normal((n, 2))


def test_assertions_in_predict(sa):
    # test-assertions.R:3-17
    ab = np.load(os.path.join(GOLD, "abalone.npz"))
    x, y = ab["x"][:300, 1:4], ab["y"][:300]
    fit = sa.sgdnet(x, y, nlambda=5)
    for bad in (dict(), dict(newx=x, type=1), dict(newx=x, s=-3), dict(newx=x, type="class")):
        with pytest.raises(ValueError):
            sa.predict(fit, **bad)
    ir = np.load(os.path.join(GOLD, "iris.npz"))
    fit = sa.sgdnet(ir["x"], ir["y"], family="multinomial", nlambda=5)
    with pytest.raises(ValueError):
        sa.predict(fit, ir["x"], s=-3, type="class")
    with pytest.raises(ValueError):
        sa.predict(fit)


def test_assertions_in_cv_sgdnet(sa):
    # test-assertions.R:54-63
    rng = np.random.default_rng(1)
    n = 100
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    for bad in (dict(nfolds=n + 1), dict(nfolds=1), dict(alpha=[]), dict(foldid=np.arange(n - 1))):
        with pytest.raises(ValueError):
            sa.cv_sgdnet(x, y, **bad)


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
def test_cross_validation_works_for_every_family_and_measure(sa, family):
    # test-cross-validation.R:3-36
    from sgdnet_amd.score import _MEASURES
    rng = np.random.default_rng(2)
    x, y = _random_data(rng, 500, 2, family)
    for alpha in (0, 1, [0.2, 0.5]):
        for measure in _MEASURES[family]:
            cv = sa.cv_sgdnet(x, y, family=family, alpha=alpha, nlambda=5, maxit=10, thresh=1e-2,
                              type_measure=measure, seed=2)
            assert isinstance(cv, sa.CvSgdnet) and np.all(np.isfinite(cv.cv_summary[:, 2]))
            pred = sa.predict(cv.fit, x, s=cv.lambda_1se)          # predict.cv_sgdnet
            assert pred.shape[0] == 500 and np.all(np.isfinite(pred))


def test_various_cross_validation_arguments(sa):
    # test-cross-validation.R:38-46: leave-one-out and user fold ids
    rng = np.random.default_rng(3)
    n = 100
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    cv = sa.cv_sgdnet(x, y, nfolds=n, nlambda=4, seed=1)
    assert cv.cv_raw[0].shape == (n, 4)
    folds = sa.RRng(1).sample(n) % 10 + 1                          # sample(rep(1:10, 10))
    cv = sa.cv_sgdnet(x, y, foldid=folds, nlambda=4)
    assert cv.cv_raw[0].shape == (10, 4)


def test_multinomial_predicted_responses_sum_to_one(sa):
    # test-multinomial.R:8-13
    ir = np.load(os.path.join(GOLD, "iris.npz"))
    fit = sa.sgdnet(ir["x"], ir["y"], alpha=0, family="multinomial", nlambda=10)
    pr = sa.predict(fit, ir["x"], type="response")
    assert pr.shape == (150, 3, 10) and np.allclose(pr.sum(axis=1), 1.0, atol=1e-12)


def test_linear_interpolation_succeeds(sa):
    # test-predictions.R:75-92
    ab = np.load(os.path.join(GOLD, "abalone.npz"))
    x, y = ab["x"][:, 1:4], ab["y"]
    fit = sa.sgdnet(x, y, thresh=1e-7)
    s = float(np.sqrt(fit.lambda_[20] * fit.lambda_[21]))           # between two path points
    old = sa.predict(fit, x, s=s, type="coefficients")
    new = sa.predict(sa.sgdnet(x, y, lambda_=[s], thresh=1e-7), type="coefficients")
    assert np.abs(old - new).mean() / np.abs(new).mean() < 0.01    # expect_equivalent(tolerance = 0.01)
    lo = sa.predict(fit, x, s=fit.lambda_[21], type="coefficients")
    hi = sa.predict(fit, x, s=fit.lambda_[20], type="coefficients")
    assert np.all(np.minimum(lo, hi) - 1e-12 <= old) and np.all(old <= np.maximum(lo, hi) + 1e-12)


def test_nas_in_new_data_propagate(sa):
    # test-predictions.R:109-125
    rng = np.random.default_rng(5)
    x, y = _random_data(rng, 60, 2, "mgaussian")
    fit = sa.sgdnet(x, y, family="mgaussian", nlambda=5)
    x[10, :] = np.nan
    out = sa.predict(fit, x)
    assert np.all(np.isnan(out[10])) and np.all(np.isfinite(np.delete(out, 10, axis=0)))


@pytest.mark.gpu
@pytest.mark.parametrize("family,sparse", [("gaussian", True), ("gaussian", False), ("binomial", True),
                                           ("binomial", False), ("multinomial", True), ("mgaussian", False)])
def test_device_predict_and_score_match_the_host_mirror(sa, family, sparse):
    # score.hip: linear predictors + every measure of R/score.R on the GPU
    # against the numpy mirror of the R code, on held-out rows, for the whole lambda path
    import scipy.sparse as sp
    from sgdnet_amd.score import _MEASURES
    rng = np.random.default_rng(12)
    n, p, K = 700, 40, 3
    X = rng.standard_normal((n, p)) * (rng.random((n, p)) < 0.3)
    B = rng.standard_normal((p, K)) * (rng.random((p, K)) < 0.4)
    eta = X @ B
    if family == "gaussian":
        y = eta[:, 0] + 0.3 * rng.standard_normal(n)
    elif family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-eta[:, 0]))).astype(int)
    elif family == "multinomial":
        y = np.argmax(eta + rng.gumbel(size=eta.shape), axis=1)
    else:
        y = eta + 0.3 * rng.standard_normal(eta.shape)
    Xin = sp.csc_matrix(X) if sparse else X
    fit = sa.sgdnet(Xin[:500], y[:500], family=family, alpha=0.7, nlambda=30, thresh=1e-6, standardize=False)
    xt, yt = Xin[500:], y[500:]
    for typ in ("link", "response"):
        a = sa.predict(fit, xt, type=typ)
        b = sa.predict(fit, xt, type=typ, device=0)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-11, atol=1e-12)
    s = fit.lambda_[[0, 3, 7]] * 0.9                             # interpolated lambdas too
    for measure in _MEASURES[family]:
        for sel in (None, s):
            host = sa.score(fit, xt, yt, measure, s=sel)
            dev = sa.score(fit, xt, yt, measure, s=sel, device=0)
            assert host.shape == dev.shape
            assert np.allclose(host, dev, rtol=1e-10, atol=1e-12), (measure, host, dev)
    with pytest.raises(ValueError):
        sa.score(fit, xt[:, :5], yt, device=0)


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", [False, True])
def test_device_auc_breaks_ties_like_the_reference(sa, sparse):
    # R/score.R auc(): equal probabilities are ordered by runif(2n) drawn per lambda.  Held-out rows with
    # many exact duplicates (and the intercept-only first lambda, where every probability ties): the device
    # sort reproduces the host mirror for a given draw, for one draw shared by all lambdas, and without one
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    n, p = 900, 15
    X = np.round(rng.standard_normal((n, p)) * (rng.random((n, p)) < 0.4), 1)
    X[600:] = X[rng.integers(500, 600, 300)]                       # duplicates among the held-out rows
    y = (rng.random(n) < 1 / (1 + np.exp(-X[:, 0] + X[:, 1]))).astype(int)
    Xin = sp.csc_matrix(X) if sparse else X
    fit = sa.sgdnet(Xin[:500], y[:500], family="binomial", alpha=0.8, nlambda=12, thresh=1e-6, standardize=False)
    xt, yt = Xin[500:], y[500:]
    m, L = yt.size, len(fit.lambda_)
    per_lambda = rng.random((2 * m, L))
    shared = rng.random(2 * m)
    for tb in (None, shared, per_lambda):
        host = sa.score(fit, xt, yt, "auc", tie_break=tb)
        dev = sa.score(fit, xt, yt, "auc", device=0, tie_break=tb)
        assert np.allclose(host, dev, rtol=1e-13, atol=0), (host, dev)
    assert not np.allclose(sa.score(fit, xt, yt, "auc", tie_break=shared), sa.score(fit, xt, yt, "auc"))   # ties matter here
    assert abs(host[0] - 0.5) < 0.15                                # intercept only: the area of a coin flip
    # round 3: the draws made on the device from an R generator (sgdnet_auc_*_rng) -- the same areas as with the
    # generator's draws made on the host, and the generator left in the same state
    r_host, r_dev = sa.RRng(5), sa.RRng(5)
    drawn = r_host.unif(2 * m * L).reshape(L, 2 * m).T
    want = sa.score(fit, xt, yt, "auc", device=0, tie_break=drawn)
    got = sa.score(fit, xt, yt, "auc", device=0, rng=r_dev)
    assert np.array_equal(got, want)
    assert r_dev.unif(3).tolist() == r_host.unif(3).tolist()
    r_h2 = sa.RRng(5)
    assert np.allclose(sa.score(fit, xt, yt, "auc", rng=r_h2), want, rtol=1e-13, atol=0)   # host mirror of the same protocol


@pytest.mark.gpu
def test_device_score_with_many_lambdas_and_classes_is_chunked(sa):
    # more (lambda, class) pairs than one launch holds: the C entry point walks the path in chunks
    rng = np.random.default_rng(13)
    n, p, K = 400, 12, 7
    X = rng.standard_normal((n, p))
    y = rng.integers(0, K, n)
    y[:K] = np.arange(K)
    fit = sa.sgdnet(X, y, family="multinomial", alpha=0.5, nlambda=20, thresh=1e-5, standardize=False)
    s = np.geomspace(fit.lambda_[0], fit.lambda_[-1], 400)       # 400 x 7 pairs
    host = sa.score(fit, X, y, "deviance", s=s)
    dev = sa.score(fit, X, y, "deviance", s=s, device=0)
    assert np.allclose(host, dev, rtol=1e-10)
    assert np.allclose(sa.predict(fit, X, s=s), sa.predict(fit, X, s=s, device=0), rtol=1e-11, atol=1e-12)


def test_small_correlated_data_in_batched_mode(sa):
    # abalone (4177 x 9, strongly collinear columns): the automatic mode must neither diverge nor
    # use virtual shards here, and a run forced onto shards recovers by restarting without them
    import os as _os
    ab = np.load(os.path.join(GOLD, "abalone.npz"))
    x, y = ab["x"], ab["y"]
    ref = sa.sgdnet(x, y, lambda_=[0.01], thresh=1e-9, maxit=20000, seed=1)
    for forced in (-1, 2, 4):
        with sa.option("virtual_shards", forced):
            fit = sa.sgdnet(x, y, lambda_=[0.01], thresh=1e-9, maxit=20000, seed=1, mode="batched")
        assert fit.return_codes[0] == 0 and np.all(np.isfinite(fit.beta))
        assert np.abs(fit.beta - ref.beta).max() < 1e-5 * max(1.0, np.abs(ref.beta).max())
    path = sa.sgdnet(x, y, mode="auto", seed=1)
    assert len(path.lambda_) == 100 and np.all(np.isfinite(path.beta)) and np.all(path.return_codes == 0)


@pytest.mark.parametrize("no_lmax", [False, True])
def test_nonnegative_sparse_features_keep_the_automatic_window_stable(sa, monkeypatch, no_lmax):
    # non-negative sparse data (counts, tf-idf; here uniform(0, 1) values) has a common component:
    # the largest eigenvalue of X'X/n is ~8x its diagonal and a window sized by the diagonal settles
    # into a bounded oscillation at small lambda -- finite numbers, deviance above the null
    # model's.  The driver sizes the window by the eigenvalue (device power iteration) and, as a
    # safety net, redoes a lambda with a shorter window when the deviance rises along the path.
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    n, p = 200_000, 2_000
    X = sp.random(n, p, density=0.005, format="csc", random_state=3)
    b = rng.standard_normal(p) * (rng.random(p) < 0.1)
    y = (rng.random(n) < 1 / (1 + np.exp(-np.asarray(X @ b).ravel()))).astype(int)
    with sa.option("window_eigenvalue", 0 if no_lmax else 1):      # 0: only the safety net
        fit = sa.sgdnet(X, y, family="binomial", alpha=1.0, nlambda=20, thresh=1e-5, standardize=False, mode="auto",
                        maxit=300, seed=3)
    assert np.all(fit.return_codes == 0)
    assert np.all(np.diff(fit.dev_ratio) > -1e-6) and fit.dev_ratio[-1] > 0.05
    ref = sa.sgdnet(X, y, family="binomial", alpha=1.0, lambda_=fit.lambda_, thresh=1e-5, standardize=False,
                    mode="batched", batch=500, maxit=300, seed=3)
    # (at lambda_max the solution is 0 and what is left of a few flickering coefficients at thresh = 1e-5
    # depends on the draws: 1e-4 there, 2e-5 below)
    assert abs(fit.dev_ratio[0] - ref.dev_ratio[0]) <= 1e-4
    assert np.allclose(fit.dev_ratio[1:], ref.dev_ratio[1:], atol=2e-5)


def test_increasing_lambda_sequence_is_caught_by_the_null_model_net(sa, monkeypatch):
    """A user-supplied lambda sequence that is not decreasing defeats the "deviance can only fall along the
    path" safety net; a fit that is worse than the null model is still redone with a shorter window."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    n, p = 200_000, 2_000
    X = sp.random(n, p, density=0.005, format="csc", random_state=3)
    b = rng.standard_normal(p) * (rng.random(p) < 0.1)
    y = (rng.random(n) < 1 / (1 + np.exp(-np.asarray(X @ b).ravel()))).astype(int)
    lam = np.geomspace(2e-6, 2e-4, 6)                              # increasing: smallest (hardest) lambda first
    with sa.option("window_eigenvalue", 0):                        # window from the diagonal bound: ~8x too long
        fit = sa.sgdnet(X, y, family="binomial", alpha=1.0, lambda_=lam, thresh=1e-5, standardize=False, mode="auto",
                        maxit=300, seed=3)
    ref = sa.sgdnet(X, y, family="binomial", alpha=1.0, lambda_=lam, thresh=1e-5, standardize=False, mode="batched",
                    batch=500, maxit=300, seed=3)
    assert np.all(fit.dev_ratio > 0.0) and np.all(np.isfinite(fit.beta))
    assert np.allclose(fit.dev_ratio, ref.dev_ratio, atol=1e-4)


@pytest.mark.parametrize("family,mode", [("binomial", "exact"), ("gaussian", "batched")])
def test_large_dense_x_is_prepared_on_the_device(sa, monkeypatch, family, mode):
    """Dense matrices of >= 4M elements are standardised, multiplied for lambda_max, transposed and normed
    on the device (dense_setup_*); the host loops (option host_setup = 1, the path small
    matrices keep because their sums run in the reference's order) must give the same fit."""
    rng = np.random.default_rng(5)
    n, p = 4100, 1000
    x = rng.standard_normal((n, p)) * np.linspace(0.5, 3.0, p) + np.linspace(-1, 1, p)
    b = rng.standard_normal(p) * (rng.random(p) < 0.05)
    lp = (x - x.mean(0)) / x.std(0) @ b
    y = (rng.random(n) < 1 / (1 + np.exp(-lp))).astype(int) if family == "binomial" else lp + rng.standard_normal(n)
    kw = dict(family=family, alpha=0.5, nlambda=4, lambda_min_ratio=0.2, thresh=1e-7, maxit=400, seed=2, mode=mode)
    if mode == "exact":                       # one wavefront, p = 1000: keep the test short
        kw.update(nlambda=2, lambda_min_ratio=0.5, thresh=1e-4, maxit=25)
    dev = sa.sgdnet(x, y, **kw)
    with sa.option("host_setup", 1):
        host = sa.sgdnet(x, y, **kw)
    assert np.allclose(dev.lambda_, host.lambda_, rtol=1e-11) and dev.nulldev == pytest.approx(host.nulldev, rel=1e-12)
    assert np.array_equal(dev.return_codes, host.return_codes) and dev.npasses == host.npasses
    assert np.abs(dev.beta - host.beta).max() <= 1e-6 * np.abs(host.beta).max()
    assert np.allclose(dev.dev_ratio, host.dev_ratio, atol=1e-7)


def test_multinomial_fit_on_virtual_shards_reaches_the_unsharded_optimum(sa):
    """Round 3: the fit driver runs 2..4-class fits of sparse x on virtual shards too (250 000 x 300, K = 3: four
    replicas).  Same path, same optimum as the unsharded batched fit at a tight thresh."""
    from sgdnet_amd import data as D
    n, p, K = 250_000, 300, 3
    pr = D.make_sparse_glm(n, p, 0.03, family="multinomial", n_classes=K, seed=17)
    X = D.as_scipy(pr).T.tocsc()
    y = pr["y"].ravel()
    kw = dict(family="multinomial", alpha=0.5, nlambda=5, lambda_min_ratio=0.01, standardize=False, thresh=1e-8,
              maxit=400, seed=3, mode="batched")
    with sa.option("virtual_shards", 0):
        ref = sa.sgdnet(X, y, **kw)
    fit = sa.sgdnet(X, y, **kw)                                   # the driver's rule: 4 shards
    assert np.all(np.asarray(fit.return_codes) == 0) and np.all(np.asarray(ref.return_codes) == 0)
    assert np.allclose(fit.lambda_, ref.lambda_, rtol=1e-12)
    fb, rb = np.stack(fit.beta), np.stack(ref.beta)
    assert np.abs(fb - rb).max() <= 2e-6 * np.abs(rb).max()
    assert np.allclose(fit.dev_ratio, ref.dev_ratio, atol=1e-7)
