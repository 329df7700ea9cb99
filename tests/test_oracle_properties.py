"""Pins the CPU oracle against the reference's own known-answer tests.

The reference stores no golden vectors; its testthat files assert closed forms and
external solvers.  Each test below restates one of them with numpy in
place of R (lm -> lstsq, glm -> IRLS, glmnet comparisons -> KKT conditions of the
elastic-net optimum) and checks the oracle at the tolerance the reference states.
"""
import numpy as np
import pytest
import scipy.sparse as sp


def sd2(v, axis=0):
    return np.sqrt(((v - v.mean(axis=axis, keepdims=True)) ** 2).mean(axis=axis))


def test_r_mersenne_twister_known_values(oracle):
    # R: set.seed(s); runif(3)
    known = {1: [0.2655087, 0.3721239, 0.5728534],
             42: [0.9148060, 0.9370754, 0.2861395],
             123: [0.2875775, 0.7883051, 0.4089769]}
    for seed, vals in known.items():
        np.testing.assert_allclose(oracle.Rng(seed).unif(3), vals, atol=5e-8)
    # floor(runif(0, n)) stays inside [0, n)
    s = oracle.Rng(7).stream(10, 10000)
    assert s.min() == 0 and s.max() == 9


def test_ols_at_lambda_zero(oracle):
    # tests/testthat/test-gaussian.R:3-15 (tolerance 1e-3)
    rng = np.random.default_rng(1)
    n = 111
    x = np.c_[rng.normal(10, 3.5, n), rng.normal(78, 9, n)]
    y = 2.0 - 3.0 * x[:, 0] + 1.8 * x[:, 1] + rng.normal(0, 20, n)
    fit = oracle.fit(x, y, family="gaussian", lambda_=[0.0], maxit=1000, thresh=1e-4, seed=1)
    ols = np.linalg.lstsq(np.c_[np.ones(n), x], y, rcond=None)[0]
    got = np.r_[fit["a0"][0, 0], fit["beta"][0, :, 0]]
    np.testing.assert_allclose(got, ols, rtol=1e-3, atol=1e-3)


def test_lambda_max_gives_null_model(oracle):
    # test-gaussian.R:17-36: max(lambda) == closed form, first fit all-zero
    from sklearn.datasets import load_iris
    iris = load_iris().data
    x, y = iris[:, :3], iris[:, 3]
    sy = sd2(y)
    xx = (x - x.mean(0)) / sd2(x)
    yy = (y - y.mean()) / sy
    lambda_max = np.abs(yy @ xx).max() * sy / x.shape[0]
    fit = oracle.fit(x, y, family="gaussian", maxit=1000, thresh=1e-4, seed=1)
    assert fit["lambda"].max() == pytest.approx(lambda_max, rel=1e-12)
    assert np.all(fit["beta"][0, :, 0] == 0.0)


def test_closed_form_ridge(oracle):
    # test-gaussian.R:38-60 (tolerance 1e-3)
    rng = np.random.default_rng(1)
    n, p = 500, 3
    b = np.array([-5.0, 3.0, 2.0])
    x = rng.standard_normal((n, p))
    x = (x - x.mean(0)) / x.std(0, ddof=1)
    y = x @ b + rng.standard_normal(n)
    lam = 0.01
    sd_y = sd2(y)
    theory = np.linalg.solve(x.T @ x + lam * np.eye(p), x.T @ y)
    fit = oracle.fit(x, y, family="gaussian", alpha=0.0, lambda_=[sd_y * lam / n], intercept=False,
                     thresh=1e-5, maxit=1000, seed=1)
    np.testing.assert_allclose(fit["beta"][0, :, 0], theory, rtol=1e-3)


def test_closed_form_multivariate_ridge(oracle):
    # test-mgaussian.R:3-28 (tolerance 1e-6)
    rng = np.random.default_rng(1)
    n, p = 500, 3
    b = np.array([[-5.0, 0.0], [3.0, -5.0], [2.0, 9.0]])
    x = rng.standard_normal((n, p))
    x = (x - x.mean(0)) / x.std(0, ddof=1)
    E = x @ b
    y = np.c_[rng.normal(E.mean(0)[0], 1, n), rng.normal(E.mean(0)[1], 1, n)]
    lam = 0.01
    sd_y = sd2(y)
    theory = np.linalg.solve(x.T @ x + lam * np.eye(p), x.T @ y)
    # testthat's expect_equivalent(tolerance) is all.equal's mean relative difference.  With
    # the reference's thresh = 1e-5 the stopping rule itself bounds the accuracy near 1e-5 on
    # this (numpy-drawn) data; at thresh = 1e-8 the 1e-6 of the reference's test is met.
    for thresh, tol in ((1e-5, 2e-5), (1e-8, 1e-6)):
        fit = oracle.fit(x, y, family="mgaussian", alpha=0.0, lambda_=[sd_y[0] * lam / n],
                         intercept=False, thresh=thresh, maxit=10000, seed=1)
        got = fit["beta"][:, :, 0].T          # (p, K)
        assert np.abs(got - theory).mean() / np.abs(theory).mean() < tol


def _irls_logistic(x, y, iters=50):
    A = np.c_[np.ones(len(y)), x]
    b = np.zeros(A.shape[1])
    for _ in range(iters):
        mu = 1 / (1 + np.exp(-(A @ b)))
        W = mu * (1 - mu)
        b = b + np.linalg.solve(A.T @ (A * W[:, None]), A.T @ (y - mu))
    return b


def test_unpenalised_logistic_is_the_mle(oracle):
    # test-binomial.R:3-14 (tolerance 1e-5 at thresh 1e-9)
    rng = np.random.default_rng(3)
    n = 60
    x = np.c_[rng.uniform(0.02, 1.1, n), rng.uniform(50, 200, n)]
    eta = -1.0 + 2.5 * x[:, 0] - 0.005 * x[:, 1]
    y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    fit = oracle.fit(x, y, family="binomial", lambda_=[0.0], thresh=1e-9, maxit=100000, seed=1)
    mle = _irls_logistic(x, y)
    got = np.r_[fit["a0"][0, 0], fit["beta"][0, :, 0]]
    np.testing.assert_allclose(got, mle, rtol=1e-5, atol=1e-5)


def test_constant_response(oracle):
    # test-gaussian.R:62-71
    from sklearn.datasets import load_iris
    x = load_iris().data
    y = np.full(x.shape[0], 5.0)
    fit = oracle.fit(x, y, family="gaussian", seed=1)
    assert np.all(fit["lambda"] == 0.0)
    assert np.all(fit["beta"] == 0.0)
    np.testing.assert_allclose(fit["a0"], 5.0)


def _manual_lambda_max(x, y, family, standardize, alpha):
    # test-lambda-path.R:52-100
    x2 = (x - x.mean(0)) / sd2(x) if standardize else x
    if family == "multinomial":
        levels = np.unique(y)
        y2 = (y[:, None] == levels[None, :]).astype(float)
    else:
        y2 = np.asarray(y, dtype=float).reshape(len(y), -1)
    ys = sd2(y2)
    y3 = (y2 - y2.mean(0)) / ys
    ip = (x2.T @ y3) * ys[None, :]
    if family == "multinomial":
        return np.abs(ip).max() / (x.shape[0] * max(alpha, 1e-3))
    return np.sqrt((ip ** 2).sum(1)).max() / (x.shape[0] * max(alpha, 1e-3))


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
@pytest.mark.parametrize("standardize", [True, False])
def test_lambda_max_manual(oracle, family, alpha, standardize):
    # test-lambda-path.R:49-129
    rng = np.random.default_rng(5)
    n = 32
    x = np.c_[rng.integers(4, 9, n), rng.normal(230, 120, n), rng.normal(146, 68, n),
              rng.integers(0, 2, n)].astype(float)
    y = {"gaussian": rng.normal(20, 6, n), "binomial": rng.integers(0, 2, n).astype(float),
         "multinomial": rng.integers(0, 3, n).astype(float),
         "mgaussian": np.c_[rng.normal(146, 68, n), rng.normal(3.6, 0.5, n)]}[family]
    fit = oracle.fit(x, y, family=family, alpha=alpha, standardize=standardize, nlambda=5,
                     thresh=1e-1, seed=1)
    assert fit["lambda"].max() == pytest.approx(_manual_lambda_max(x, y, family, standardize, alpha),
                                                rel=1e-10)


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
@pytest.mark.parametrize("intercept", [True, False])
def test_null_deviance_formulas(oracle, family, intercept):
    # test-deviance.R:8-96 manual formulas
    rng = np.random.default_rng(2)
    n = 100
    x = rng.standard_normal((n, 2))
    y = {"gaussian": rng.normal(10, 2, n), "binomial": (rng.random(n) < 0.8).astype(float),
         "multinomial": rng.binomial(2, 0.5, n).astype(float),
         "mgaussian": np.c_[rng.normal(100, 1, n), rng.normal(0, 1, n)]}[family]
    if family == "multinomial" and len(np.unique(y)) < 3:
        pytest.skip("degenerate draw")
    fit = oracle.fit(x, y, family=family, intercept=intercept, lambda_=[1.0 / n], thresh=0.1, seed=1)

    def link(v):
        v = min(max(v, 1e-9), 1 - 1e-9)
        return np.log(v / (1 - v))

    if family == "gaussian":
        want = ((y - y.mean()) ** 2).sum()
    elif family == "mgaussian":
        want = ((y - y.mean(0)) ** 2).sum()
    elif family == "binomial":
        pr = link(y.mean()) if intercept else 0.0
        want = -2 * (y * pr - np.log(1 + np.exp(pr))).sum()
    else:
        nc = 3
        pred = np.bincount(y.astype(int), minlength=nc) / n if intercept else np.ones(nc) / nc
        pred2 = np.log(pred) - np.log(pred).sum() / nc
        want = 2 * sum(np.log(np.exp(pred2).sum()) - pred2[int(c)] for c in y)
    assert fit["nulldev"] == pytest.approx(want, rel=1e-10)


def _random_data(rng, n, p, family, density=0.5):
    # tests/testthat/setup.R:6-54 in spirit
    x = rng.normal(0, 0.01, (n, p))
    x[rng.random((n, p)) > density] = 0.0
    k = 3 if family in ("multinomial", "mgaussian") else 1
    beta = rng.uniform(-1, 1, (k, p))
    centre = rng.uniform(-0.1, 0.1, p)
    scale = rng.uniform(0.95, 1.05, p)
    x = (x + centre) * scale                   # as in the reference: every entry shifted
    z = x @ beta.T + rng.uniform(-1, 1, k)
    if family == "gaussian":
        y = rng.normal(z[:, 0], 0.01)
    elif family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float)
    elif family == "multinomial":
        y = np.argmax(z + rng.gumbel(size=z.shape), axis=1).astype(float)
    else:
        y = rng.normal(z, 0.01)
    return x, y


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
@pytest.mark.parametrize("standardize", [True, False])
@pytest.mark.parametrize("intercept", [True, False])
def test_sparse_equals_dense(oracle, family, alpha, standardize, intercept):
    # test-sparse.R:3-35: same RNG stream, coefficients within 1e-3
    rng = np.random.default_rng(11)
    x, y = _random_data(rng, 1000, 2, family)
    if family == "multinomial" and len(np.unique(y)) < 3:
        pytest.skip("degenerate draw")
    kw = dict(family=family, alpha=alpha, standardize=standardize, intercept=intercept, thresh=1e-6,
              nlambda=5, seed=3)
    fs = oracle.fit(sp.csc_matrix(x), y, **kw)
    fd = oracle.fit(x, y, **kw)
    # compare_predictions(type = "coefficients", tol = 1e-3): all.equal's mean relative
    # difference over the stacked (intercept, coefficient) matrix (tests/testthat/setup.R:56-100)
    cs = np.concatenate([fs["a0"].ravel(), fs["beta"].ravel()])
    cd = np.concatenate([fd["a0"].ravel(), fd["beta"].ravel()])
    denom = np.abs(cd).mean()
    assert np.abs(cs - cd).mean() / (denom if denom > 0 else 1.0) < 1e-3


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
def test_refit_with_returned_path(oracle, family):
    # test-lambda-path.R:173-198
    rng = np.random.default_rng(8)
    n = 64
    x = rng.normal(size=(n, 4)) * [2, 100, 60, 0.5] + [6, 230, 146, 0.4]
    y = {"gaussian": rng.normal(20, 6, n), "binomial": rng.integers(0, 2, n).astype(float),
         "multinomial": rng.integers(0, 3, n).astype(float),
         "mgaussian": np.c_[rng.normal(146, 68, n), rng.normal(3.6, 0.5, n)]}[family]
    f1 = oracle.fit(x, y, family=family, nlambda=8, seed=4)
    f2 = oracle.fit(x, y, family=family, lambda_=f1["lambda"], seed=4)
    np.testing.assert_allclose(f2["beta"], f1["beta"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(f2["a0"], f1["a0"], rtol=1e-9, atol=1e-12)


def test_first_lasso_fit_is_sparse_and_debug_losses(oracle):
    # test-lambda-path.R:131-170, test-options.R
    rng = np.random.default_rng(9)
    x = rng.normal(size=(50, 1)) * 5 + 15
    y = 3.9 * x[:, 0] - 17 + rng.normal(0, 15, 50)
    fit = oracle.fit(x, y, family="gaussian", alpha=1.0, nlambda=10, debug=True, seed=1)
    assert np.all(np.abs(fit["beta"][0, :, 0]) < 1e-5)
    for ls in fit["losses"]:
        assert len(ls) >= 1 and np.all(ls > 0) and np.all(np.isfinite(ls))


def _kkt_residual(X, y, w, b, lam, mix):
    """Elastic-net KKT residual of the binomial problem on unstandardised data."""
    n = X.shape[0]
    mu = 1 / (1 + np.exp(-(X @ w + b)))
    g = X.T @ (mu - y) / n + (1 - mix) * lam * w
    l1 = mix * lam
    res = np.where(w != 0, np.abs(g + l1 * np.sign(w)), np.maximum(np.abs(g) - l1, 0.0))
    return max(res.max(), abs((mu - y).mean()))


def test_elastic_net_kkt_conditions(oracle):
    # stands in for the glmnet agreement tests (test-families.R): the fixed point of the
    # oracle is the elastic-net optimum
    rng = np.random.default_rng(12)
    n, p = 2000, 30
    X = sp.random(n, p, density=0.2, random_state=5, data_rvs=rng.standard_normal).tocsc()
    bt = rng.normal(size=p) * (rng.random(p) < 0.4)
    y = (rng.random(n) < 1 / (1 + np.exp(-(X @ bt + 0.3)))).astype(float)
    lam, mix = 0.01, 0.7
    fit = oracle.fit(X, y, family="binomial", alpha=mix, lambda_=[lam], standardize=False,
                     thresh=1e-10, maxit=5000, seed=2)
    assert fit["return_codes"][0] == 0
    r = _kkt_residual(X, y, fit["beta"][0, :, 0], fit["a0"][0, 0], lam, mix)
    assert r < 1e-6


@pytest.mark.parametrize("batch", [2, 37, 512])
def test_batched_oracle_shares_the_fixed_point(oracle, batch):
    # DESIGN.md "Batched mode": same optimum as the exact iteration
    rng = np.random.default_rng(13)
    n, p = 5000, 200
    X = sp.random(n, p, density=0.04, random_state=6, data_rvs=rng.standard_normal).tocsc()
    bt = rng.normal(size=p) * (rng.random(p) < 0.2)
    y = (rng.random(n) < 1 / (1 + np.exp(-(X @ bt)))).astype(float)
    kw = dict(family="binomial", alpha=0.5, lambda_=[20.0 / n], standardize=False, thresh=1e-11,
              maxit=4000, seed=5)
    ref = oracle.fit(X, y, **kw)
    got = oracle.fit(X, y, batch=batch, **kw)
    scale = np.abs(ref["beta"]).max()
    assert np.abs(got["beta"] - ref["beta"]).max() / scale < 1e-8
    assert abs(got["a0"] - ref["a0"]).max() < 1e-8


def test_batch_factors_match_cumulative_table(oracle):
    # closed form of lag_scaling (saga-sparse.h:229-240)
    alpha, gamma = 3e-4, 0.07
    r = 1.0 - alpha * gamma
    geo, ls = 1.0, [0.0, 1.0]
    for _ in range(2, 5001):
        geo *= r
        ls.append(ls[-1] + geo)
    for m in (1, 2, 17, 5000):
        rm, lsm = oracle.batch_factors(alpha, gamma, m)
        assert rm == pytest.approx(r ** m, rel=1e-13)
        assert lsm == pytest.approx(ls[m], rel=1e-12)
    assert oracle.batch_factors(0.0, gamma, 9) == (1.0, 9.0)


def test_batched_intercept_step_sparse_and_dense_rule(oracle):
    # orc_batch_sweep: gb += d0 / n; b -= gamma (c m gb + d0 / n) with the reference's decay c = 0.01 for sparse x
    # (saga-sparse.h:300-304) and c = 1 for a dense matrix handed over with every entry stored (saga-dense.h:170-173);
    # nothing else depends on the flag
    rng = np.random.default_rng(5)
    K, p, n, m, gamma = 3, 7, 50, 9, 0.05
    out = {}
    for dense in (False, True):
        st = oracle.new_state(K, p, n)
        st["intercept"][:] = rng.standard_normal(K) * 0 + np.array([0.3, -0.2, 0.1])
        st["g_sum_intercept"][:] = np.array([0.02, -0.01, 0.05])
        st["w"][...] = np.asfortranarray(np.arange(K * p, dtype=float).reshape(K, p) / 50.0)
        D = np.asfortranarray(np.linspace(-1, 1, K * p).reshape(K, p))
        d0 = np.array([0.4, -0.7, 0.2])
        b0, gb0 = st["intercept"].copy(), st["g_sum_intercept"].copy()
        oracle.batch_sweep((p, n), st, m, D, d0.copy(), family="mgaussian", penalty="ridge", gamma=gamma, alpha=1e-3, beta=0.0,
                           dense_intercept=dense)
        gb = gb0 + d0 / n
        c = 1.0 if dense else 0.01
        assert np.allclose(st["g_sum_intercept"], gb, rtol=1e-15)
        assert np.allclose(st["intercept"], b0 - gamma * (gb * c * m + d0 / n), rtol=1e-14)
        out[dense] = (st["w"].copy(), st["g_sum"].copy())
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])


def test_golden_paths_of_configs_1_and_2_are_the_oracles(oracle):
    """The committed golden lambda paths (tests/golden/*_path.npz: BASELINE configs 1 and 2, the latter in the
    in-repo 9-column frame and in the 4177 x 8 libsvm layout BASELINE.json names) are what the oracle computes
    today -- the GPU tests compare the HIP exact path with these files."""
    import os
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ab = np.load(os.path.join(gold_dir, "abalone.npz"))
    x8 = np.column_stack([1.0 + ab["x"][:, 0] + 2.0 * ab["x"][:, 1], ab["x"][:, 2:]])
    ir = np.load(os.path.join(gold_dir, "iris.npz"))
    for name, x, y, kw in (("abalone_gaussian_path", ab["x"], ab["y"], dict(family="gaussian", alpha=1.0, seed=2)),
                           ("abalone8_gaussian_path", x8, ab["y"], dict(family="gaussian", alpha=1.0, seed=2)),
                           ("iris_multinomial_path", ir["x"], ir["y"], dict(family="multinomial", alpha=0.8, seed=1))):
        gold = np.load(os.path.join(gold_dir, name + ".npz"))
        fit = oracle.fit(x, y, nlambda=100, thresh=1e-3, maxit=1000, **kw)
        assert fit["npasses"] == float(gold["npasses"]), name
        assert np.array_equal(fit["lambda"], gold["lambda_"]) and np.array_equal(fit["beta"], gold["beta"]), name
        assert np.array_equal(fit["a0"], gold["a0"]) and np.array_equal(fit["dev_ratio"], gold["dev_ratio"]), name
