"""Independent optimum check of the CPU oracle ("optional secondary cross-check"):
scikit-learn minimises the same elastic-net GLM objectives with different solvers and a
different parameterisation (binomial/multinomial: lambda = 1 / (C n), alpha = l1_ratio;
gaussian: sklearn's `alpha` = lambda on the original-y scale).  It is not the reference and
not trajectory-compatible; at tight tolerance the optima must agree.  CPU only, never on the
GPU box's product path."""
import numpy as np
import pytest

sk = pytest.importorskip("sklearn.linear_model")


def _data(rng, n, p):
    x = rng.standard_normal((n, p)) * rng.uniform(0.5, 1.5, p)
    return x, rng.standard_normal(p) * (rng.random(p) < 0.6)


@pytest.mark.parametrize("alpha", [0.0, 0.4, 1.0])
def test_binomial_optimum_equals_sklearn(oracle, alpha):
    rng = np.random.default_rng(0)
    n, p, lam = 400, 6, 0.02
    x, b = _data(rng, n, p)
    y = (rng.random(n) < 1 / (1 + np.exp(-(x @ b + 0.3)))).astype(float)
    fit = oracle.fit(x, y, family="binomial", alpha=alpha, lambda_=[lam], standardize=False, thresh=1e-12,
                     maxit=20000, seed=1)
    m = sk.LogisticRegression(penalty="elasticnet", solver="saga", l1_ratio=alpha, C=1.0 / (lam * n),
                              tol=1e-13, max_iter=200000, fit_intercept=True).fit(x, y)
    assert np.abs(fit["beta"][0, :, 0] - m.coef_[0]).max() < 2e-6
    assert abs(fit["a0"][0, 0] - m.intercept_[0]) < 2e-6


@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
def test_gaussian_optimum_equals_sklearn(oracle, alpha):
    rng = np.random.default_rng(1)
    n, p, lam = 300, 5, 0.05
    x, b = _data(rng, n, p)
    y = x @ b + 2.0 + 0.5 * rng.standard_normal(n)
    fit = oracle.fit(x, y, family="gaussian", alpha=alpha, lambda_=[lam], standardize=False, thresh=1e-13,
                     maxit=20000, seed=1)
    # the reference standardises y (population sd s) and fits with lambda / s (src/families.h
    # Preprocess, src/utils.h:170-179): on the original scale the L1 weight is lambda * alpha but the
    # L2 weight is lambda * (1 - alpha) / s -- the glmnet convention
    sd = y.std()
    l1, l2 = lam * alpha, lam * (1 - alpha) / sd
    if alpha == 0.0:
        m = sk.Ridge(alpha=l2 * n, fit_intercept=True, tol=1e-14).fit(x, y)       # ||.||^2 + a ||b||^2
    else:
        m = sk.ElasticNet(alpha=l1 + l2, l1_ratio=l1 / (l1 + l2), fit_intercept=True, tol=1e-14,
                          max_iter=1000000).fit(x, y)
    assert np.abs(fit["beta"][0, :, 0] - m.coef_).max() < 1e-6
    assert abs(fit["a0"][0, 0] - m.intercept_) < 1e-6


def test_multinomial_optimum_equals_sklearn(oracle):
    rng = np.random.default_rng(2)
    n, p, K, lam, alpha = 500, 4, 3, 0.01, 0.5
    x = rng.standard_normal((n, p))
    y = np.argmax(x @ rng.standard_normal((p, K)) + rng.gumbel(size=(n, K)), axis=1).astype(float)
    fit = oracle.fit(x, y, family="multinomial", alpha=alpha, lambda_=[lam], standardize=False, thresh=1e-12,
                     maxit=20000, seed=1)
    m = sk.LogisticRegression(penalty="elasticnet", solver="saga", l1_ratio=alpha, C=1.0 / (lam * n),
                              tol=1e-13, max_iter=400000, fit_intercept=True).fit(x, y)
    # the softmax parameterisation is unique up to a per-feature shift only where no coefficient
    # of the feature is thresholded; compare class-centred coefficients and probabilities
    bo = fit["beta"][:, :, 0]
    assert np.abs((bo - bo.mean(axis=0)) - (m.coef_ - m.coef_.mean(axis=0))).max() < 5e-5
    lp = x @ bo.T + fit["a0"][:, 0]
    pr = np.exp(lp - lp.max(axis=1, keepdims=True))
    pr /= pr.sum(axis=1, keepdims=True)
    assert np.abs(pr - m.predict_proba(x)).max() < 2e-5
