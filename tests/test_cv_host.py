"""Host-side mirrors of the reference's R/predict.sgdnet.R, R/score.R and R/cv_sgdnet.R
(SURVEY.md 8 row f4): formulas against independent numpy restatements, and R's sample()
against values R itself prints (no GPU needed: nothing here fits a model)."""
import numpy as np
import pytest

import sgdnet_amd as sa
from sgdnet_amd import cv as CV
from sgdnet_amd.api import SgdnetFit
from sgdnet_amd.predict import lambda_interpolate
from sgdnet_amd.score import auc


def test_r_sample_reproduces_r():
    # R >= 3.6.0 (sample.kind = "Rejection"): set.seed(s); sample(10)
    assert sa.RRng(1).sample(10).tolist() == [9, 4, 7, 1, 2, 5, 3, 10, 6, 8]
    assert sa.RRng(42).sample(10).tolist() == [1, 5, 10, 8, 2, 4, 6, 9, 7, 3]
    assert sa.RRng(123).sample(10).tolist() == [3, 10, 2, 8, 6, 9, 1, 7, 5, 4]
    p = sa.RRng(7).sample(1000)
    assert sorted(p.tolist()) == list(range(1, 1001))
    assert sa.RRng(7).sample(1000, 5).tolist() == p[:5].tolist()


def test_cut_into_folds_like_r():
    # as.numeric(cut(1:10, 3)) in R: 1 1 1 1 2 2 2 3 3 3
    assert CV.r_cut(np.arange(1, 11), 3).tolist() == [1, 1, 1, 1, 2, 2, 2, 3, 3, 3]
    f = CV.r_cut(sa.RRng(1).sample(103), 10)
    assert f.min() == 1 and f.max() == 10 and np.bincount(f)[1:].min() >= 10


def _fit(family, a0, beta, lam, classnames=None, grouped=False):
    return SgdnetFit(a0=a0, beta=beta, lambda_=np.asarray(lam, float), dev_ratio=None, df=None, nulldev=1.0,
                     npasses=1, alpha=1, offset=False, classnames=classnames, grouped=grouped, nobs=1,
                     family=family)


def test_lambda_interpolate_and_coef():
    lam = np.array([1.0, 0.5, 0.25, 0.125])
    left, right, frac = lambda_interpolate(lam, [0.5, 0.3, 2.0, 0.01])
    assert (left[0], right[0]) == (1, 1) and frac[0] == 1.0
    assert (left[1], right[1]) == (1, 2) and np.isclose(frac[1], (0.3 - 0.25) / (0.5 - 0.25))
    assert left[2] == right[2] == 0 and left[3] == right[3] == 3              # clamped to the path
    beta = np.arange(8.0).reshape(2, 4)
    fit = _fit("gaussian", np.array([10.0, 20.0, 30.0, 40.0]), beta, lam)
    c = sa.coef(fit, s=[0.3])
    w = frac[1]
    assert np.allclose(c[:, 0], w * np.r_[20.0, beta[:, 1]] + (1 - w) * np.r_[30.0, beta[:, 2]])


def test_predict_and_score_single_response():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((30, 3))
    beta = rng.standard_normal((3, 2))
    a0 = np.array([0.3, -0.2])
    lp = X @ beta + a0
    fit = _fit("gaussian", a0, beta, [0.1, 0.01])
    y = rng.standard_normal(30)
    assert np.allclose(sa.predict(fit, X), lp)
    assert np.allclose(sa.score(fit, X, y, "mse"), ((lp - y[:, None]) ** 2).mean(axis=0))
    assert np.allclose(sa.score(fit, X, y, "mae"), np.abs(lp - y[:, None]).mean(axis=0))
    fb = _fit("binomial", a0, beta, [0.1, 0.01], classnames=["no", "yes"])
    yb = np.where(rng.random(30) < 0.5, "yes", "no")
    pr = 1 / (1 + np.exp(-lp))
    t = (yb == "yes").astype(float)[:, None]
    assert np.allclose(sa.predict(fb, X, type="response"), pr)
    assert (sa.predict(fb, X, type="class") == np.where(lp > 0, "yes", "no")).all()
    prc = np.clip(pr, 1e-5, 1 - 1e-5)
    assert np.allclose(sa.score(fb, X, yb, "deviance"), (-2 * (t * np.log(prc) + (1 - t) * np.log(1 - prc))).mean(axis=0))
    assert np.allclose(sa.score(fb, X, yb, "class"), ((pr > 0.5) != (t > 0.5)).mean(axis=0))
    assert np.allclose(sa.score(fb, X, yb, "mse"), (2 * (pr - t) ** 2).mean(axis=0))
    # auc against the rank-sum definition (no ties here)
    a = sa.score(fb, X, yb, "auc")
    pos, neg = pr[t[:, 0] == 1, 0], pr[t[:, 0] == 0, 0]
    assert np.isclose(a[0], (pos[:, None] > neg[None, :]).mean())
    assert np.isclose(auc((t[:, 0] == 1).astype(float), pr[:, 0]), a[0])
    with pytest.raises(ValueError):
        sa.score(fit, X, y, "auc")


def test_predict_and_score_multi_response():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((25, 4))
    K, L = 3, 2
    beta = [rng.standard_normal((4, L)) for _ in range(K)]
    a0 = rng.standard_normal((K, L))
    dp = np.stack([X @ beta[k] + a0[k] for k in range(K)], axis=1)            # (n, K, L)
    fm = _fit("multinomial", a0, beta, [0.1, 0.01], classnames=["a", "b", "c"])
    assert np.allclose(sa.predict(fm, X), dp)
    pr = np.exp(dp) / np.exp(dp).sum(axis=1, keepdims=True)
    assert np.allclose(sa.predict(fm, X, type="response"), pr)
    y = np.array(["a", "b", "c"])[rng.integers(0, 3, 25)]
    Y = (y[:, None] == np.array(["a", "b", "c"])[None, :]).astype(float)[:, :, None]
    assert np.allclose(sa.score(fm, X, y, "deviance"), (-2 * Y * np.log(np.clip(pr, 1e-5, 1 - 1e-5))).sum(axis=1).mean(axis=0))
    assert np.allclose(sa.score(fm, X, y, "class"), (np.argmax(pr, axis=1) != np.argmax(Y, axis=1)).mean(axis=0))
    fg = _fit("mgaussian", a0, beta, [0.1, 0.01], grouped=True)
    ym = rng.standard_normal((25, K))
    assert np.allclose(sa.score(fg, X, ym, "mse"), ((dp - ym[:, :, None]) ** 2).sum(axis=0).mean(axis=0))
    nz = sa.predict(fg, type="nonzero")
    assert len(nz) == L and all(len(v) == 4 for v in nz)


def test_cv_summaries():
    raw = np.array([[1.0, 2.0, 4.0], [3.0, 2.0, 0.0], [2.0, 2.0, 5.0]])
    s = CV.summarize_cv_raw(raw)
    assert np.allclose(s[:, 0], raw.mean(axis=0)) and np.allclose(s[:, 1], raw.std(axis=0, ddof=1))
    assert np.allclose(s[:, 2], s[:, 0] - s[:, 1]) and np.allclose(s[:, 3], s[:, 0] + s[:, 1])
    block = np.column_stack([np.full(3, 0.5), [1.0, 0.5, 0.25], s])
    o = CV.find_optimum(block)
    assert o["lambda_min"] == 1.0 and o["error_min"] == 2.0 and o["lambda_1se"] == 1.0
    block[:, 2] = [2.0, 1.0, 1.2]
    block[:, 3] = [0.1, 0.3, 0.1]
    o = CV.find_optimum(block)
    assert o["lambda_min"] == 0.5 and o["lambda_1se"] == 0.5      # 1.2 <= 1.3 but its lambda (0.25) is smaller
