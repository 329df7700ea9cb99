"""R-side behaviour the reference's documented examples depend on, restated for the tests
(TEST INFRASTRUCTURE): the random-number consumers of base R as they were when the reference's
pkgdown site was built (pkgdown 1.1.0, 2018 => R 3.5.x), and the reference's own R front-end
(R/sgdnet.R post-processing) around a pluggable native fit, so that the same flow can be driven
by the CPU oracle (CPU tests) or by libsgdnet_hip.so (GPU tests).

Every generator below takes `rng`, any object with `.unif(count) -> ndarray` that advances R's
Mersenne-Twister (oracle.pyoracle.Rng and sgdnet_amd.RRng both do).
"""
import numpy as np

from sgdnet_amd.api import SgdnetFit


def sample_rounding(rng, n, size=None):
    """sample(n, size) of R < 3.6.0 (sample.kind = "Rounding"; R's src/main/random.c
    do_sample -> without replacement: j = floor(n * unif_rand()); y[i] = x[j] + 1; x[j] = x[--n]).
    1-based, like R."""
    k = n if size is None else size
    xs = list(range(n))
    out = np.empty(k, dtype=np.int64)
    m = n
    for i in range(k):
        j = int(np.floor(m * rng.unif()[0]))
        out[i] = xs[j] + 1
        m -= 1
        xs[j] = xs[m]
    return out


def rnorm(rng, count):
    """rnorm(count) with normal.kind = "Inversion" (R's src/nmath/snorm.c):
    u = unif_rand(); u = (int)(2^27 u) + unif_rand(); qnorm(u / 2^27)."""
    from scipy.special import ndtri
    big = 134217728.0
    out = np.empty(count)
    for i in range(count):
        u = rng.unif()[0]
        u = float(int(big * u)) + rng.unif()[0]
        out[i] = ndtri(u / big)
    return out


def runif(rng, count, lo=0.0, hi=1.0):
    return lo + (hi - lo) * rng.unif(count)


def encode_response(y, family):
    """R/sgdnet.R:277-339: (encoded y as an (n, Ky) float matrix, n_classes, class names)."""
    y = np.asarray(y)
    if family in ("binomial", "multinomial"):
        levels = np.unique(y.reshape(-1))
        codes = np.searchsorted(levels, y.reshape(-1)).astype(np.float64)
        K = 1 if family == "binomial" else levels.size
        return codes.reshape(-1, 1), K, [str(v) for v in levels]
    y = y.astype(np.float64)
    y = y.reshape(y.shape[0], -1)
    return y, (1 if family == "gaussian" else y.shape[1]), None


def make_fit(raw, *, family, alpha, n_samples, class_names):
    """R/sgdnet.R:368-431 on the native list (a0 K x L, beta K x p x L, ...)."""
    a0, beta = raw["a0"], raw["beta"]
    K = a0.shape[0]
    dfmat = None
    if family in ("gaussian", "binomial"):
        a0_out = a0[0, :].copy()
        beta_out = beta[0, :, :].copy()
        df = (beta_out != 0).sum(axis=0)
    else:
        a0_out = a0.copy()
        beta_out = [beta[k, :, :].copy() for k in range(K)]
        df = (sum(beta_out) != 0).sum(axis=0)
        dfmat = np.vstack([(np.abs(bk) > 0).sum(axis=0) for bk in beta_out])
    if family == "multinomial":
        a0_out = a0_out - a0_out.mean(axis=0, keepdims=True)
    return SgdnetFit(a0=a0_out, beta=beta_out, lambda_=raw["lambda"], dev_ratio=raw["dev_ratio"], df=df,
                     nulldev=raw["nulldev"], npasses=raw["npasses"], alpha=alpha, offset=False,
                     classnames=class_names, grouped=family == "mgaussian", nobs=n_samples, family=family,
                     dfmat=dfmat, return_codes=raw.get("return_codes"), draws_used=raw.get("draws_used", 0))


def oracle_sgdnet(x, y, family="gaussian", alpha=1, nlambda=100, lambda_min_ratio=None, lambda_=None,
                  maxit=1000, standardize=True, intercept=True, thresh=0.001, standardize_response=False, *,
                  rng=None, **_ignored):
    """sgdnet() of the reference with the CPU oracle behind the front-end (tests only)."""
    from oracle import pyoracle as po
    import scipy.sparse as sp
    if not sp.issparse(x):
        x = np.asarray(x, dtype=np.float64)
        x = x.reshape(x.shape[0], -1)
    y_enc, K, names = encode_response(y, family)
    raw = po.fit(x, y_enc, family=family, alpha=float(alpha), nlambda=nlambda, lambda_min_ratio=lambda_min_ratio,
                 lambda_=lambda_, maxit=maxit, standardize=standardize, intercept=intercept, thresh=thresh,
                 standardize_response=standardize_response, n_classes=K, rng=rng)
    return make_fit(raw, family=family, alpha=alpha, n_samples=x.shape[0], class_names=names)
