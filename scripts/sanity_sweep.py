"""Auto-mode sanity sweep: families x data kinds x alpha; path must be monotone in deviance,
converge, and agree with a conservative small-window fit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
rng = np.random.default_rng(7)


def data(kind, n, p):
    if kind == "sparse_pos":
        X = sp.random(n, p, density=0.01, format="csc", random_state=1)
    elif kind == "sparse_gauss":
        X = sp.random(n, p, density=0.01, format="csc", random_state=1, data_rvs=rng.standard_normal)
    elif kind == "dense_corr":
        z = rng.standard_normal((n, 1))
        X = 0.8 * z + 0.6 * rng.standard_normal((n, p))
    else:
        X = rng.standard_normal((n, p))
    return X


def response(X, fam, K):
    p = X.shape[1]
    B = rng.standard_normal((p, K)) * (rng.random((p, K)) < 0.1)
    eta = np.asarray(X @ B)
    eta = eta / max(eta.std(), 1e-9)
    if fam == "gaussian":
        return eta[:, 0] + rng.standard_normal(X.shape[0])
    if fam == "binomial":
        return (rng.random(X.shape[0]) < 1 / (1 + np.exp(-eta[:, 0]))).astype(int)
    if fam == "multinomial":
        return np.argmax(eta + rng.gumbel(size=eta.shape), axis=1)
    return eta + rng.standard_normal(eta.shape)


bad = 0
for kind, n, p in (("sparse_pos", 100_000, 1_000), ("sparse_gauss", 100_000, 1_000), ("dense_corr", 50_000, 60), ("dense_iid", 50_000, 60)):
    X = data(kind, n, p)
    for fam, K in (("gaussian", 1), ("binomial", 1), ("multinomial", 3), ("mgaussian", 2)):
        y = response(X, fam, K)
        for alpha in (1.0, 0.5, 0.0):
            for std in (False, True):
                t = time.time()
                try:
                    fit = sa.sgdnet(X, y, family=fam, alpha=alpha, nlambda=15, thresh=1e-5, standardize=std, mode="auto", maxit=500, seed=1)
                    ref = sa.sgdnet(X, y, family=fam, alpha=alpha, lambda_=fit.lambda_, thresh=1e-5, standardize=std, mode="batched",
                                    batch=64, maxit=500, seed=1)
                    dt = time.time() - t
                    mono = np.all(np.diff(fit.dev_ratio) > -1e-4)
                    agree = np.abs(fit.dev_ratio - ref.dev_ratio).max()
                    ok = mono and agree < 2e-3 and fit.return_codes.sum() <= ref.return_codes.sum()
                    if not ok:
                        bad += 1
                    print(f"{'ok ' if ok else 'BAD'} {kind:12s} {fam:11s} alpha={alpha} std={std}: npasses={fit.npasses:.0f} (ref {ref.npasses:.0f}) "
                          f"rc={fit.return_codes.sum():.0f}/{ref.return_codes.sum():.0f} max|ddev|={agree:.2e} mono={mono} {dt:.1f}s", flush=True)
                except Exception as e:
                    bad += 1
                    print(f"ERR {kind} {fam} alpha={alpha} std={std}: {str(e)[:100]}", flush=True)
print("bad cases:", bad)
