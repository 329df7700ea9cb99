"""Virtual shards on non-negative sparse data (common component): auto mode vs a conservative fit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
rng = np.random.default_rng(9)
n, p = 1_000_000, 1_000
X = sp.random(n, p, density=0.01, format="csc", random_state=4)
b = rng.standard_normal(p) * (rng.random(p) < 0.1)
eta = np.asarray(X @ b).ravel(); eta /= eta.std()
for fam, y in (("binomial", (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(int)), ("gaussian", eta + rng.standard_normal(n))):
    for alpha in (1.0, 0.5):
        t = time.time()
        fit = sa.sgdnet(X, y, family=fam, alpha=alpha, nlambda=20, thresh=1e-5, standardize=False, mode="auto", maxit=300, seed=1)
        dt = time.time() - t
        os.environ["SGDNET_VSHARDS"] = "0"
        ref = sa.sgdnet(X, y, family=fam, alpha=alpha, lambda_=fit.lambda_, thresh=1e-5, standardize=False, mode="batched", batch=256,
                        maxit=300, seed=1)
        del os.environ["SGDNET_VSHARDS"]
        print(f"{fam} alpha={alpha}: auto {dt:.2f}s npasses={fit.npasses:.0f} rc={fit.return_codes.sum():.0f}; ref npasses={ref.npasses:.0f}; "
              f"max|d dev.ratio|={np.abs(fit.dev_ratio-ref.dev_ratio).max():.2e} mono={np.all(np.diff(fit.dev_ratio) > -1e-5)} "
              f"max|d beta|={np.abs(fit.beta-ref.beta).max():.2e}", flush=True)
