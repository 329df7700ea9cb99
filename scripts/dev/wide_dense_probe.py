import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time
import torch
import sgdnet_amd as sa
rng = np.random.default_rng(4)
n, p = 1500, 10500
X = 0.1 * rng.standard_normal((n, p))
bt = np.zeros(p); bt[:20] = 10 * rng.standard_normal(20)
y = X @ bt + 0.1 * rng.standard_normal(n) + 0.7
ys = y.std()
for lam, a in ((0.3, 0.3),):
    for mode, thresh, batch in (("batched", 1e-10, 0), ("exact", 1e-10, 0)):
        t = time.time()
        fit = sa.sgdnet(X, y, seed=3, mode=mode, family="gaussian", alpha=a, lambda_=[lam], standardize=False, thresh=thresh, maxit=4000, batch=batch)
        w, b = fit.beta[:, 0], fit.a0[0]
        r = X @ w + b - y
        g = X.T @ r / n + lam * (1 - a) * w / ys
        nz = w != 0
        print(lam, a, mode, thresh, batch, "rc", fit.return_codes, "npasses", fit.npasses, "sec", time.time() - t, "nz", nz.sum(),
              "kkt", np.abs(g[nz] + lam * a * np.sign(w[nz])).max() / lam, np.abs(g[~nz]).max() / (lam * a), abs(r.mean()), flush=True)
