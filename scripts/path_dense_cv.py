"""End-to-end checks: a dense 100-lambda path and a cross-validation (device scoring)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
rng = np.random.default_rng(2)
n, p = 300_000, 200
X = rng.standard_normal((n, p)); beta = rng.standard_normal(p) * (rng.random(p) < 0.2)
y = X @ beta + rng.standard_normal(n)
for std in (False, True):
    t = time.time(); fit = sa.sgdnet(X, y, family="gaussian", alpha=0.5, nlambda=100, standardize=std, thresh=1e-5, mode="auto", seed=1)
    print(f"dense {n}x{p} gaussian 100-lambda path standardize={std}: {time.time()-t:.2f}s, npasses={fit.npasses:.0f}", flush=True)
Xs = sp.random(200_000, 2_000, density=0.005, format="csc", random_state=3)
b = rng.standard_normal(2000) * (rng.random(2000) < 0.1)
ys = (rng.random(200_000) < 1 / (1 + np.exp(-np.asarray(Xs @ b).ravel()))).astype(int)
for dev in ([0], [0, 0]):
    t = time.time(); cv = sa.cv_sgdnet(Xs, ys, family="binomial", alpha=[0.5, 1.0], nfolds=5, nlambda=50, thresh=1e-5,
                                      standardize=False, mode="auto", devices=dev, train_on="rest")
    print(f"cv_sgdnet 200k x 2k sparse binomial, 2 alphas x 5 folds x 50 lambdas, devices={dev}: {time.time()-t:.2f}s, "
          f"lambda_min={cv.lambda_min:.5f}", flush=True)
