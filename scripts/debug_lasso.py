import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
rng = np.random.default_rng(2)
Xs = sp.random(200_000, 2_000, density=0.005, format="csc", random_state=3)
b = rng.standard_normal(2000) * (rng.random(2000) < 0.1)
ys = (rng.random(200_000) < 1 / (1 + np.exp(-np.asarray(Xs @ b).ravel()))).astype(int)
lam = np.geomspace(2e-3, 2e-4, 6)
for a in (1.0, 0.99, 0.9):
    for batch in (0, 4000, 1000, 250):
        t = time.time()
        fit = sa.sgdnet(Xs, ys, family="binomial", alpha=a, lambda_=lam, thresh=1e-5, standardize=False, mode="batched",
                        batch=batch, maxit=400, seed=3)
        print(f"alpha={a} batch={batch}: {time.time()-t:.2f}s npasses={fit.npasses:.0f} rc={fit.return_codes.astype(int)} "
              f"df={fit.df}", flush=True)
t = time.time(); fit = sa.sgdnet(Xs[:20000], ys[:20000], family="binomial", alpha=1.0, lambda_=lam, thresh=1e-5, standardize=False, mode="exact", maxit=400, seed=3)
print(f"exact mode on 20k rows alpha=1: {time.time()-t:.2f}s npasses={fit.npasses:.0f} rc={fit.return_codes.astype(int)}")
