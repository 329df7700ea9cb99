import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
rng = np.random.default_rng(2)
Xs = sp.random(200_000, 2_000, density=0.005, format="csr", random_state=3)
b = rng.standard_normal(2000) * (rng.random(2000) < 0.1)
ys = (rng.random(200_000) < 1 / (1 + np.exp(-np.asarray(Xs @ b).ravel()))).astype(int)
full = sa.sgdnet(Xs.tocsc(), ys, family="binomial", alpha=1.0, nlambda=50, thresh=1e-5, standardize=False, mode="auto")
print("full fit npasses", full.npasses, "rc", full.return_codes.sum())
for nsub in (160_000, 199_000):
    for a in (1.0,):
        t = time.time()
        fit = sa.sgdnet(Xs[:nsub].tocsc(), ys[:nsub], family="binomial", alpha=a, lambda_=full.lambda_, thresh=1e-5,
                        standardize=False, mode="auto", seed=3)
        print(f"n={nsub} alpha={a}: {time.time()-t:.2f}s npasses={fit.npasses:.0f} rc sum={fit.return_codes.sum():.0f} "
              f"first rc=1 at {np.flatnonzero(fit.return_codes)[:3]}", flush=True)
