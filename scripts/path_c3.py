"""End-to-end 100-lambda path on BASELINE config 3 (1M x 1000 sparse logistic) through sgdnet()."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
n, p, dens, seed = 1_000_000, 1_000, 0.01, 3
pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed)
X = D.as_scipy(pr).T.tocsc()
y = pr["y"][0]
for std in ((False,) if os.environ.get("ONLY_UNSTD") else (False, True)):
    for mode in ("auto",):
        t = time.time()
        fit = sa.sgdnet(X, y, family="binomial", alpha=0.5, nlambda=100, standardize=std, thresh=1e-5, maxit=1000,
                        seed=seed, mode=mode, batch=int(os.environ.get("FIX_BATCH", "0")))
        dt = time.time() - t
        print(f"C3 100-lambda path standardize={std} mode={mode}: {dt:.2f}s, npasses={fit.npasses:.0f}, "
              f"{dt / fit.npasses * 1e3:.2f} ms/epoch incl. everything, df(last)={fit.df[-1]}, "
              f"dev.ratio(last)={fit.dev_ratio[-1]:.4f}, rc sum={fit.return_codes.sum():.0f}", flush=True)
