import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
rng = np.random.default_rng(2)
Xs = sp.random(200_000, 2_000, density=0.005, format="csc", random_state=3)
b = rng.standard_normal(2000) * (rng.random(2000) < 0.1)
ys = (rng.random(200_000) < 1 / (1 + np.exp(-np.asarray(Xs @ b).ravel()))).astype(int)
for a, mode, extra in ((1.0, "auto", {}), (1.0, "batched", dict(batch=1000)), (0.5, "auto", {})):
    fit = sa.sgdnet(Xs, ys, family="binomial", alpha=a, nlambda=20, thresh=1e-5, standardize=False, mode=mode, maxit=300, seed=3,
                    debug=False, **extra)
    print(f"alpha={a} mode={mode} {extra}: npasses={fit.npasses:.0f}")
    print("  lambda", np.array2string(fit.lambda_, precision=5, max_line_width=200))
    print("  rc    ", fit.return_codes.astype(int))
    print("  df    ", fit.df)
    print("  devrat", np.array2string(fit.dev_ratio, precision=4, max_line_width=200))
