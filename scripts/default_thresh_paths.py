import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import sgdnet_amd as sa
from sgdnet_amd import data as D
pr = D.make_sparse_glm(1_000_000, 1_000, 0.01, family="binomial", seed=3)
X = D.as_scipy(pr).T.tocsc(); y = pr["y"][0]
for std in (False, True):
    t=time.time(); fit = sa.sgdnet(X, y, family="binomial", alpha=0.5, nlambda=100, standardize=std, mode="auto", seed=3)
    print(f"C3 default thresh std={std}: {time.time()-t:.2f}s npasses={fit.npasses:.0f} monotone={np.all(np.diff(fit.dev_ratio) > -1e-3)}", flush=True)
Xp = sp.random(300_000, 1_500, density=0.01, format="csc", random_state=5)
b = np.random.default_rng(1).standard_normal(1500) * (np.random.default_rng(2).random(1500) < 0.1)
yp = np.asarray(Xp @ b).ravel() + np.random.default_rng(3).standard_normal(300_000)
for std in (False, True):
    t=time.time(); fit = sa.sgdnet(Xp, yp, family="gaussian", alpha=1.0, nlambda=100, standardize=std, mode="auto", seed=3)
    print(f"positive sparse gaussian default thresh std={std}: {time.time()-t:.2f}s npasses={fit.npasses:.0f} monotone={np.all(np.diff(fit.dev_ratio) > -1e-3)}", flush=True)
