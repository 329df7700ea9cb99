import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
n, p = 240_000, 100
pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=31)
x, y = D.as_scipy(pr).T.tocsc(), pr["y"].ravel()
kw = dict(family="binomial", alpha=0.5, lambda_=[3e-3, 1.5e-3, 7e-4], standardize=True, thresh=1e-7, maxit=80, mode="batched")
r1, r2, r3 = sa.RRng(12), sa.RRng(12), sa.RRng(12)
one = sa.sgdnet(x, y, rng=r1, **kw)
with sa.option("host_setup", 1):
    oneh = sa.sgdnet(x, y, rng=r3, **kw)
two = sa.sgdnet(x, y, rng=r2, devices=[0, 0], **kw)
for nm, f in (("one", one), ("one_host", oneh), ("two", two)):
    print(nm, f.npasses, f.draws_used, np.asarray(f.dev_ratio), np.abs(f.beta).max(), f.a0)
print("two vs one_host beta", np.abs(two.beta - oneh.beta).max(), "one vs one_host", np.abs(one.beta - oneh.beta).max())
print(np.array_equal(r3.unif(8), r2.unif(8)))
