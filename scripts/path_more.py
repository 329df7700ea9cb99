"""More end-to-end lambda paths (catching pathologies of the path driver): C4 with 100 lambdas,
a C5-shaped multinomial problem, a gaussian one with standardize=TRUE."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
which = sys.argv[1] if len(sys.argv) > 1 else "multi"
if which == "C4":
    n, p, dens, fam, K, seed = 10_000_000, 10_000, 0.001, "binomial", 1, 4
elif which == "multi":
    n, p, dens, fam, K, seed = 500_000, 20_000, 0.0005, "multinomial", 10, 5
else:
    n, p, dens, fam, K, seed = 1_000_000, 2_000, 0.005, "gaussian", 1, 6
pr = D.make_sparse_glm(n, p, dens, family=fam, seed=seed, n_classes=K) if K > 1 else D.make_sparse_glm(n, p, dens, family=fam, seed=seed)
X = D.as_scipy(pr).T.tocsc()
y = pr["y"][0]
for std in (False, True):
    t = time.time()
    fit = sa.sgdnet(X, y, family=fam, alpha=0.5, nlambda=100, standardize=std, thresh=1e-5, maxit=1000, seed=seed, mode="auto")
    dt = time.time() - t
    print(f"{which} {n}x{p} {fam}: 100-lambda path standardize={std}: {dt:.2f}s, npasses={fit.npasses:.0f}, "
          f"{dt / fit.npasses * 1e3:.2f} ms/epoch incl. everything, rc sum={fit.return_codes.sum():.0f}", flush=True)
