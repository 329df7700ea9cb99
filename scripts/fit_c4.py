"""End-to-end sgdnet() on the C4 problem through the C ABI (host buffers in, coefficients out)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
n, p, dens, seed = {"C4": (10_000_000, 10_000, 0.001, 4), "C3": (1_000_000, 1_000, 0.01, 3)}[wl]
pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed)
X = D.as_scipy(pr).T.tocsc()    # n x p feature-major, what R passes
y = pr["y"][0]
for std in (False, True):
    t = time.time()
    fit = sa.sgdnet(X, y, family="binomial", alpha=0.5, lambda_=[1.0 / n], standardize=std, thresh=1e-6,
                    maxit=200, seed=seed, mode="batched")
    dt = time.time() - t
    print(f"{wl} standardize={std}: {dt:.2f}s total, npasses={fit.npasses:.0f}, dev.ratio={fit.dev_ratio[0]:.6f}, "
          f"nnz(beta)={np.count_nonzero(fit.beta)}, a0={fit.a0[0]:.6f}, rc={fit.return_codes[0]}", flush=True)
