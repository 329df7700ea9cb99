"""End-to-end sgdnet() on a dense problem (host buffers in, coefficients out), phase trace."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 100)
rng = np.random.default_rng(1)
X = np.asfortranarray(rng.standard_normal((n, p)))
beta = rng.standard_normal(p) * (rng.random(p) < 0.3)
y = (rng.random(n) < 1 / (1 + np.exp(-(X @ beta)))).astype(float)
for mode in ("batched",):
    t = time.time()
    fit = sa.sgdnet(X, y, family="binomial", alpha=0.5, lambda_=[1.0 / n], thresh=1e-5, maxit=200, seed=1, mode=mode)
    print(f"dense {n}x{p} mode={mode}: {time.time()-t:.2f}s total, npasses={fit.npasses:.0f}, nnz(beta)={np.count_nonzero(fit.beta)}", flush=True)
