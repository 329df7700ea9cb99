"""Epochs to tolerance, exact iteration against mode = auto, multinomial elastic net on sparse x with p > n (the shape
whose tight-threshold fits hit maxit in scripts/dev/binned_thresh_sweep.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa
out = open(os.path.join(ROOT, "gpurun_out", "mep.log"), "w")
def say(*a):
    print(*a, flush=True); print(*a, file=out, flush=True)
r = np.random.default_rng(19003)
K, n, p, nnz_row = 3, 20000, 30000, 15
rows = np.repeat(np.arange(n), nnz_row); cols = r.integers(0, p, n * nnz_row); vals = r.standard_normal(n * nnz_row)
X = sp.csc_matrix((vals, (rows, cols)), shape=(n, p)); X.sum_duplicates()
W = r.standard_normal((p, K)) * (r.random((p, K)) < 0.05)
z = X @ W
y = np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float); y[:K] = np.arange(K)
path = sa.sgdnet(X, y, family="multinomial", alpha=0.5, standardize=False, nlambda=8, maxit=1, mode="auto").lambda_
for li in (3, 5):
    lam = [path[li]]
    for thresh in (1e-3, 1e-5):
        t = time.time(); a = sa.sgdnet(X, y, family="multinomial", alpha=0.5, standardize=False, lambda_=lam, thresh=thresh, maxit=1200, mode="auto"); ta = time.time() - t
        say(f"lambda[{li}] thresh {thresh}: auto {a.npasses:.0f} epochs ({ta:.2f}s) rc {a.return_codes} dev_ratio {a.dev_ratio[0]:.6f}")
        t = time.time(); e = sa.sgdnet(X, y, family="multinomial", alpha=0.5, standardize=False, lambda_=lam, thresh=thresh, maxit=1200); te = time.time() - t
        say(f"lambda[{li}] thresh {thresh}: exact {e.npasses:.0f} epochs ({te:.2f}s) rc {e.return_codes} dev_ratio {e.dev_ratio[0]:.6f}")
