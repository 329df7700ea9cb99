"""Multi-class sparse fits large enough for the binned form: default thresh against thresh 1e-7, and the optimality
of the tight fit (max |KKT residual| / lambda for the elastic net, computed on the host)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa

out = open(os.path.join(ROOT, "gpurun_out", "bts.log"), "w")
def say(*a):
    print(*a, flush=True); print(*a, file=out, flush=True)

for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    r = np.random.default_rng(19000 + seed)
    K = int(r.choice([3, 10, 24]))
    n = int(r.choice([100_000, 300_000]))
    p = int(r.choice([5000, 30000]))
    nnz_row = int(r.choice([5, 15]))
    rows = np.repeat(np.arange(n), nnz_row)
    cols = r.integers(0, p, n * nnz_row)
    vals = r.standard_normal(n * nnz_row)
    X = sp.csc_matrix((vals, (rows, cols)), shape=(n, p)); X.sum_duplicates()
    W = r.standard_normal((p, K)) * (r.random((p, K)) < 0.05)
    z = X @ W
    y = np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float)
    y[:K] = np.arange(K)
    kw = dict(family="multinomial", alpha=0.5, standardize=False, nlambda=8, mode="auto")
    t = time.time(); a = sa.sgdnet(X, y, seed=seed, **kw); ta = time.time() - t
    t = time.time(); b = sa.sgdnet(X, y, seed=seed, thresh=1e-7, maxit=2000, **kw); tb = time.time() - t
    d = np.abs(np.asarray(a.dev_ratio) - np.asarray(b.dev_ratio))
    # KKT of the tight fit at the last lambda: g = X'(P - Y)/n + lam (1 - alpha) B ; |g| <= lam alpha where B == 0
    lam = b.lambda_[-1]
    B = np.stack([bk[:, -1] for bk in b.beta], axis=1)            # (p, K)
    a0 = b.a0[:, -1]
    eta = X @ B + a0
    eta -= eta.max(axis=1, keepdims=True)
    P = np.exp(eta); P /= P.sum(axis=1, keepdims=True)
    Y = np.zeros_like(P); Y[np.arange(n), y.astype(int)] = 1.0
    G = (X.T @ (P - Y)) / n + lam * 0.5 * B
    nz = B != 0
    kkt = max(np.abs(G[nz] + lam * 0.5 * np.sign(B[nz])).max() if nz.any() else 0.0, np.maximum(np.abs(G[~nz]) - lam * 0.5, 0).max()) / lam
    say(f"{seed} K={K} n={n} p={p} nnz/row={nnz_row}: default thresh {a.npasses:.0f} epochs ({ta:.2f}s), 1e-7 {b.npasses:.0f} epochs ({tb:.2f}s) "
        f"rc {int(np.sum(b.return_codes))}; max|d dev_ratio| {d.max():.2e}; KKT residual of the tight fit / lambda {kkt:.2e}"
        + ("  <-- CHECK" if d.max() > 3e-3 or kkt > 1e-3 else ""))
