"""Large one-response sparse fits (virtual shards on): deviance ratio at the default thresh against thresh 1e-8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa

out = open(os.path.join(ROOT, "gpurun_out", "vts.log"), "w")
def say(*a):
    print(*a, flush=True); print(*a, file=out, flush=True)

for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    r = np.random.default_rng(17000 + seed)
    family = ["binomial", "gaussian"][seed % 2]
    n = int(r.choice([250_000, 600_000, 1_200_000]))
    p = int(r.choice([100, 1000, 5000]))
    nnz_row = int(r.choice([3, 10, 30]))
    rows = np.repeat(np.arange(n), nnz_row)
    cols = r.integers(0, p, n * nnz_row)
    vals = r.standard_normal(n * nnz_row) * r.uniform(0.3, 2.0, p)[cols]
    if seed % 3 == 0:
        vals = np.abs(vals)                                        # non-negative features
    X = sp.csc_matrix((vals, (rows, cols)), shape=(n, p))
    X.sum_duplicates()
    w = r.standard_normal(p) * (r.random(p) < 0.2)
    z = X @ w * 0.3 + 0.3
    y = (r.random(n) < 1 / (1 + np.exp(-z))).astype(float) if family == "binomial" else z + 0.5 * r.standard_normal(n)
    kw = dict(family=family, alpha=float(r.choice([0.5, 1.0])), standardize=bool(r.random() < 0.5), nlambda=12, mode="auto")
    t = time.time(); a = sa.sgdnet(X, y, seed=seed, **kw); ta = time.time() - t
    t = time.time(); b = sa.sgdnet(X, y, seed=seed, thresh=1e-8, maxit=3000, **kw); tb = time.time() - t
    d = np.abs(np.asarray(a.dev_ratio) - np.asarray(b.dev_ratio))
    say(f"{seed} {family} n={n} p={p} nnz/row={nnz_row} alpha={kw['alpha']} std={int(kw['standardize'])}: default thresh {a.npasses:.0f} epochs ({ta:.2f}s), "
        f"1e-8 {b.npasses:.0f} epochs ({tb:.2f}s) rc {int(np.sum(b.return_codes))}; max|d dev_ratio| {d.max():.2e} at lambda {int(d.argmax())}"
        + ("  <-- CHECK" if d.max() > 3e-3 else ""))
