"""KKT residual of the C4 bench configuration vs epochs (GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import sgdnet_amd as sa
from sgdnet_amd import data as D

wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
n, p, dens, seed = {"C3": (1_000_000, 1_000, 0.01, 3), "C4": (10_000_000, 10_000, 0.001, 4)}[wl]
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed)
X = D.as_scipy(pr)            # p x n
Xt = X.T.tocsr()              # n x p
y = pr["y"].ravel()
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
col_sq = np.bincount(pr["idx"], weights=pr["val"] ** 2, minlength=p)
lam = 1.0 / n
a_l2 = b_l1 = 0.5 * lam
gamma = D.step_size(row_sq.max(), a_l2, True, "binomial", n)
batch = sa.auto_batch(float(row_sq.max()), float(col_sq.max()) / n)
S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
S.set_penalty("elasticnet", gamma, a_l2, b_l1)
ybar = y.mean(); S.set("intercept", np.array([np.log(ybar / (1 - ybar))]))
if V > 1: S.set_virtual_shards(V)
rng = sa.RRng(seed + 1000)
draws = V * (n // V) if V > 1 else n

def kkt():
    w = S.get("w")[0]; b = S.get("intercept")[0]
    lp = Xt @ w + b
    r = 1.0 / (1.0 + np.exp(-lp)) - y
    g = (X @ r) / n + a_l2 * w
    res = np.where(w == 0, np.maximum(np.abs(g) - b_l1, 0.0), np.abs(g + b_l1 * np.sign(w)))
    return res.max() / lam, abs(r.sum()) / n / lam, int((w != 0).sum())

S.convergence(0.0)
t0 = time.time(); ep = 0
for target in (25, 50, 100, 150, 200, 300, 400, 600):
    while ep < target:
        S.generate_stream(rng, n)
        S.enqueue_epochs(1, batch=batch, draws_per_epoch=draws)
        ep += 1
    S.sync()
    ch = S.last_change() if hasattr(S, "last_change") else None
    print(ep, "epochs: KKT residual / lambda = %.3e, intercept %.3e, nnz %d" % kkt(), "t=%.1fs" % (time.time() - t0), flush=True)
