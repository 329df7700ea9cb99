"""Epochs-to-tolerance of the batched mode as a function of the staleness window (C4 shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
batches = [int(b) for b in sys.argv[2].split(",")]
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
n, p, dens, seed = {"C4": (10_000_000, 10_000, 0.001, 4), "C3": (1_000_000, 1_000, 0.01, 3)}[wl]
pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed); X = D.as_scipy(pr)
maxit = 120
stream = sa.RRng(seed).stream(n, n * maxit) if n * maxit < 1_300_000_000 else None
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
gamma = D.step_size(row_sq.max(), 0.5 / n, True, "binomial", n)
ybar = pr["y"].mean(); b0 = np.array([np.log(ybar / (1 - ybar))])
ref = None
for batch in batches:
    S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
    S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n); S.set("intercept", b0)
    S.upload_stream(stream)
    t = time.time(); ep, conv = S.run(mode="batched", batch=batch, max_epochs=maxit, tol=tol); dt = time.time() - t
    w = S.get("w"); dev = S.deviance()
    if ref is None: ref = w
    print(f"batch={batch:7d}: epochs={ep:4d} converged={conv} time={dt:6.2f}s deviance={dev:.10e} |w|max={np.abs(w).max():.4f} "
          f"rel diff vs first={np.abs(w-ref).max()/np.abs(ref).max():.2e} nnz(w)={np.count_nonzero(w)}", flush=True)
    S.close()
