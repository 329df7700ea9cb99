"""Per-epoch cost of sgdnet_solver_run (one epoch per call, convergence check, host sync) next to
enqueue_epochs on the same problem."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
n, p, dens, seed = 1_000_000, 1_000, 0.01, 3
pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed); X = D.as_scipy(pr)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1]); col_sq = np.bincount(pr["idx"], weights=pr["val"] ** 2, minlength=p)
gamma = D.step_size(row_sq.max(), 0.5 / n, True, "binomial", n)
batch = sa.auto_batch(float(row_sq.max()), float(col_sq.max()) / n)
S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n)
V = 8; S.set_virtual_shards(V); draws = (n // V) * V
E = 60
S.upload_stream(S.sharded_stream([sa.RRng(seed + v) for v in range(V)], E + 2))
S.enqueue_epochs(1, batch=batch, stream_offset=0, draws_per_epoch=draws); S.sync()
t = time.time(); S.enqueue_epochs(E, batch=batch, stream_offset=draws, draws_per_epoch=draws); S.sync(); a = (time.time() - t) / E
t = time.time()
for e in range(E):
    S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=1, tol=0.0, stream_offset=draws * (1 + e % 2))
b = (time.time() - t) / E
t = time.time(); S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=E, tol=0.0, stream_offset=draws); c = (time.time() - t) / E
print(f"C3 V=8: enqueue_epochs {a*1e3:.3f} ms/epoch; run(1 epoch per call) {b*1e3:.3f} ms/epoch; run({E} epochs in one call) {c*1e3:.3f} ms/epoch")
