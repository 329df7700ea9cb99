"""Virtual shards for 2..16 classes (round 3): epochs/s of a mid-size multinomial fit with and without them.
Usage: vshard_multiclass_probe.py [n_classes=3] [n_features=1000]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
import sgdnet_amd as sa
from sgdnet_amd import data as D

n, dens = 1_000_000, 0.01
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pr = D.make_sparse_glm(n, p, dens, family="multinomial", n_classes=K, seed=3)
X = D.as_scipy(pr)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
col_sq = np.bincount(pr["idx"], weights=pr["val"] ** 2, minlength=p)
lam = 1.0 / n
gamma = D.step_size(row_sq.max(), 0.5 * lam, True, "multinomial", n)
batch = sa.auto_batch(float(row_sq.max()), float(col_sq.max()) / n)
for V in (0, 2, 4, 8):
    S = sa.SagaSolver(X, pr["y"], family="multinomial", n_classes=K)
    S.set_penalty("elasticnet", gamma, 0.5 * lam, 0.5 * lam)
    if V:
        S.set_virtual_shards(V)
    rng = sa.RRng(3)
    S.convergence(1e-6)
    S.generate_stream(rng, n)
    S.enqueue_epochs(1, batch=batch, draws_per_epoch=(V or 1) * (n // (V or 1)))
    S.sync()
    t0 = time.perf_counter()
    ep, done = 0, False
    while not done and ep < 200:
        S.generate_stream(rng, n)
        S.enqueue_epochs(1, batch=batch, draws_per_epoch=(V or 1) * (n // (V or 1)))
        S.sync()
        done = S.convergence(1e-6)
        ep += 1
    dt = time.perf_counter() - t0
    print(f"K={K} p={p} V={V}: window {batch}, {ep} epochs to 1e-6 in {dt:.3f} s = {1e3 * dt / ep:.3f} ms/epoch (incl. host sync + stream), deviance {S.deviance():.6f}", flush=True)
    if V:
        S.set_virtual_shards(0)
    S.close()
