"""C4 with V virtual shards on one GPU: ms/epoch and epochs to tolerance."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
Vs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
n, p, dens, seed = {"C4": (10_000_000, 10_000, 0.001, 4), "C3": (1_000_000, 1_000, 0.01, 3)}[wl]
pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed); X = D.as_scipy(pr)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
col_sq = np.bincount(pr["idx"], weights=pr["val"] ** 2, minlength=p)
gamma = D.step_size(row_sq.max(), 0.5 / n, True, "binomial", n)
batch = sa.auto_batch(float(row_sq.max()), float(col_sq.max()) / n)
ybar = pr["y"].mean(); b0 = np.array([np.log(ybar / (1 - ybar))])
S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n)
for V in Vs:
    S.set_virtual_shards(V if V > 1 else 0)
    draws = (n // V) * V
    for k, z in (("w", np.zeros((1, p))), ("g_sum", np.zeros((1, p))), ("g_sum_intercept", np.zeros(1)),
                 ("g_memory", np.zeros((1, n))), ("intercept", b0)):
        S.set(k, z)
    epochs = 45
    stream = S.sharded_stream([sa.RRng(seed + v) for v in range(max(V, 1))], epochs) if V > 1 else sa.RRng(seed).stream(n, n * epochs)
    S.upload_stream(stream)
    t = time.time(); ep, conv = S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=epochs, tol=1e-6); dt = time.time() - t
    dev = S.deviance()
    S.enqueue_epochs(1, batch=batch, stream_offset=0, draws_per_epoch=draws); S.sync()
    t = time.time(); S.enqueue_epochs(4, batch=batch, stream_offset=draws, draws_per_epoch=draws); S.sync(); de = (time.time() - t) / 4
    print(f"{wl} V={V} batch/shard={batch}: converged={conv} in {ep} epochs, deviance={dev:.8e}; {de*1e3:.3f} ms/epoch = {1/de:.1f} epochs/s", flush=True)
