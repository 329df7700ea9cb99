"""Dense x: batched (wave per draw) vs exact mode vs the CPU oracle, epochs/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 100)
rng = np.random.default_rng(1)
x = np.asfortranarray(rng.standard_normal((p, n)))           # p x n: a sample is contiguous
beta = rng.standard_normal(p) * (rng.random(p) < 0.3)
y = (rng.random(n) < 1 / (1 + np.exp(-(beta @ x)))).astype(float).reshape(1, n)
row_sq = (x ** 2).sum(axis=0)
gamma = D.step_size(row_sq.max(), 0.5 / n, True, "binomial", n)
batch = sa.auto_batch(float(row_sq.max()), float((x ** 2).mean(axis=1).max()))
stream = sa.RRng(1).stream(n, n * 4)
S = sa.SagaSolver(x, y, family="binomial", n_classes=1)
S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n)
S.upload_stream(stream)
S.run(mode="batched", batch=batch, max_epochs=1, tol=0.0)
t = time.time(); S.run(mode="batched", batch=batch, max_epochs=3, tol=0.0, stream_offset=n); dt = (time.time() - t) / 3
print(f"dense {n}x{p}: batched (batch {batch}) {dt*1e3:.2f} ms/epoch = {n/dt/1e6:.1f} M draws/s, "
      f"{8*p*n/dt/1e9:.0f} GB/s of rows", flush=True)
if n <= 200_000:
    t = time.time(); S.run(mode="exact", max_epochs=1, tol=0.0); dt = time.time() - t
    print(f"  exact mode {dt*1e3:.1f} ms/epoch = {n/dt/1e6:.2f} M draws/s", flush=True)
    from oracle import pyoracle as po
    st = po.new_state(1, p, n)
    t = time.time(); po.saga(x, y, st, family="binomial", penalty="elasticnet", gamma=gamma, alpha=0.5 / n, beta=0.5 / n,
                             max_iter=1, tol=0.0, stream=stream[:n]); dt = time.time() - t
    print(f"  CPU oracle {dt*1e3:.1f} ms/epoch = {n/dt/1e6:.2f} M draws/s", flush=True)
# virtual shards: V replicas over sample ranges, one launch per batch of all shards
for V in [int(v) for v in os.environ.get("DENSE_VSHARDS", "2,4,8").split(",") if v]:
    if 2 * V * 100 * p > n * 2:
        continue
    S.set_virtual_shards(V)
    for k, z in (("w", np.zeros((1, p))), ("g_sum", np.zeros((1, p))), ("g_sum_intercept", np.zeros(1)),
                 ("g_memory", np.zeros((1, n))), ("intercept", np.zeros(1))):
        S.set(k, z)
    draws = (n // V) * V
    S.upload_stream(S.sharded_stream([sa.RRng(1 + v) for v in range(V)], 4))
    S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=1, tol=0.0)
    t = time.time(); S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=3, tol=0.0, stream_offset=draws)
    dt = (time.time() - t) / 3
    print(f"  {V} virtual shards: {dt*1e3:.2f} ms/epoch = {n/dt/1e6:.1f} M draws/s, {8*p*n/dt/1e9:.0f} GB/s of rows",
          flush=True)
