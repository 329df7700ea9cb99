"""Tiled dense batched form (K*p beyond the LDS table): epochs/s and bytes of rows per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sgdnet_amd as sa

n, p = int(os.environ.get("N", 40000)), int(os.environ.get("P", 12000))
rng = np.random.default_rng(0)
X = rng.standard_normal((n, p))
w = rng.standard_normal(p) / np.sqrt(p)
z = X @ w
y = (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float64)
x = np.asfortranarray(X.T)
S = sa.SagaSolver(x, y, family="binomial", n_classes=1)
L = float((X[:2000] ** 2).sum(1).max())
S.set_penalty("elasticnet", 1.0 / (0.25 * L + 1e-4), 1e-4, 1e-4)
E_ = 6
S.upload_stream(sa.RRng(1).stream(n, n * E_))
for batch in (int(b) for b in os.environ.get("BATCHES", "2048,8192").split(",")):
    S.run(mode="batched", batch=batch, max_epochs=1, tol=0.0)
    S.sync()
    t = time.time()
    S.run(mode="batched", batch=batch, max_epochs=E_ - 1, tol=0.0, stream_offset=n)
    S.sync()
    dt = (time.time() - t) / (E_ - 1)
    prof = S.profile_epoch(batch=batch, draws_per_epoch=n)
    print(f"n={n} p={p} batch={batch}: {dt * 1e3:.2f} ms/epoch, {n * p * 8 / dt / 1e9:.0f} GB/s of rows (read twice: "
          f"{2 * n * p * 8 / dt / 1e9:.0f} GB/s); form {S._L.sgdnet_solver_gather_form(S._h, batch)}; "
          f"gather+accumulate {prof['gather_ms']:.2f} ms, sweep {prof['sweep_ms']:.2f} ms per epoch", flush=True)
S.close()
