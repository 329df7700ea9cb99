"""Explicit mode = "batched" when the window rule gives a handful of draws on many samples: epochs of tens of thousands of launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import sgdnet_amd as sa
r = np.random.default_rng(3)
n, p = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000, 10
f = r.standard_normal((n, 1))
x = (np.sqrt(0.1) * r.standard_normal((n, p)) + np.sqrt(0.9) * f) * r.uniform(0.5, 3.0, p)
y = (r.random(n) < 1 / (1 + np.exp(-(x[:, 0] - x[:, 1])))).astype(float)
for mode in ("auto", "batched"):
    t = time.time()
    fit = sa.sgdnet(x, y, family="binomial", alpha=0.5, nlambda=3, mode=mode, maxit=30)
    print(f"n={n} p={p} mode={mode}: {time.time() - t:.2f} s, npasses {fit.npasses:.0f}, rc {fit.return_codes}", flush=True)
