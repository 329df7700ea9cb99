"""The 500000 x 100 dense gaussian fit whose first lambda blew up under the rule's window: with and without virtual shards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sgdnet_amd as sa
seed = 1
r = np.random.default_rng(29000 + seed)
family = ["binomial", "gaussian"][seed % 2]
n = int(r.choice([250_000, 500_000])); p = int(r.choice([20, 100, 300])); corr = float(r.choice([0.0, 0.6, 0.9]))
f = r.standard_normal((n, 1))
x = (np.sqrt(1 - corr) * r.standard_normal((n, p)) + np.sqrt(corr) * f) * r.uniform(0.3, 3.0, p) + r.uniform(-1, 1, p)
z = x[:, :5] @ r.uniform(-1, 1, 5) * 0.4 + 0.5
y = z + 0.5 * r.standard_normal(n)
kw = dict(family=family, alpha=float(r.choice([0.5, 1.0])), standardize=bool(r.random() < 0.6), nlambda=10, mode="auto")
print(n, p, corr, kw, flush=True)
t = time.time(); a = sa.sgdnet(x, y, seed=seed, **kw); print("fit", time.time() - t, a.npasses, flush=True)
