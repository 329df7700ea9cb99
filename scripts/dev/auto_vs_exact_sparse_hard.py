"""Second path sweep, sparse x with structure that stresses the window rule: binary (non-negative) features, a frequent
bias-like column, duplicated columns, a few rows scaled 30x, p > n.  mode = auto against the exact iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa

seeds = [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else range(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
bad = 0
for seed in seeds:
    r = np.random.default_rng(13000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n = int(r.choice([300, 1000, 2500]))
    p = int(r.choice([30, 200, 800]))
    dens = float(r.choice([0.01, 0.05, 0.2]))
    kind = ["gauss", "binary", "bias", "dup", "outlier", "counts"][seed % 6]
    m = (r.random((n, p)) < dens)
    x = r.standard_normal((n, p)) * m
    if kind == "binary":
        x = m.astype(float)
    elif kind == "counts":
        x = r.poisson(3.0, (n, p)) * m * 1.0
    elif kind == "bias":
        x[:, 0] = 1.0 + 0.1 * r.standard_normal(n)
        x[:, 1] = (r.random(n) < 0.8) * 2.0
    elif kind == "dup":
        x[:, p // 2:p // 2 + min(10, p // 2)] = x[:, :min(10, p // 2)]
    elif kind == "outlier":
        rows = r.choice(n, max(1, n // 200), replace=False)
        x[rows] *= 30.0
    x[np.arange(n), r.integers(0, p, n)] += 0.7
    k0 = min(p, 6)
    z = x[:, :k0] @ r.uniform(-1, 1, (k0, 3)) * 0.5 + 0.2
    y = {"gaussian": z[:, 0] + 0.3 * r.standard_normal(n),
         "binomial": (r.random(n) < 1 / (1 + np.exp(-np.clip(z[:, 0], -30, 30)))).astype(float),
         "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.3 * r.standard_normal((n, 2))}[family]
    if family in ("binomial", "multinomial"):
        y[:3] = [0, 1, 2 if family == "multinomial" else 1]
    xx = sp.csc_matrix(x)
    kw = dict(family=family, alpha=float(r.choice([0.0, 0.5, 1.0])), standardize=bool(r.random() < 0.6), nlambda=25)
    print(f"{seed:3d} start {family} {kind} n={n} p={p} dens={dens}", flush=True)
    t = time.time(); ex = sa.sgdnet(xx, y, seed=seed, **kw); te = time.time() - t
    t = time.time(); au = sa.sgdnet(xx, y, seed=seed, mode="auto", **kw); ta = time.time() - t
    d = np.abs(np.asarray(au.dev_ratio) - np.asarray(ex.dev_ratio))
    flag = ""
    if d.max() > 5e-3 or (np.any(np.asarray(au.return_codes) != 0) and not np.any(np.asarray(ex.return_codes) != 0)):
        flag = "  <-- CHECK"; bad += 1
    print(f"{seed:3d} {family:11s} {kind:8s} n={n:5d} p={p:4d} dens={dens:.2f} alpha={kw['alpha']:.1f} std={int(kw['standardize'])}: "
          f"max|d dev_ratio| {d.max():.2e} at lambda {int(d.argmax())}, npasses exact {ex.npasses:.0f} ({te:.2f}s) auto {au.npasses:.0f} ({ta:.2f}s), "
          f"rc exact {int(np.sum(ex.return_codes))} auto {int(np.sum(au.return_codes))}{flag}", flush=True)
print("flagged", bad)
