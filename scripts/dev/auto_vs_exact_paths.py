"""Random 30-lambda paths at the default thresh: mode = auto against the exact iteration (same data, same lambdas).
Prints the worst deviation of the deviance ratio along the path and flags non-converged lambdas."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa

bad = 0
seeds = [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else range(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
for seed in seeds:
    r = np.random.default_rng(9000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n = int(r.choice([500, 2000, 5000]))
    p = int(r.choice([8, 40, 200, 1000]))
    sparse = bool(r.random() < 0.6)
    corr = float(r.choice([0.0, 0.0, 0.7, 0.95]))                 # common factor: collinear features
    f = r.standard_normal((n, 1))
    x = (np.sqrt(1 - corr) * r.standard_normal((n, p)) + np.sqrt(corr) * f) * r.uniform(0.3, 3.0, p)
    if sparse:
        x = x * (r.random((n, p)) < float(r.choice([0.02, 0.1, 0.4])))
        x[np.arange(n), r.integers(0, p, n)] += 0.7
    else:
        x = x + r.uniform(-2, 2, p)                                # dense: non-zero means
    k0 = min(p, 6)
    z = x[:, :k0] @ r.uniform(-1, 1, (k0, 3)) + 0.2
    y = {"gaussian": z[:, 0] + 0.3 * r.standard_normal(n),
         "binomial": (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.3 * r.standard_normal((n, 2))}[family]
    xx = sp.csc_matrix(x) if sparse else x
    kw = dict(family=family, alpha=float(r.choice([0.0, 0.5, 1.0])), standardize=bool(r.random() < 0.6), nlambda=30)
    t = time.time(); ex = sa.sgdnet(xx, y, seed=seed, **kw); te = time.time() - t
    t = time.time(); au = sa.sgdnet(xx, y, seed=seed, mode="auto", **kw); ta = time.time() - t
    if os.environ.get("TIGHT"):
        # ground truth: the exact iteration at a tight threshold on the same lambdas
        tr = sa.sgdnet(xx, y, seed=seed, thresh=1e-7, maxit=20000, **kw)
        de, da = np.abs(np.asarray(ex.dev_ratio) - np.asarray(tr.dev_ratio)), np.abs(np.asarray(au.dev_ratio) - np.asarray(tr.dev_ratio))
        print(f"    against thresh 1e-7 (npasses {tr.npasses:.0f}, rc {int(np.sum(tr.return_codes))}): exact off by {de.max():.2e} (lambda {int(de.argmax())}), "
              f"auto off by {da.max():.2e} (lambda {int(da.argmax())})", flush=True)
    d = np.abs(np.asarray(au.dev_ratio) - np.asarray(ex.dev_ratio))
    flag = ""
    if d.max() > 5e-3 or np.any(np.asarray(au.return_codes) != 0) and not np.any(np.asarray(ex.return_codes) != 0):
        flag = "  <-- CHECK"
        bad += 1
    print(f"{seed:3d} {family:11s} n={n:5d} p={p:4d} {'sparse' if sparse else 'dense '} corr={corr:.2f} alpha={kw['alpha']:.1f} std={int(kw['standardize'])}: "
          f"max|d dev_ratio| {d.max():.2e} at lambda {int(d.argmax())}, npasses exact {ex.npasses:.0f} ({te:.2f}s) auto {au.npasses:.0f} ({ta:.2f}s), "
          f"rc exact {int(np.sum(ex.return_codes))} auto {int(np.sum(au.return_codes))}{flag}", flush=True)
print("flagged", bad)
