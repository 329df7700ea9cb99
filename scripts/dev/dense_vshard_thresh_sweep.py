"""Large one-response DENSE fits (virtual shards on, dense intercept step): default thresh against thresh 1e-8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sgdnet_amd as sa
out = open(os.path.join(ROOT, "gpurun_out", "dvts.log"), "w")
def say(*a):
    print(*a, flush=True); print(*a, file=out, flush=True)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    r = np.random.default_rng(29000 + seed)
    family = ["binomial", "gaussian"][seed % 2]
    n = int(r.choice([250_000, 500_000]))
    p = int(r.choice([20, 100, 300]))
    corr = float(r.choice([0.0, 0.6, 0.9]))
    f = r.standard_normal((n, 1))
    x = (np.sqrt(1 - corr) * r.standard_normal((n, p)) + np.sqrt(corr) * f) * r.uniform(0.3, 3.0, p) + r.uniform(-1, 1, p)
    z = x[:, :5] @ r.uniform(-1, 1, 5) * 0.4 + 0.5
    y = (r.random(n) < 1 / (1 + np.exp(-z))).astype(float) if family == "binomial" else z + 0.5 * r.standard_normal(n)
    kw = dict(family=family, alpha=float(r.choice([0.5, 1.0])), standardize=bool(r.random() < 0.6), nlambda=10, mode="auto")
    t = time.time(); a = sa.sgdnet(x, y, seed=seed, **kw); ta = time.time() - t
    t = time.time(); b = sa.sgdnet(x, y, seed=seed, thresh=1e-8, maxit=2000, **kw); tb = time.time() - t
    d = np.abs(np.asarray(a.dev_ratio) - np.asarray(b.dev_ratio))
    say(f"{seed} {family} dense n={n} p={p} corr={corr} alpha={kw['alpha']} std={int(kw['standardize'])}: default thresh {a.npasses:.0f} epochs ({ta:.2f}s), "
        f"1e-8 {b.npasses:.0f} epochs ({tb:.2f}s) rc {int(np.sum(b.return_codes))}; max|d dev_ratio| {d.max():.2e} at lambda {int(d.argmax())}"
        + ("  <-- CHECK" if d.max() > 3e-3 or np.sum(b.return_codes) > 0 else ""))
