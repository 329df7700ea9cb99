"""Reproduces a case of tests/test_gpu_fuzz.py::test_random_fit_in_batched_mode_reaches_the_oracle_optimum with tracing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa
from oracle import pyoracle as po

for seed in [int(a) for a in sys.argv[1:]]:
    r = np.random.default_rng(7000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n = int(r.choice([300, 1000, 2000])); p = int(r.choice([5, 20, 60, 150])); sparse = bool(r.random() < 0.6)
    x = r.standard_normal((n, p)) * r.uniform(0.5, 2.0, p) * (r.random((n, p)) < (0.3 if sparse else 1.0))
    x[np.arange(n), r.integers(0, p, n)] += 0.7
    z = x[:, : min(p, 4)] @ r.uniform(-1, 1, (min(p, 4), 3)) + 0.2
    y = {"gaussian": z[:, 0] + 0.1 * r.standard_normal(n),
         "binomial": (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.1 * r.standard_normal((n, 2))}[family]
    if family in ("binomial", "multinomial"):
        y[: 3] = [0, 1, 2 if family == "multinomial" else 1]
    xx = sp.csc_matrix(x) if sparse else x
    kw = dict(family=family, alpha=float(r.choice([0.2, 0.7, 1.0])), standardize=bool(r.random() < 0.5))
    ref0 = po.fit(xx, y, seed=seed, nlambda=5, maxit=1, **kw)
    lam = ref0["lambda"][[1, 3]]
    ref = po.fit(xx, y, seed=seed, lambda_=lam, thresh=1e-9, maxit=3000, **kw)
    print("case", seed, kw, n, p, sparse, "oracle npasses", ref["npasses"], ref["return_codes"], flush=True)
    for batch in (0, 32, 8):
        fit = sa.sgdnet(xx, y, seed=seed, lambda_=lam, thresh=1e-9, mode="auto", maxit=3000, batch=batch, **kw)
        beta = np.stack(fit.beta) if isinstance(fit.beta, list) else fit.beta[None]
        print("  batch", batch, "rc", fit.return_codes, "npasses", fit.npasses, "err", np.abs(beta - ref["beta"]).max() / np.abs(ref["beta"]).max(), flush=True)
