"""Replays one case of tests/test_gpu_fuzz.py::test_random_problem_matches_the_oracle draw by draw: the first
epoch is cut after t draws (draws_per_epoch = t), HIP against the oracle, to find where they part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
import torch
import sgdnet_amd as sa
from oracle import pyoracle as po
import test_gpu_fuzz as F

STATE = ("w", "intercept", "g_sum", "g_memory", "g_sum_intercept")
for seed in [int(a) for a in sys.argv[1:]]:
    c = F._case(seed)
    r = np.random.default_rng(seed)
    n, p, K, family = c["n"], c["p"], c["K"], c["family"]
    dens = 1.0 if c["dense"] else float(r.choice([0.05, 0.3, 0.9]))
    X = r.standard_normal((p, n)) * (r.random((p, n)) < dens)
    if not c["dense"]:
        X[:, X.any(axis=0) == 0] = 0.0
        X[r.integers(0, p, n), np.arange(n)] += 0.5
    B = r.standard_normal((K, p)) * 0.5
    lp = B @ X
    if family == "gaussian":
        y = lp[0:1] + 0.1 * r.standard_normal((1, n))
    elif family == "binomial":
        y = (r.random((1, n)) < 1 / (1 + np.exp(-lp[0:1]))).astype(float)
    elif family == "multinomial":
        y = np.argmax(lp + r.gumbel(size=lp.shape), axis=0).astype(float).reshape(1, n)
    else:
        y = lp + 0.1 * r.standard_normal(lp.shape)
    x = np.asfortranarray(X) if c["dense"] else sp.csc_matrix(X)
    y = np.asfortranarray(y)
    L = float((X ** 2).sum(axis=0).max()) + 1.0
    gamma = 0.3 / L
    a, b = (2e-3, 0.0) if c["penalty"] == "ridge" else (1e-3, 2e-3)
    if c["heavy_ridge"]:
        a = 0.3 / gamma
    cvec = r.normal(0, 0.05, p) if c["centre"] else None
    stream = po.Rng(seed).stream(n, n * 2)
    if n > 8:
        stream[3:6] = stream[3]
    print("case", c, "nnz per sample", (X != 0).sum(axis=0)[:12], "stream", stream[:10], flush=True)
    for t in list(range(1, min(n, 12) + 1)) + [n]:
        st = po.new_state(K, p, n)
        xo = sp.csc_matrix(X) if (c["dense"] and c["mode"] == "batched") else x
        kw = dict(family=family, penalty=c["penalty"], gamma=gamma, alpha=a, beta=b, fit_intercept=c["fit_intercept"],
                  x_center_scaled=cvec)
        m = c["batch"]
        for i0 in range(0, t, m):                                    # the oracle's batch halves, batch by batch
            dr = stream[i0:min(t, i0 + m)]
            Dm, d0 = np.zeros((K, p), order="F"), np.zeros(K)
            po.batch_gather(xo, y, st, dr, Dm, d0, n_total=n, **kw)
            po.batch_sweep((p, n), st, dr.size, Dm, d0, n_total=n, **kw)
        S = sa.SagaSolver(x, y, family=family, n_classes=K, fit_intercept=c["fit_intercept"], x_center_scaled=cvec)
        S.set_penalty(c["penalty"], gamma, a, b)
        S.upload_stream(stream)
        S.run(mode=c["mode"], batch=c["batch"], max_epochs=1, tol=0.0, draws_per_epoch=t)
        errs = {k: float(np.abs(np.asarray(S.get(k)) - np.asarray(st[k])).max() / max(1e-300, np.abs(st[k]).max())) for k in STATE}
        print(" t", t, {k: f"{v:.1e}" for k, v in errs.items()}, flush=True)
        if t <= 3 and max(errs.values()) > 1e-9:
            print("   hip w", np.asarray(S.get("w")).ravel()[:8], "b", S.get("intercept"))
            print("   orc w", np.asarray(st["w"]).ravel()[:8], "b", st["intercept"])
        S.close()
