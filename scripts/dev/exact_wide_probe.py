"""Per-iteration time of the exact dense kernels (run once with SGDNET_EXACT_WIDE=0, once with 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
import sgdnet_amd as sa
from test_gpu_parity import make_problem

for family, K, n, p, epochs in (("gaussian", 1, 1500, 10500, 3), ("binomial", 1, 2000, 1000, 5), ("binomial", 1, 5000, 100, 5),
                                ("multinomial", 3, 3000, 50, 5), ("multinomial", 10, 2000, 400, 3), ("mgaussian", 4, 2000, 2000, 2)):
    if os.environ.get("SGDNET_EXACT_WIDE") == "0" and p > 5000:
        epochs = 1
    x, y = make_problem(family, K, n, p, None, seed=2, dense=True)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    pen = "grouplasso" if family == "mgaussian" else "elasticnet"
    S.set_penalty(pen, 0.3 / p, 1e-3, 1e-3)
    S.upload_stream(sa.RRng(1).stream(n, n * (epochs + 1)))
    S.run(mode="exact", max_epochs=1, tol=0.0)
    S.sync()
    t = time.time()
    S.run(mode="exact", max_epochs=epochs, tol=0.0, stream_offset=n)
    S.sync()
    dt = time.time() - t
    print(f"{family} K={K} n={n} p={p}: {dt / (epochs * n) * 1e6:.2f} us per iteration", flush=True)
    S.close()
