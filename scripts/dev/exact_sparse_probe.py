"""Per-iteration time of the sparse exact kernel (one response and multinomial)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
import sgdnet_amd as sa
from test_gpu_parity import make_problem

for family, K, n, p, dens, epochs, reg in [(f, K, n, p, dn, e, r) for (f, K, n, p, dn, e) in (
        ("binomial", 1, 20000, 200, 0.05, 3), ("gaussian", 1, 20000, 1000, 0.03, 3), ("binomial", 1, 200000, 10000, 0.001, 2),
        ("binomial", 1, 100000, 50000, 0.0002, 2), ("multinomial", 3, 20000, 200, 0.05, 3), ("multinomial", 3, 100000, 10000, 0.001, 2),
        ("multinomial", 10, 100000, 20000, 0.0005, 2), ("mgaussian", 4, 100000, 10000, 0.001, 2)) for r in ((0, 1, 3, 4) if K == 1 else (0, 1))]:
    sa.set_option("exact_row_registers", reg)
    x, y = make_problem(family, K, n, p, dens, seed=2)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    S.set_penalty("elasticnet", 0.02, 1e-3, 1e-3)
    S.upload_stream(sa.RRng(1).stream(n, n * (epochs + 1)))
    S.run(mode="exact", max_epochs=1, tol=0.0)
    S.sync()
    t = time.time()
    S.run(mode="exact", max_epochs=epochs, tol=0.0, stream_offset=n)
    S.sync()
    print(f"registers={reg} sparse {family} K={K} n={n} p={p} ({dens * p:.0f} nnz/row): {(time.time() - t) / (epochs * n) * 1e6:.2f} us per iteration",
          flush=True)
    S.close()
