"""One configuration of the sparse exact kernel, for counter collection: k1_one.py <registers> [n p density]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import sgdnet_amd as sa
from test_gpu_parity import make_problem
reg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dens = float(sys.argv[4]) if len(sys.argv) > 4 else 0.001
family = sys.argv[5] if len(sys.argv) > 5 else "binomial"
sa.set_option("exact_row_registers", reg)
x, y = make_problem(family, 1, n, p, dens, seed=2)
S = sa.SagaSolver(x, y, family=family, n_classes=1)
S.set_penalty("elasticnet", 0.02, 1e-3, 1e-3)
S.upload_stream(sa.RRng(1).stream(n, n * 2))
S.sync()
t = time.time()
S.run(mode="exact", max_epochs=2, tol=0.0)
S.sync()
print(f"registers={reg} {family} n={n} p={p}: {(time.time() - t) / (2 * n) * 1e6:.2f} us per draw ({2 * n} draws)", flush=True)
S.close()
