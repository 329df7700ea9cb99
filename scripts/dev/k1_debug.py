"""Differences between the register-resident sparse exact kernel and the det-math oracle, per state array."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import sgdnet_amd as sa
from oracle import pyoracle as po
from test_gpu_parity import make_problem, run_both, STATE
po.use_det_math(True)
for family, penalty, n, p, dens, epochs in (("gaussian", "ridge", 300, 80, 0.06, 1), ("gaussian", "elasticnet", 300, 80, 0.06, 1),
                                            ("binomial", "ridge", 300, 80, 0.06, 1), ("binomial", "elasticnet", 300, 80, 0.06, 1),
                                            ("binomial", "elasticnet", 1500, 80, 0.06, 4)):
    x, y = make_problem(family, 1, n, p, dens, seed=2)
    for reg in (1, 2):
        sa.set_option("exact_row_registers", reg)
        a, b = (1e-3, 0.0) if penalty == "ridge" else (5e-4, 5e-4)
        ref, got = run_both(sa, po, x, y, family=family, K=1, penalty=penalty, gamma=0.05, alpha=a, beta=b, epochs=epochs, mode="exact", seed=3)
        out = []
        for name in STATE:
            r_, g_ = np.asarray(ref[2][name]).ravel(), np.asarray(got[2][name]).ravel()
            bad = np.flatnonzero(r_ != g_)
            out.append(f"{name}: {bad.size}/{r_.size} differ, max {np.max(np.abs(r_ - g_)) if bad.size else 0:.3g}")
        print(family, penalty, n, epochs, "registers", reg, "|", "; ".join(out), flush=True)
