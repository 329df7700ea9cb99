"""Per-fit epoch counts of the documented cv example, oracle next to HIP (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refdocs_flow as F
import rcompat as R

d = F.load()
logs = {}
for be in (F.OracleBackend(), F.HipBackend("exact")):
    log = []
    orig = be.fit
    rng = be.rng(1)
    tr = rng.sample(270, 216)
    x = np.asarray(d["HX"][tr - 1].todense()); y = d["HY"][tr - 1]
    fits = [be.fit(x, y, rng, family="binomial", alpha=a) for a in (0.0, 1.0)]
    log += [("full", a, f.npasses) for a, f in zip((0, 1), fits)]
    from sgdnet_amd.cv import r_cut
    foldid = r_cut(rng.sample(216), 7)
    for i, a in enumerate((0.0, 1.0)):
        for j in range(1, 8):
            sel = foldid == j
            f = be.fit(x[sel], y[sel], rng, family="binomial", alpha=a, lambda_=fits[i].lambda_)
            log.append((f"fold{j}", a, f.npasses, float(f.dev_ratio[-1]), float(np.abs(f.beta).max())))
    logs[be.name] = log
for a, b in zip(logs["oracle"], logs["hip"]):
    print(a, b, "" if a[2] == b[2] else "  <<<<")
