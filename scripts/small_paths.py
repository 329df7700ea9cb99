"""Wall time of the reference's small dense configurations (BASELINE configs 1-2) per mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ab = np.load(os.path.join(GOLD, "abalone.npz"))
ir = np.load(os.path.join(GOLD, "iris.npz"))
cases = [("abalone 4177x%d gaussian" % ab["x"].shape[1], ab["x"], ab["y"], dict(family="gaussian")),
         ("iris 150x4 multinomial alpha=0.8", ir["x"], ir["y"], dict(family="multinomial", alpha=0.8))]
for name, x, y, kw in cases:
    for mode in ("exact", "auto", "batched"):
        sa.sgdnet(x, y, nlambda=3, mode=mode, **kw)
        t = time.time(); fit = sa.sgdnet(x, y, mode=mode, seed=1, **kw); dt = time.time() - t
        print(f"{name}: mode={mode}: {dt:.3f} s, {len(fit.lambda_)} lambdas, npasses={fit.npasses:.0f}, "
              f"{dt / max(fit.npasses, 1) * 1e3:.3f} ms/epoch", flush=True)
