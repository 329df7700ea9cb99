import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ab = np.load(os.path.join(GOLD, "abalone.npz"))
x, y = ab["x"], ab["y"]
rng = np.random.default_rng(0)
xs = rng.standard_normal(x.shape); ys = xs @ rng.standard_normal(x.shape[1]) + rng.standard_normal(len(y))
for name, xx, yy in (("abalone", x, y), ("synthetic", xs, ys)):
    for V in ("0", "2"):
        os.environ["SGDNET_VSHARDS"] = V
        for batch in (0, 256, 64, 16):
            for nl in (100, 1):
                try:
                    kw = dict(nlambda=nl) if nl > 1 else dict(lambda_=[0.01])
                    fit = sa.sgdnet(xx, yy, family="gaussian", mode="batched", batch=batch, seed=1, **kw)
                    print(f"{name} V={V} batch={batch} nlambda={nl}: ok npasses={fit.npasses:.0f} rc={fit.return_codes[-1]}", flush=True)
                except Exception as e:
                    print(f"{name} V={V} batch={batch} nlambda={nl}: FAILED {str(e)[:80]}", flush=True)
