"""Dense x with 17..64 classes: the class-lane form of the dense kernels (round 4) against the route of rounds 2-3
(the same matrix as a CSC matrix with every entry stored -> the sparse binned form).  usage: r04_dense_cl.py [n p K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import sgdnet_amd as sa
n, p, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (200_000, 100, 32)
rng = np.random.default_rng(3)
x = rng.standard_normal((n, p))
B = rng.standard_normal((p, K)) * (rng.random((p, 1)) < 0.3)
y = (x @ B + rng.gumbel(size=(n, K))).argmax(axis=1).astype(float)
kw = dict(family="multinomial", alpha=0.5, nlambda=8, thresh=1e-5, maxit=300, mode="auto", seed=7, standardize=False)
sa.sgdnet(x[:5000], y[:5000], **dict(kw, nlambda=2))
fits = {}
for name, xx in (("dense (class-lane form)", x), ("CSC with every entry stored (binned form)", sp.csc_matrix(x))):
    t = time.time(); fit = sa.sgdnet(xx, y, **kw); dt = time.time() - t
    fits[name] = fit
    print(f"{n}x{p} multinomial K={K}, 8-lambda path, {name}: {dt:.2f} s, npasses={fit.npasses:.0f}, "
          f"{dt / fit.npasses * 1e3:.2f} ms per epoch, dev.ratio[-1]={fit.dev_ratio[-1]:.6f}", flush=True)
a, b = fits.values()
print(f"largest dev.ratio difference along the path {np.abs(np.asarray(a.dev_ratio) - np.asarray(b.dev_ratio)).max():.2e}")
