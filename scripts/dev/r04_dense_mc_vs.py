"""Dense x with several classes: a short lambda path through sgdnet(mode="auto") with and without virtual shards
(round 4: the dense gather carries K-vector replicas).  usage: r04_dense_mc_vs.py [n p K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sgdnet_amd as sa
n, p, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 100, 4)
rng = np.random.default_rng(2)
x = rng.standard_normal((n, p))
B = rng.standard_normal((p, K)) * (rng.random((p, 1)) < 0.3)
lp = x @ B
y = (lp + rng.gumbel(size=lp.shape)).argmax(axis=1).astype(float)
kw = dict(family="multinomial", alpha=0.5, nlambda=int(os.environ.get("DMC_NLAMBDA", "20")), thresh=1e-5, maxit=200, mode="auto", seed=7)
if K == 1:
    y = (y > 0).astype(float)
    kw["family"] = "binomial"
sa.sgdnet(x[:20000], y[:20000], **dict(kw, nlambda=3))          # warm the library
res = {}
for V in [int(v) for v in os.environ.get("DMC_V", "0,-1").split(",")]:
    sa.set_option("virtual_shards", V)
    t = time.time(); fit = sa.sgdnet(x, y, **kw); dt = time.time() - t
    res[V] = fit
    print(f"dense {n}x{p} multinomial K={K}, {kw['nlambda']}-lambda path, virtual_shards={V}: {dt:.2f} s, npasses={fit.npasses:.0f}, "
          f"{dt / fit.npasses * 1e3:.2f} ms per epoch, dev.ratio[-1]={fit.dev_ratio[-1]:.6f}", flush=True)
sa.set_option("virtual_shards", -1)
a, b = list(res.values())[0], list(res.values())[-1]
if K == 1:
    a.beta, b.beta = [a.beta], [b.beta]
print(f"largest dev.ratio difference along the path {np.abs(np.asarray(a.dev_ratio) - np.asarray(b.dev_ratio)).max():.2e}; "
      f"largest coefficient difference at the last lambda "
      f"{max(np.abs(np.asarray(u)[:, -1] - np.asarray(v)[:, -1]).max() for u, v in zip(a.beta, b.beta)):.2e} "
      f"(largest |beta| {max(np.abs(np.asarray(u)[:, -1]).max() for u in a.beta):.3f})")
