import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from oracle import pyoracle as po
import sgdnet_amd as sa
rng = np.random.default_rng(14)
n, p = 400, 6
x = rng.standard_normal((n, p)) * (rng.random((n, p)) < 0.5)
z = x @ rng.uniform(-1, 1, (p, 3)) + 0.3
ys = {"gaussian": z[:, 0] + 0.1 * rng.standard_normal(n),
     "binomial": (rng.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
     "multinomial": np.argmax(z + rng.gumbel(size=z.shape), axis=1).astype(float),
     "mgaussian": z[:, :2] + 0.1 * rng.standard_normal((n, 2))}
y = ys["binomial"]
kw = dict(family="binomial", alpha=0.6, nlambda=6, thresh=1e-5, standardize=False)
fit = sa.sgdnet(x, y, seed=3, debug=True, **kw)
ref = oracle = po.fit(x, y, seed=3, debug=True, **kw)
for i,(a,b) in enumerate(zip(fit.diagnostics["loss"], ref["losses"])):
    m=min(len(a),len(b)); d=np.abs(a[:m]-b[:m])/np.abs(b[:m])
    print(i, len(a), len(b), "first dev>1e-12 at", (np.argmax(d>1e-12) if (d>1e-12).any() else None), d.max())
# solver-level: replicate lambda 0 epoch by epoch, checking the convergence ratio
xt = np.asfortranarray(x.T); yy=np.asfortranarray(y.reshape(1,-1))
lam=ref["lambda"]; 
for li in range(2):
    pass
S = sa.SagaSolver(xt, yy, family="binomial", n_classes=1)
st = po.new_state(1,p,n)
b0=np.array([np.log(y.mean()/(1-y.mean()))]); S.set("intercept", b0); st["intercept"][:]=b0
r = po.Rng(3)
for li in range(6):
    g, a, b = ref["step_size"][li], ref["alpha_l2"][li], ref["beta_l1"][li]
    S.set_penalty("elasticnet", g, a, b)
    for ep in range(60):
        stream = r.stream(n, n)
        wprev = st["w"].copy()
        e1, rc, _ = po.saga(xt, yy, st, family="binomial", penalty="elasticnet", gamma=g, alpha=a, beta=b, max_iter=1, tol=1e-5, stream=stream)
        S.upload_stream(stream)
        e2, conv = S.run(mode="exact", max_epochs=1, tol=1e-5)
        wg = S.get("w")
        ratio_o = np.abs(st["w"]-wprev).max()/max(np.abs(st["w"]).max(),1e-300)
        ratio_g = np.abs(wg-wprev).max()/max(np.abs(wg).max(),1e-300)
        err = np.abs(wg-st["w"]).max()
        if (rc==0) != conv or err>1e-12 or rc==0:
            print("lam",li,"ep",ep,"oracle conv",rc==0,"gpu conv",conv,"ratio",ratio_o,ratio_g,"err",err, flush=True)
        if rc==0 or conv: break
