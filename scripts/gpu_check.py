"""First-contact GPU check: exact + batched kernels against the CPU oracle, then timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from oracle import pyoracle as po
import sgdnet_amd as sa
from sgdnet_amd import data as D

def relerr(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))

def check_sparse(family, K, penalty, n=3000, p=200, dens=0.05, epochs=3, batch=0, seed=1, fit_intercept=True):
    pr = D.make_sparse_glm(n, p, dens, family=family, n_classes=K, seed=seed)
    X = D.as_scipy(pr)
    stream = po.Rng(seed).stream(n, n * epochs)
    gamma, alpha, beta = 0.02, 1e-3, (2e-3 if penalty != "ridge" else 0.0)
    st = po.new_state(K, p, n)
    ep, rc, _ = po.saga(X, pr["y"], st, family=family, penalty=penalty, gamma=gamma, alpha=alpha, beta=beta,
                        fit_intercept=fit_intercept, max_iter=epochs, tol=0.0, stream=stream, batch=batch)
    S = sa.SagaSolver(X, pr["y"], family=family, n_classes=K, fit_intercept=fit_intercept)
    S.set_penalty(penalty, gamma, alpha, beta)
    S.upload_stream(stream)
    t = time.time()
    ep2, conv = S.run(mode="batched" if batch else "exact", batch=batch, max_epochs=epochs, tol=0.0)
    dt = time.time() - t
    errs = {k: relerr(S.get(k), st[k]) for k in ("w", "intercept", "g_sum", "g_memory", "g_sum_intercept")}
    print(f"sparse {family:11s} K={K} {penalty:10s} batch={batch:5d} ep={ep2} {dt*1e3:8.1f} ms  " +
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
    S.close()
    return max(errs.values())

def check_dense(family, K, penalty, n=500, p=6, epochs=3, seed=2):
    rng = np.random.default_rng(seed)
    x = np.asfortranarray(rng.standard_normal((p, n)))
    bt = rng.standard_normal((K, p))
    lp = bt @ x
    if family == "gaussian": y = lp[0] + 0.1 * rng.standard_normal(n)
    elif family == "binomial": y = (rng.random(n) < 1 / (1 + np.exp(-lp[0]))).astype(float)
    elif family == "multinomial": y = np.argmax(lp + rng.gumbel(size=lp.shape), axis=0).astype(float)
    else: y = lp + 0.1 * rng.standard_normal(lp.shape)
    y = np.asfortranarray(np.reshape(y, (-1, n)))
    stream = po.Rng(seed).stream(n, n * epochs)
    gamma, alpha, beta = 0.01, 1e-3, (2e-3 if penalty != "ridge" else 0.0)
    st = po.new_state(K, p, n)
    po.saga(x, y, st, family=family, penalty=penalty, gamma=gamma, alpha=alpha, beta=beta, max_iter=epochs,
            tol=0.0, stream=stream)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    S.set_penalty(penalty, gamma, alpha, beta)
    S.upload_stream(stream)
    t = time.time()
    ep2, conv = S.run(mode="exact", max_epochs=epochs, tol=0.0)
    dt = time.time() - t
    errs = {k: relerr(S.get(k), st[k]) for k in ("w", "intercept", "g_sum", "g_memory", "g_sum_intercept")}
    print(f"dense  {family:11s} K={K} {penalty:10s} ep={ep2} {dt*1e3:8.1f} ms  " +
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
    S.close()
    return max(errs.values())

worst = 0.0
for fam, K, pen in [("binomial", 1, "elasticnet"), ("gaussian", 1, "ridge"), ("multinomial", 3, "elasticnet"),
                    ("mgaussian", 2, "grouplasso")]:
    worst = max(worst, check_sparse(fam, K, pen))
    worst = max(worst, check_sparse(fam, K, pen, batch=64))
    worst = max(worst, check_dense(fam, K, pen))
worst = max(worst, check_sparse("multinomial", 10, "elasticnet", batch=256))
print("worst rel err", worst, flush=True)

# fit() path
rng = np.random.default_rng(0)
n, p = 300, 5
X = rng.normal(size=(n, p)); y = X @ np.array([1., -2, 0.5, 0, 0]) + 0.3 + 0.01 * rng.normal(size=n)
f = sa.sgdnet(X, y, family="gaussian", nlambda=5, thresh=1e-6, seed=1)
o = po.fit(X, y, family="gaussian", nlambda=5, thresh=1e-6, seed=1)
print("fit dense: a0", relerr(f.a0, o["a0"][0]), "beta", relerr(f.beta, o["beta"][0]), "npasses", f.npasses, o["npasses"],
      "dev", relerr(f.dev_ratio, o["dev_ratio"]), flush=True)
Xs = sp.random(2000, 50, density=0.1, format="csc", random_state=1)
bt = rng.normal(size=50) * (rng.random(50) < 0.3)
ys = (rng.random(2000) < 1 / (1 + np.exp(-(Xs @ bt)))).astype(float)
for std in (False, True):
    f = sa.sgdnet(Xs, ys, family="binomial", alpha=0.5, nlambda=5, thresh=1e-5, seed=3, standardize=std)
    o = po.fit(Xs, ys, family="binomial", alpha=0.5, nlambda=5, thresh=1e-5, seed=3, standardize=std)
    print(f"fit sparse std={std}: a0", relerr(f.a0, o["a0"][0]), "beta", relerr(f.beta, o["beta"][0]), "npasses",
          f.npasses, o["npasses"], "lambda", relerr(f.lambda_, o["lambda"]), flush=True)

# timing on a C3-shaped problem (1M x 1000, 1%)
t = time.time()
pr = D.make_sparse_glm(1_000_000, 1000, 0.01, family="binomial", seed=3)
X = D.as_scipy(pr)
print("gen C3", time.time() - t, "nnz", X.nnz, flush=True)
n = 1_000_000
epochs = 4
stream = sa.RRng(3).stream(n, n * epochs)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
gamma = D.step_size(row_sq.max(), 0.5 / n, True, "binomial", n)
S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n)
S.upload_stream(stream)
for batch in (2048, 8192, 32768):
    S.enqueue_epochs(1, batch=batch); S.sync()
    t = time.time(); S.enqueue_epochs(3, batch=batch, stream_offset=n); S.sync(); dt = (time.time() - t) / 3
    pe = S.profile_epoch(batch=batch)
    ab = D.algorithmic_bytes(S.row_nnz, stream[:n], 1)
    print(f"C3 batch={batch}: {dt*1e3:.2f} ms/epoch  {ab/dt/1e9:.1f} GB/s algorithmic; gather {pe['gather_ms']:.2f} ms / "
          f"{pe['gather_launches']} launches, sweep {pe['sweep_ms']:.2f} ms", flush=True)
t = time.time(); ep, conv = S.run(mode="exact", max_epochs=1, tol=0.0); print("C3 exact epoch", time.time() - t, flush=True)
