"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the committed golden fixtures,
and -- at BASELINE.json's full sizes -- through size-independent properties.

Tolerances (fp64): exact mode executes the reference iteration in the same order, so only
libm rounding differs: 1e-10 relative (observed ~1e-15).  Batched mode sums scatter
contributions with atomics in arbitrary order: 1e-9 relative (observed ~1e-14).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STATE = ("w", "intercept", "g_sum", "g_memory", "g_sum_intercept")
TOL_EXACT, TOL_BATCHED = 1e-10, 1e-9


@pytest.fixture(scope="module")
def sa():
    import torch  # noqa: F401  -- before libsgdnet_hip.so: one HIP runtime per process (sgdnet_amd/_lib.py)
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the backend has no CPU fallback")
    return sgdnet_amd


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(b).max()))


def make_problem(family, K, n, p, dens, seed, dense=False):
    from sgdnet_amd import data as D
    if dense:
        rng = np.random.default_rng(seed)
        x = np.asfortranarray(rng.standard_normal((p, n)))
        lp = rng.standard_normal((K, p)) @ x
        if family == "gaussian":
            y = lp[0] + 0.1 * rng.standard_normal(n)
        elif family == "binomial":
            y = (rng.random(n) < 1 / (1 + np.exp(-lp[0]))).astype(float)
        elif family == "multinomial":
            y = np.argmax(lp + rng.gumbel(size=lp.shape), axis=0).astype(float)
        else:
            y = lp + 0.1 * rng.standard_normal(lp.shape)
        return x, np.asfortranarray(np.reshape(y, (-1, n)))
    pr = D.make_sparse_glm(n, p, dens, family=family, n_classes=K, seed=seed)
    return D.as_scipy(pr), pr["y"]


def run_both(sa, oracle, x, y, *, family, K, penalty, gamma, alpha, beta, epochs, mode="exact",
             batch=0, fit_intercept=True, c=None, stream=None, tol=0.0, seed=5):
    p, n = x.shape
    if stream is None:
        stream = oracle.Rng(seed).stream(n, n * epochs)
    st = oracle.new_state(K, p, n)
    ep, rc, _ = oracle.saga(x, y, st, family=family, penalty=penalty, gamma=gamma, alpha=alpha,
                            beta=beta, fit_intercept=fit_intercept, standardize=c is not None,
                            x_center_scaled=c, max_iter=epochs, tol=tol, stream=stream,
                            batch=batch if mode == "batched" else 0)
    S = sa.SagaSolver(x, y, family=family, n_classes=K, fit_intercept=fit_intercept,
                      x_center_scaled=c)
    S.set_penalty(penalty, gamma, alpha, beta)
    S.upload_stream(stream)
    ep2, conv = S.run(mode=mode, batch=batch, max_epochs=epochs, tol=tol)
    got = {k: S.get(k) for k in STATE}
    S.close()
    return (ep, rc, st), (ep2, conv, got)


CASES = [("binomial", 1, "elasticnet"), ("binomial", 1, "ridge"), ("gaussian", 1, "elasticnet"),
         ("multinomial", 3, "elasticnet"), ("multinomial", 10, "ridge"),
         ("mgaussian", 2, "grouplasso"), ("mgaussian", 3, "ridge")]


@pytest.mark.parametrize("family,K,penalty", CASES)
@pytest.mark.parametrize("fit_intercept", [True, False])
def test_sparse_exact_matches_oracle(sa, oracle, family, K, penalty, fit_intercept):
    x, y = make_problem(family, K, 2500, 120, 0.06, seed=3)
    (ep, rc, st), (ep2, conv, got) = run_both(
        sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.02, alpha=1e-3,
        beta=0.0 if penalty == "ridge" else 2e-3, epochs=3, fit_intercept=fit_intercept)
    assert ep2 == ep
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_EXACT, k


@pytest.mark.parametrize("family,K,penalty", CASES)
def test_dense_exact_matches_oracle(sa, oracle, family, K, penalty):
    x, y = make_problem(family, K, 700, 7, None, seed=4, dense=True)
    (ep, rc, st), (ep2, conv, got) = run_both(
        sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.01, alpha=1e-3,
        beta=0.0 if penalty == "ridge" else 2e-3, epochs=3)
    assert ep2 == ep
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_EXACT, k


def test_dense_exact_wide_matrix(sa, oracle):
    # p > 256: beyond the register-prefetched part of a row
    x, y = make_problem("gaussian", 1, 300, 300, None, seed=6, dense=True)
    (ep, rc, st), (ep2, conv, got) = run_both(sa, oracle, x, y, family="gaussian", K=1,
                                              penalty="elasticnet", gamma=1e-3, alpha=1e-3,
                                              beta=1e-3, epochs=2)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_EXACT, k


@pytest.mark.parametrize("family,K,penalty", CASES)
@pytest.mark.parametrize("batch", [1, 7, 64, 1000, 5000])
def test_sparse_batched_matches_batched_oracle(sa, oracle, family, K, penalty, batch):
    if batch == 1 and K > 1:
        pytest.skip("batch=1 covered for K=1")
    x, y = make_problem(family, K, 2500, 120, 0.06, seed=7)
    (ep, rc, st), (ep2, conv, got) = run_both(
        sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.004, alpha=1e-3,
        beta=0.0 if penalty == "ridge" else 2e-3, epochs=3, mode="batched", batch=batch)
    assert ep2 == ep
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k


@pytest.mark.parametrize("family,K,penalty,p", [
    ("gaussian", 1, "elasticnet", 37), ("binomial", 1, "ridge", 130), ("multinomial", 3, "elasticnet", 37),
    ("mgaussian", 2, "grouplasso", 70), ("multinomial", 10, "ridge", 8), ("binomial", 1, "elasticnet", 700)])
@pytest.mark.parametrize("batch", [7, 64, 1000])
def test_dense_batched_matches_batched_oracle(sa, oracle, family, K, penalty, p, batch):
    # dense x in batched mode (wave per draw); the oracle restates the same batched iteration on
    # the CSC form of the same matrix (every entry stored)
    n = 1500
    x, y = make_problem(family, K, n, p, None, seed=11, dense=True)
    stream = oracle.Rng(5).stream(n, n * 3)
    kw = dict(family=family, penalty=penalty, gamma=0.4 / p, alpha=1e-3, beta=0.0 if penalty == "ridge" else 2e-3)
    st = oracle.new_state(K, p, n)
    oracle.saga(sp.csc_matrix(x), y, st, max_iter=3, tol=0.0, stream=stream, batch=batch, dense_intercept=True, **kw)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    S.set_penalty(penalty, kw["gamma"], kw["alpha"], kw["beta"])
    S.upload_stream(stream)
    ep, _ = S.run(mode="batched", batch=batch, max_epochs=3, tol=0.0)
    assert ep == 3
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.close()


@pytest.mark.parametrize("family,K,penalty,p,n,batch", [
    ("gaussian", 1, "elasticnet", 10300, 700, 128), ("binomial", 1, "ridge", 11000, 500, 37),
    ("multinomial", 3, "elasticnet", 3500, 900, 200), ("mgaussian", 2, "grouplasso", 5200, 800, 800),
    ("multinomial", 10, "elasticnet", 1100, 1500, 64), ("multinomial", 16, "ridge", 700, 1500, 1000)])
def test_dense_tiled_form_matches_batched_oracle(sa, oracle, family, K, penalty, p, n, batch):
    # dense x whose K x p accumulator fits no LDS (n_classes * n_features > 10 240): gradient changes
    # of the batch to a buffer, D = X_batch^T gc by feature tiles, global sweep -- same batched
    # iteration as the oracle's, repeats within a batch included (batch close to n)
    assert K * p > 10240
    x, y = make_problem(family, K, n, p, None, seed=13, dense=True)
    stream = oracle.Rng(6).stream(n, n * 2)
    kw = dict(family=family, penalty=penalty, gamma=0.4 / p, alpha=1e-3, beta=0.0 if penalty == "ridge" else 2e-3)
    st = oracle.new_state(K, p, n)
    oracle.saga(sp.csc_matrix(x), y, st, max_iter=2, tol=0.0, stream=stream, batch=batch, dense_intercept=True, **kw)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    S.set_penalty(penalty, kw["gamma"], kw["alpha"], kw["beta"])
    S.upload_stream(stream)
    ep, _ = S.run(mode="batched", batch=batch, max_epochs=2, tol=0.0)
    assert ep == 2
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.close()


@pytest.mark.parametrize("family,K,penalty,p,n,batch", [
    ("multinomial", 17, "elasticnet", 30, 900, 64), ("mgaussian", 40, "grouplasso", 130, 800, 800),
    ("multinomial", 64, "ridge", 65, 1200, 500), ("multinomial", 33, "elasticnet", 700, 600, 90),
    ("mgaussian", 20, "elasticnet", 1, 500, 33)])
def test_dense_class_lane_form_matches_batched_oracle(sa, oracle, family, K, penalty, p, n, batch):
    # dense x with 17..64 classes (round 4): a wavefront per draw with lane k = class k, D = X_batch^T gc by feature
    # tiles sixteen classes at a time, a wavefront per feature in the sweep -- the oracle's batched iteration, repeats
    # within a batch included, with a table that would fit the LDS (K p <= 10 240) and one that would not
    x, y = make_problem(family, K, n, p, None, seed=17, dense=True)
    stream = oracle.Rng(8).stream(n, n * 2)
    kw = dict(family=family, penalty=penalty, gamma=0.3 / max(p, 4), alpha=1e-3, beta=0.0 if penalty == "ridge" else 2e-3)
    st = oracle.new_state(K, p, n)
    oracle.saga(sp.csc_matrix(x), y, st, max_iter=2, tol=0.0, stream=stream, batch=batch, dense_intercept=True, **kw)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    S.set_penalty(penalty, kw["gamma"], kw["alpha"], kw["beta"])
    S.upload_stream(stream)
    ep, _ = S.run(mode="batched", batch=batch, max_epochs=2, tol=0.0)
    assert ep == 2
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.close()


def test_wide_dense_fit_in_batched_mode_satisfies_the_kkt_conditions(sa):
    # p > 10 240 through sgdnet(): round 1 fell back to the exact iteration here (0.8 ms per draw at this
    # width).  The CPU oracle needs minutes for this shape, so the optimum is checked by its optimality
    # conditions.  Objective in the units of y (the reference fits y / sd(y) with lambda / sd(y),
    # src/sgdnet.cpp Rescale): RSS / (2n) + lambda ((1 - alpha) / (2 sd(y)) |b|^2 + alpha |b|_1)
    rng = np.random.default_rng(4)
    n, p = 1500, 10500
    X = 0.1 * rng.standard_normal((n, p))
    bt = np.zeros(p)
    bt[:20] = 10 * rng.standard_normal(20)
    y = X @ bt + 0.1 * rng.standard_normal(n) + 0.7
    lam, a = 0.3, 0.3
    fit = sa.sgdnet(X, y, seed=3, mode="batched", family="gaussian", alpha=a, lambda_=[lam], standardize=False,
                    thresh=1e-10, maxit=4000)
    assert fit.return_codes[0] == 0
    w, b = fit.beta[:, 0], fit.a0[0]
    r = X @ w + b - y
    g = X.T @ r / n + lam * (1 - a) * w / y.std()
    nz = w != 0
    assert 3 <= nz.sum() <= 200
    assert abs(r.mean()) < 1e-6
    assert np.abs(g[nz] + lam * a * np.sign(w[nz])).max() < 1e-8 * lam
    assert (np.abs(g[~nz]) <= lam * a * (1 + 1e-8)).all()


def test_dense_fit_batched_reaches_the_exact_optimum(sa, oracle):
    rng = np.random.default_rng(3)
    n, p = 4000, 12
    X = rng.standard_normal((n, p)) * rng.uniform(0.5, 2.0, p)
    y = X @ rng.standard_normal(p) + 0.3 * rng.standard_normal(n) + 1.5
    kw = dict(family="gaussian", alpha=0.6, lambda_=[0.02], thresh=1e-11, maxit=3000)
    ref = oracle.fit(X, y, seed=8, **kw)
    fit = sa.sgdnet(X, y, seed=8, mode="batched", **kw)      # automatic batch (about 2p draws)
    assert fit.return_codes[0] == 0
    assert relerr(fit.beta[:, 0], ref["beta"][0, :, 0]) < 1e-8
    assert abs(fit.a0[0] - ref["a0"][0, 0]) < 1e-8


def test_batched_handles_repeats_and_empty_rows(sa, oracle):
    # a stream that draws the same few samples over and over inside one batch, on a
    # matrix with empty samples and one long row (several 16-lane chunks)
    rng = np.random.default_rng(8)
    n, p = 400, 90
    X = sp.random(p, n, density=0.05, random_state=3, data_rvs=rng.standard_normal).tolil()
    X[:, 5] = 0.0
    X[:, 17] = 0.0
    X[:, 9] = rng.standard_normal((p, 1))
    X = X.tocsc()
    y = np.asfortranarray((rng.random(n) < 0.5).astype(float).reshape(1, n))
    stream = rng.choice([5, 9, 9, 17, 3, 3, 3, 250], size=n * 2).astype(np.uint32)
    for mode, batch, tol in (("exact", 0, TOL_EXACT), ("batched", 50, TOL_BATCHED)):
        (ep, rc, st), (ep2, conv, got) = run_both(
            sa, oracle, X, y, family="binomial", K=1, penalty="elasticnet", gamma=0.01, alpha=1e-3,
            beta=1e-3, epochs=2, mode=mode, batch=batch, stream=stream)
        for k in STATE:
            assert relerr(got[k], st[k]) < tol, (mode, k)


def test_exact_wscale_reset_branch(sa, oracle):
    # alpha*gamma = 0.3: wscale falls below SMALL every ~90 iterations, forcing the
    # mid-epoch Reset of saga-sparse.h:285-295 (and saga-dense.h:162-166)
    x, y = make_problem("gaussian", 1, 1500, 60, 0.1, seed=9)
    (_, _, st), (_, _, got) = run_both(sa, oracle, x, y, family="gaussian", K=1, penalty="elasticnet",
                                       gamma=0.3, alpha=1.0, beta=1e-3, epochs=2)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_EXACT, k
    xd, yd = make_problem("gaussian", 1, 600, 5, None, seed=9, dense=True)
    (_, _, st), (_, _, got) = run_both(sa, oracle, xd, yd, family="gaussian", K=1, penalty="ridge",
                                       gamma=0.1, alpha=3.0, beta=0.0, epochs=2)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_EXACT, k


@pytest.mark.parametrize("family,K,penalty,n,p,fit_intercept", [
    ("gaussian", 1, "elasticnet", 50, 65, True),        # one feature past a wavefront
    ("binomial", 1, "ridge", 40, 513, True),            # one feature past the 512-thread workgroup: two chunks
    ("binomial", 1, "elasticnet", 300, 1500, False),    # no intercept
    ("multinomial", 16, "elasticnet", 300, 5, True),    # K * p just beyond the register-resident kernel
    ("multinomial", 3, "ridge", 200, 1100, True),       # 512 threads, three chunks, staged in LDS
    ("mgaussian", 5, "grouplasso", 120, 70, True),      # 256-thread variant, group lasso on register columns
    ("mgaussian", 2, "grouplasso", 150, 300, True),
    ("gaussian", 1, "elasticnet", 30, 9800, True),      # 2 K p doubles exceed the LDS: state in global memory
])
def test_wide_dense_exact_kernel_edge_shapes(sa, oracle, family, K, penalty, n, p, fit_intercept):
    # saga_dense_exact_wide_kernel (workgroup per iteration) against the oracle's iteration in stream order;
    # the sample stream repeats samples back to back (register forwarding of the gradient memory)
    x, y = make_problem(family, K, n, p, None, seed=17, dense=True)
    stream = oracle.Rng(4).stream(n, 3 * n)
    stream[5:9] = stream[5]
    stream[n - 1] = stream[n]                             # across the epoch boundary
    a, b = (2e-3, 0.0) if penalty == "ridge" else (1e-3, 2e-3)
    ref, got = run_both(sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.3 / p, alpha=a, beta=b,
                        epochs=3, fit_intercept=fit_intercept, stream=stream)
    for name in STATE:
        assert relerr(got[2][name], ref[2][name]) < TOL_EXACT, name


def test_wide_dense_exact_kernel_scale_reset(sa, oracle):
    # alpha * gamma = 0.4: w_scale falls below SMALL every ~65 iterations (saga-dense.h:162-166) in the
    # workgroup kernel, whose threads each fold the scale into their own features
    x, y = make_problem("gaussian", 1, 400, 200, None, seed=19, dense=True)
    ref, got = run_both(sa, oracle, x, y, family="gaussian", K=1, penalty="ridge", gamma=0.002, alpha=200.0, beta=0.0,
                        epochs=2)
    for name in STATE:
        assert relerr(got[2][name], ref[2][name]) < TOL_EXACT, name


def test_sparse_exact_implicit_centring(sa, oracle):
    # standardize=TRUE on sparse x: the dense O(p) terms of saga-sparse.h:127-128,276-277
    x, y = make_problem("binomial", 1, 1200, 40, 0.1, seed=10)
    c = np.random.default_rng(1).normal(0, 0.05, 40)
    (_, _, st), (_, _, got) = run_both(sa, oracle, x, y, family="binomial", K=1, penalty="elasticnet",
                                       gamma=0.02, alpha=1e-3, beta=1e-3, epochs=2, c=c)
    for k in STATE:
        assert relerr(got[k], st[k]) < 1e-9, k


def test_in_kernel_convergence_and_epoch_count(sa, oracle):
    x, y = make_problem("gaussian", 1, 800, 20, 0.2, seed=12)
    for mode, batch in (("exact", 0), ("batched", 16)):
        (ep, rc, st), (ep2, conv, got) = run_both(
            sa, oracle, x, y, family="gaussian", K=1, penalty="elasticnet", gamma=0.05, alpha=0.05,
            beta=0.01, epochs=200, tol=1e-6, mode=mode, batch=batch)
        assert 1 < ep < 200 and ep2 == ep and conv and rc == 0
        assert relerr(got["w"], st["w"]) < 1e-9


def test_golden_sparse_epochs(sa):
    g = np.load(os.path.join(GOLD, "sparse_binomial_epochs.npz"))
    n = g["y"].shape[1]
    p = g["w_exact"].shape[1]
    X = sp.csc_matrix((g["val"], g["idx"], g["ptr"]), shape=(p, n))
    for tag, mode, batch, tol in (("exact", "exact", 0, TOL_EXACT), ("batch64", "batched", 64, TOL_BATCHED)):
        S = sa.SagaSolver(X, g["y"], family="binomial", n_classes=1)
        S.set_penalty("elasticnet", float(g["gamma"]), float(g["alpha"]), float(g["beta"]))
        S.upload_stream(g["stream"])
        S.run(mode=mode, batch=batch, max_epochs=3, tol=0.0)
        assert relerr(S.get("w"), g[f"w_{tag}"]) < tol
        assert relerr(S.get("intercept"), g[f"b_{tag}"]) < tol
        assert relerr(S.get("g_sum"), g[f"G_{tag}"]) < tol
        S.close()


def test_config1_iris_multinomial_path(sa):
    # BASELINE config 1: iris 150x4, multinomial, alpha = 0.8, default lambda path
    d = np.load(os.path.join(GOLD, "iris.npz"))
    gold = np.load(os.path.join(GOLD, "iris_multinomial_path.npz"))
    fit = sa.sgdnet(d["x"], d["y"], family="multinomial", alpha=0.8, seed=1)
    assert fit.npasses == float(gold["npasses"])
    assert relerr(fit.lambda_, gold["lambda_"]) < 1e-12
    beta = np.stack(fit.beta)                     # (K, p, n_lambda)
    assert relerr(beta, gold["beta"]) < 1e-8
    a0 = gold["a0"] - gold["a0"].mean(axis=0, keepdims=True)   # R/sgdnet.R:409-410
    assert relerr(fit.a0, a0) < 1e-8
    assert relerr(fit.dev_ratio, gold["dev_ratio"]) < 1e-8
    assert fit.nulldev == pytest.approx(float(gold["nulldev"]), rel=1e-12)
    assert fit.draws_used == int(fit.npasses) * 150


def test_config2_abalone_gaussian_path(sa):
    # BASELINE config 2: abalone, gaussian, full lambda path (dense kernel bring-up)
    d = np.load(os.path.join(GOLD, "abalone.npz"))
    gold = np.load(os.path.join(GOLD, "abalone_gaussian_path.npz"))
    fit = sa.sgdnet(d["x"], d["y"], family="gaussian", seed=2)
    assert fit.npasses == float(gold["npasses"])
    assert relerr(fit.lambda_, gold["lambda_"]) < 1e-12
    assert relerr(fit.beta, gold["beta"][0]) < 1e-8
    assert relerr(fit.a0, gold["a0"][0]) < 1e-8
    assert relerr(fit.dev_ratio, gold["dev_ratio"]) < 1e-8


def test_config2_abalone_libsvm_layout_path(sa):
    # BASELINE config 2 as worded there, "abalone 4177 x 8": the libsvm table the reference's benchmark vignette
    # reads -- the in-repo 9-column frame with its two sex dummies merged back into the 3-level code
    # (tests/golden/make_fixtures.py: abalone_libsvm_layout; data-raw/datasets.R:13-25)
    d = np.load(os.path.join(GOLD, "abalone.npz"))
    x8 = np.column_stack([1.0 + d["x"][:, 0] + 2.0 * d["x"][:, 1], d["x"][:, 2:]])
    assert x8.shape == (4177, 8)
    gold = np.load(os.path.join(GOLD, "abalone8_gaussian_path.npz"))
    fit = sa.sgdnet(x8, d["y"], family="gaussian", seed=2)
    assert fit.npasses == float(gold["npasses"])
    assert relerr(fit.lambda_, gold["lambda_"]) < 1e-12
    assert relerr(fit.beta, gold["beta"][0]) < 1e-8
    assert relerr(fit.a0, gold["a0"][0]) < 1e-8
    assert relerr(fit.dev_ratio, gold["dev_ratio"]) < 1e-8


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("standardize", [True, False])
def test_fit_paths_match_oracle(sa, oracle, family, sparse, standardize):
    rng = np.random.default_rng(14)
    n, p = 400, 6
    x = rng.standard_normal((n, p)) * (rng.random((n, p)) < 0.5)
    z = x @ rng.uniform(-1, 1, (p, 3)) + 0.3
    y = {"gaussian": z[:, 0] + 0.1 * rng.standard_normal(n),
         "binomial": (rng.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + rng.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.1 * rng.standard_normal((n, 2))}[family]
    xx = sp.csc_matrix(x) if sparse else x
    kw = dict(family=family, alpha=0.6, thresh=1e-5, standardize=standardize)
    # the automatic path (lambda_max formulas, null deviance) must agree exactly ...
    auto = sa.sgdnet(xx, y, seed=3, nlambda=6, maxit=1, **kw)
    ref0 = oracle.fit(xx, y, seed=3, nlambda=6, maxit=1, **kw)
    assert relerr(auto.lambda_, ref0["lambda"]) < 1e-12
    assert auto.nulldev == pytest.approx(ref0["nulldev"], rel=1e-12)
    # ... and the fits are compared below lambda_max: AT lambda_max a coefficient sits on the
    # soft-threshold boundary at rounding-noise level (|w| ~ 1e-15), where the reference's
    # relative stopping rule is decided by the last bit of libm's exp
    lam = ref0["lambda"][1:]
    fit = sa.sgdnet(xx, y, seed=3, debug=True, lambda_=lam, **kw)
    ref = oracle.fit(xx, y, seed=3, debug=True, lambda_=lam, **kw)
    assert fit.npasses == ref["npasses"]
    beta = np.stack(fit.beta) if isinstance(fit.beta, list) else fit.beta[None]
    assert relerr(beta, ref["beta"]) < 1e-8
    a0 = ref["a0"] - ref["a0"].mean(axis=0, keepdims=True) if family == "multinomial" else ref["a0"]
    assert relerr(np.atleast_2d(fit.a0), a0) < 1e-8
    assert relerr(fit.dev_ratio, ref["dev_ratio"]) < 1e-8
    assert fit.nulldev == pytest.approx(ref["nulldev"], rel=1e-12)
    for got, want in zip(fit.diagnostics["loss"], ref["losses"]):
        assert len(got) == len(want) and relerr(got, want) < 1e-9


@pytest.mark.parametrize("family", ["gaussian", "binomial", "multinomial", "mgaussian"])
@pytest.mark.parametrize("standardize", [True, False])
def test_fit_paths_on_wider_dense_rows_match_oracle(sa, oracle, family, standardize):
    # the default (exact) mode through sgdnet() on dense rows beyond the register-resident kernel: the
    # workgroup kernel under the lambda-path driver (several epochs per launch, draws consumed contiguously
    # across epochs and lambdas, warm starts) against the oracle's fit under the same set.seed()
    rng = np.random.default_rng(24)
    n, p = 300, 90
    x = rng.standard_normal((n, p)) * rng.uniform(0.5, 2.0, p) + rng.uniform(-1, 1, p)
    z = x[:, :6] @ rng.uniform(-1, 1, (6, 3)) + 0.3
    y = {"gaussian": z[:, 0] + 0.1 * rng.standard_normal(n),
         "binomial": (rng.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + rng.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.1 * rng.standard_normal((n, 2))}[family]
    kw = dict(family=family, alpha=0.6, thresh=1e-5, standardize=standardize)
    ref0 = oracle.fit(x, y, seed=3, nlambda=8, maxit=1, **kw)
    lam = ref0["lambda"][1:]                                    # below lambda_max, as in test_fit_paths_match_oracle
    fit = sa.sgdnet(x, y, seed=3, debug=True, lambda_=lam, **kw)
    ref = oracle.fit(x, y, seed=3, debug=True, lambda_=lam, **kw)
    assert fit.npasses == ref["npasses"]
    beta = np.stack(fit.beta) if isinstance(fit.beta, list) else fit.beta[None]
    assert relerr(beta, ref["beta"]) < 1e-8
    a0 = ref["a0"] - ref["a0"].mean(axis=0, keepdims=True) if family == "multinomial" else ref["a0"]
    assert relerr(np.atleast_2d(fit.a0), a0) < 1e-8
    assert relerr(fit.dev_ratio, ref["dev_ratio"]) < 1e-8
    for got, want in zip(fit.diagnostics["loss"], ref["losses"]):
        assert len(got) == len(want) and relerr(got, want) < 1e-9


def test_fit_batched_mode_reaches_the_exact_optimum(sa, oracle):
    # BASELINE config 3 shape scaled down: the relaxed mode is checked at tight convergence
    from sgdnet_amd import data as D
    n, p = 20000, 300
    pr = D.make_sparse_glm(n, p, 0.03, family="binomial", seed=15)
    X = D.as_scipy(pr).T.tocsc()               # n x p, as R would pass it
    y = pr["y"][0]
    kw = dict(family="binomial", alpha=0.5, lambda_=[20.0 / n], standardize=False, thresh=1e-11,
              maxit=2000)
    ref = oracle.fit(X, y, seed=4, **kw)
    fit = sa.sgdnet(X, y, seed=4, mode="batched", batch=512, **kw)
    assert fit.return_codes[0] == 0
    assert relerr(fit.beta[:, 0], ref["beta"][0, :, 0]) < 1e-8
    assert abs(fit.a0[0] - ref["a0"][0, 0]) < 1e-8


def test_fit_auto_mode_is_batched_where_implemented(sa, oracle):
    from sgdnet_amd import data as D
    n, p = 5000, 60
    pr = D.make_sparse_glm(n, p, 0.1, family="gaussian", seed=21)
    X = D.as_scipy(pr).T.tocsc()
    y = pr["y"][0]
    kw = dict(family="gaussian", alpha=0.7, lambda_=[0.05], standardize=False, thresh=1e-10, maxit=500)
    a = sa.sgdnet(X, y, seed=2, mode="auto", **kw)           # sparse: automatic batch
    b = sa.sgdnet(X, y, seed=2, mode="batched", **kw)
    # same path taken (scatter sums are order-dependent in the last bits, so not bitwise)
    assert relerr(a.beta[:, 0], b.beta[:, 0]) < 1e-9 and abs(a.npasses - b.npasses) <= 1
    e = sa.sgdnet(X, y, seed=2, mode="exact", **kw)
    assert relerr(a.beta[:, 0], e.beta[:, 0]) < 1e-7        # same optimum, different trajectory
    Xd = np.asarray(X.todense())
    c = sa.sgdnet(Xd, y, seed=2, mode="auto", **kw)         # dense: batched as well (K*p fits the LDS)
    d = sa.sgdnet(Xd, y, seed=2, mode="batched", **kw)
    assert relerr(c.beta[:, 0], d.beta[:, 0]) < 1e-9 and c.npasses == d.npasses
    assert relerr(c.beta[:, 0], e.beta[:, 0]) < 1e-7


def test_unsupported_and_stream_errors(sa):
    from sgdnet_amd import SgdnetError
    x, y = make_problem("gaussian", 1, 200, 5, None, seed=1, dense=True)
    S = sa.SagaSolver(x, y, family="gaussian", n_classes=1)
    S.set_penalty("ridge", 0.01, 1e-3, 0.0)
    S.upload_stream(np.zeros(100, dtype=np.uint32))
    with pytest.raises(SgdnetError) as e:
        S.run(mode="exact", max_epochs=1)          # 200 draws needed, 100 resident
    assert e.value.code == -6
    S.close()
    # more than 64 classes have no batched path (round 4: dense x with 17..64 classes has one, the class-lane form --
    # it used to be refused when its K x p accumulator exceeded a workgroup's LDS copy)
    x, y = make_problem("mgaussian", 18, 64, 600, None, seed=1, dense=True)
    S = sa.SagaSolver(x, y, family="mgaussian", n_classes=18)
    S.set_penalty("ridge", 1e-5, 1e-3, 0.0)
    S.upload_stream(np.zeros(64, dtype=np.uint32))
    ep, _ = S.run(mode="batched", batch=8, max_epochs=1, tol=0.0)
    assert ep == 1 and np.isfinite(S.get("w")).all()
    S.close()
    x, y = make_problem("mgaussian", 65, 64, 20, None, seed=1, dense=True)
    S = sa.SagaSolver(x, y, family="mgaussian", n_classes=65)
    S.set_penalty("ridge", 1e-5, 1e-3, 0.0)
    S.upload_stream(np.zeros(64, dtype=np.uint32))
    with pytest.raises(SgdnetError) as e:
        S.run(mode="batched", batch=8, max_epochs=1)
    assert e.value.code == -5
    S.close()


def _gradient_average_invariant(S, X, n_total):
    """g_sum == (1/n) sum_i x_i g_memory[:, i]^T -- the invariant every SAGA variant keeps
    (saga-sparse.h:281-282, 328-335); checked on the host from the device state."""
    M = S.get("g_memory")                      # (K, n)
    G = S.get("g_sum")                         # (K, p)
    want = (X @ M.T).T / n_total               # (K, p)
    scale = max(np.abs(want).max(), 1e-300)
    return float(np.abs(G - want).max() / scale), float(
        np.abs(S.get("g_sum_intercept") - M.sum(axis=1) / n_total).max())


@pytest.mark.parametrize("workload", ["C3", "C4"])
def test_full_size_invariants(sa, workload):
    # BASELINE configs 3 and 4 at full size: size-independent properties of the batched path
    from sgdnet_amd import data as D
    n, p, dens, seed = {"C3": (1_000_000, 1_000, 0.01, 3), "C4": (10_000_000, 10_000, 0.001, 4)}[workload]
    pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed)
    X = D.as_scipy(pr)
    epochs = 2
    stream = sa.RRng(seed).stream(n, n * epochs)
    row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
    a_l2 = b_l1 = 0.5 / n
    gamma = D.step_size(row_sq.max(), a_l2, True, "binomial", n)
    batch = min(65536, 2 * p)
    S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
    S.set_penalty("elasticnet", gamma, a_l2, b_l1)
    S.upload_stream(stream)
    dev0 = S.deviance()
    S.enqueue_epochs(epochs, batch=batch)
    S.sync()
    err_g, err_gb = _gradient_average_invariant(S, X, n)
    assert err_g < 1e-9 and err_gb < 1e-12
    # every drawn sample holds sigma(lp) - y in (-1, 1); untouched samples still hold 0
    M = S.get("g_memory")[0]
    touched = np.zeros(n, dtype=bool)
    touched[stream] = True
    assert np.all(M[~touched] == 0.0) and np.all(np.abs(M) < 1.0)
    assert np.count_nonzero(M) == np.count_nonzero(touched)
    # two epochs of SAGA reduce the deviance, and the run is reproducible to rounding
    dev1 = S.deviance()
    assert dev1 < dev0
    w1 = S.get("w")
    S2 = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
    S2.set_penalty("elasticnet", gamma, a_l2, b_l1)
    S2.upload_stream(stream)
    S2.enqueue_epochs(epochs, batch=batch)
    S2.sync()
    assert relerr(S2.get("w"), w1) < 1e-9
    S.close()
    S2.close()


@pytest.mark.parametrize("workload,V", [("C3", 4), ("C4", 8)])
def test_full_size_invariants_with_virtual_shards(sa, workload, V):
    # the bench.py configuration (compact records, 8-lane gather, virtual shards, automatic
    # window) at BASELINE's full sizes: the merged state satisfies the same invariants
    from sgdnet_amd import data as D
    n, p, dens, seed = {"C3": (1_000_000, 1_000, 0.01, 3), "C4": (10_000_000, 10_000, 0.001, 4)}[workload]
    pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed)
    X = D.as_scipy(pr)
    epochs = 2
    row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
    col_sq = np.bincount(pr["idx"], weights=pr["val"] ** 2, minlength=p)
    a_l2 = b_l1 = 0.5 / n
    gamma = D.step_size(row_sq.max(), a_l2, True, "binomial", n)
    batch = sa.auto_batch(float(row_sq.max()), float(col_sq.max()) / n)
    draws = V * (n // V)

    def run():
        S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
        S.set_penalty("elasticnet", gamma, a_l2, b_l1)
        S.set_virtual_shards(V)
        stream = S.sharded_stream([sa.RRng(seed + v) for v in range(V)], epochs)
        S.upload_stream(stream)
        dev0 = S.deviance()
        S.enqueue_epochs(epochs, batch=batch, draws_per_epoch=draws)
        S.sync()
        return S, stream, dev0

    S, stream, dev0 = run()
    err_g, err_gb = _gradient_average_invariant(S, X, n)
    assert err_g < 1e-9 and err_gb < 1e-12
    M = S.get("g_memory")[0]
    touched = np.zeros(n, dtype=bool)
    touched[stream[:epochs * draws]] = True
    assert np.all(M[~touched] == 0.0) and np.all(np.abs(M) < 1.0)
    assert np.count_nonzero(M) == np.count_nonzero(touched)
    assert S.deviance() < dev0
    w1 = S.get("w")
    S.set_virtual_shards(0)
    S.close()
    S2, _, _ = run()                                              # reproducible to rounding
    assert relerr(S2.get("w"), w1) < 1e-9
    S2.set_virtual_shards(0)
    S2.close()


@pytest.mark.parametrize("family,K,penalty", [("binomial", 1, "elasticnet"), ("multinomial", 3, "elasticnet"),
                                              ("mgaussian", 2, "grouplasso"), ("multinomial", 10, "ridge")])
def test_lds_privatised_gather_matches_batched_oracle(sa, oracle, family, K, penalty):
    # batch * mean nnz >= 48 * K * p: the LDS-privatised gather + slab sweep are selected
    n, p, batch = 30000, 40, 15000
    x, y = make_problem(family, K, n, p, 0.15, seed=16)
    (ep, rc, st), (ep2, conv, got) = run_both(
        sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.002, alpha=1e-3,
        beta=0.0 if penalty == "ridge" else 2e-3, epochs=2, mode="batched", batch=batch)
    S = sa.SagaSolver(x, y, family=family, n_classes=K)
    assert S._L.sgdnet_solver_gather_form(S._h, batch) == 1
    assert S._L.sgdnet_solver_gather_form(S._h, 64) == 0
    S.close()
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k


def test_lds_full_batches_with_a_global_atomic_tail(sa, oracle):
    # full batches take the LDS-privatised gather (256 workgroups), the short tail batch takes
    # the global-atomic gather with many more, smaller workgroups: scratch sizing must cover both
    n, p, batch = 25000, 400, 12000
    x, y = make_problem("binomial", 1, n, p, 0.015, seed=18)
    S = sa.SagaSolver(x, y, family="binomial", n_classes=1)
    assert S._L.sgdnet_solver_gather_form(S._h, batch) == 1
    assert S._L.sgdnet_solver_gather_form(S._h, n - 2 * batch) == 0
    S.close()
    (ep, rc, st), (ep2, conv, got) = run_both(
        sa, oracle, x, y, family="binomial", K=1, penalty="elasticnet", gamma=0.01, alpha=1e-3,
        beta=1e-3, epochs=3, mode="batched", batch=batch)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k


@pytest.mark.parametrize("family,K,penalty,batch,n,p", [
    ("binomial", 1, "elasticnet", 64, 2500, 120), ("binomial", 1, "elasticnet", 15000, 30000, 40),
    ("multinomial", 3, "elasticnet", 500, 2500, 120), ("mgaussian", 2, "grouplasso", 15000, 30000, 40),
    ("gaussian", 1, "ridge", 2, 1500, 50)])
def test_batched_implicit_centring_matches_batched_oracle(sa, oracle, family, K, penalty, batch, n, p):
    # sparse standardize=TRUE in batched mode: lp -= c.w and D_j -= c_j * sum(gc), O(z) per draw
    x, y = make_problem(family, K, n, p, 0.1, seed=19)
    c = np.random.default_rng(2).normal(0.05, 0.1, p)
    (ep, rc, st), (ep2, conv, got) = run_both(
        sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.003, alpha=1e-3,
        beta=0.0 if penalty == "ridge" else 2e-3, epochs=3, mode="batched", batch=batch, c=c)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k


def test_fit_batched_standardized_sparse_reaches_the_dense_optimum(sa, oracle):
    # the package default standardize=TRUE on sparse x, batched mode, against the oracle's DENSE
    # fit (explicit centring): same optimum.  (The reference's own sparse standardize path only
    # agrees with its dense path to ~1e-3, tests/testthat/test-sparse.R.)
    rng = np.random.default_rng(3)
    n, p = 4000, 60
    X = sp.random(n, p, density=0.1, format="csc", random_state=3,
                  data_rvs=lambda k: rng.normal(1.0, 1.0, k))
    bt = rng.normal(size=p) * (rng.random(p) < 0.3)
    y = (rng.random(n) < 1 / (1 + np.exp(-(X @ bt - 1)))).astype(float)
    kw = dict(family="binomial", alpha=0.5, lambda_=[0.002], thresh=1e-11, maxit=5000, standardize=True)
    dense = oracle.fit(X.toarray(), y, seed=1, **kw)
    fit = sa.sgdnet(X, y, seed=1, mode="batched", batch=64, **kw)
    assert fit.return_codes[0] == 0
    assert relerr(fit.beta[:, 0], dense["beta"][0, :, 0]) < 1e-8
    assert abs(fit.a0[0] - dense["a0"][0, 0]) < 1e-8


@pytest.mark.parametrize("seed,n,count,burn", [(1, 150, 1000, 0), (4, 1_000_000, 3_000_000, 0),
                                               (123, 4177, 624 * 3, 0), (7, 1000, 5000, 317),
                                               (42, 65536, 100_000, 1000)])
def test_device_mersenne_twister_equals_host(sa, oracle, seed, n, count, burn):
    # r_rng_device.hip reproduces floor(R::runif(0, n)) draw for draw, from any position
    # inside a 624-word block, and leaves the generator state where the host would
    S = sa.SagaSolver(sp.csc_matrix(np.ones((1, n))), np.zeros((1, n)), family="gaussian", n_classes=1)
    host, dev = sa.RRng(seed), sa.RRng(seed)
    if burn:
        host.stream(n, burn)
        dev.stream(n, burn)
    want = host.stream(n, count)
    S.generate_stream(dev, count)
    assert np.array_equal(S.get_stream(), want)
    # continuation: both generators are in the same state
    assert np.array_equal(dev.stream(n, 2000), host.stream(n, 2000))
    assert np.array_equal(want[:5], oracle.Rng(seed).stream(n, burn + 5)[burn:])
    S.close()


def test_rng_state_and_callback_paths(sa):
    # three ways to hand over R's sample order give the same fit: set.seed(seed), the generator
    # state itself (advanced in place like .Random.seed), and a unif_rand() callback
    rng = np.random.default_rng(20)
    n, p = 500, 6
    x = rng.standard_normal((n, p))
    y = x @ rng.uniform(-1, 1, p) + 0.1 * rng.standard_normal(n)
    kw = dict(family="gaussian", nlambda=4, thresh=1e-4)
    a = sa.sgdnet(x, y, seed=9, **kw)
    st = sa.RRng(9)
    b = sa.sgdnet(x, y, rng=st, **kw)
    cb_rng = sa.RRng(9)
    c = sa.sgdnet(x, y, unif=lambda: float(cb_rng.unif(1)[0]), **kw)
    for other in (b, c):
        assert other.npasses == a.npasses
        assert np.array_equal(other.beta, a.beta) and np.array_equal(other.a0, a.a0)
    # the state object was advanced by exactly the draws consumed
    ref = sa.RRng(9)
    ref.stream(n, int(a.draws_used))
    assert np.array_equal(st.stream(n, 100), ref.stream(n, 100))


def test_fit_edge_cases(sa, oracle):
    rng = np.random.default_rng(21)
    # (a) one sparse feature, a column of zeros next to it, standardize on (sd of the zero
    #     column is replaced by 1, math.h:108), tiny n
    x = np.zeros((12, 3))
    x[:, 0] = rng.standard_normal(12) * (rng.random(12) < 0.6)
    x[3, 2] = 2.0
    y = 0.5 * x[:, 0] + rng.standard_normal(12) * 0.1
    for xx in (x, sp.csc_matrix(x)):
        for mode in ("exact", "batched"):
            fit = sa.sgdnet(xx, y, family="gaussian", nlambda=3, thresh=1e-7, seed=1, mode=mode)
            ref = oracle.fit(xx, y, family="gaussian", nlambda=3, thresh=1e-7, seed=1)
            assert np.all(np.isfinite(fit.beta)) and fit.beta.shape == (3, 3)
            assert relerr(fit.lambda_, ref["lambda"]) < 1e-12
            if mode == "exact":
                assert fit.npasses == ref["npasses"] and relerr(fit.beta, ref["beta"][0]) < 1e-8
            else:                                             # same optimum within the tolerance
                assert np.abs(fit.beta - ref["beta"][0]).max() < 1e-4
    # (b) a single user lambda, maxit = 1: return code 1, one epoch, n draws
    xs = sp.random(300, 20, density=0.2, format="csc", random_state=1)
    yb = (rng.random(300) < 0.5).astype(float)
    fit = sa.sgdnet(xs, yb, family="binomial", lambda_=[0.01], maxit=1, standardize=False, seed=2)
    assert fit.npasses == 1 and fit.return_codes[0] == 1 and fit.draws_used == 300
    # (c) batch larger than n, batched mode on a problem smaller than one workgroup
    fit_b = sa.sgdnet(xs[:40], yb[:40], family="binomial", lambda_=[0.01], standardize=False, seed=2,
                      mode="batched", batch=10**6, thresh=1e-9, maxit=3000)
    ref_b = oracle.fit(xs[:40], yb[:40], family="binomial", lambda_=[0.01], standardize=False, seed=2,
                       batch=40, thresh=1e-9, maxit=3000)
    assert fit_b.npasses == ref_b["npasses"] and relerr(fit_b.beta[:, 0], ref_b["beta"][0, :, 0]) < 1e-8


# ---------------------------------------------------------------------------------------------
# Synchronous sample-sharded mode (sgdnet_solver_sync_*, sgdnet_amd/parallel.py)
# ---------------------------------------------------------------------------------------------
def _oracle_sync_rounds(oracle, x, y, K, plan, histories, lows, kw, c=None):
    """Single-process restatement: per global batch, gather over the ranks' draws, then sweep."""
    from sgdnet_amd.parallel import round_share
    p, n = x.shape
    st = oracle.new_state(K, p, n)
    Dm, d0 = np.zeros((K, p), order="F"), np.zeros(K)
    for e in range(len(histories[0])):
        for k in range(plan.rounds):
            draws = np.concatenate([
                histories[r][e][slice(*round_share(plan.sizes[r], plan.rounds, k))].astype(np.int64) + lows[r]
                for r in range(len(histories))]).astype(np.uint32)
            oracle.batch_gather(x, y, st, draws, Dm, d0, x_center_scaled=c, **kw)
            oracle.batch_sweep((p, n), st, draws.size, Dm, d0, x_center_scaled=c, **kw)
    return st


@pytest.mark.parametrize("family,K,penalty,centre", [
    ("binomial", 1, "elasticnet", False), ("multinomial", 3, "elasticnet", False),
    ("mgaussian", 2, "grouplasso", False), ("binomial", 1, "elasticnet", True)])
def test_sync_mode_single_rank_matches_oracle_rounds(sa, oracle, family, K, penalty, centre):
    import torch
    from sgdnet_amd.parallel import HipSyncShard, SyncShardedSaga
    n, p, epochs, batch = 2500, 120, 3, 333        # 333 does not divide 2500: rounds of 312/313 draws
    x, y = make_problem(family, K, n, p, 0.06, seed=23)
    c = np.random.default_rng(2).normal(0.05, 0.1, p) if centre else None
    kw = dict(family=family, penalty=penalty, gamma=0.004, alpha=1e-3, beta=2e-3, n_total=n)
    stream = oracle.Rng(9).stream(n, n * epochs)
    S = sa.SagaSolver(x, y, family=family, n_classes=K, x_center_scaled=c)
    S.set_penalty(penalty, 0.004, 1e-3, 2e-3)
    S.upload_stream(stream)
    shard = HipSyncShard(S, draws_per_epoch=n, device=torch.device("cuda", 0))
    job = SyncShardedSaga(shard, n, 1, batch)
    for _ in range(epochs):
        job.epoch(0)
    S.sync()
    got = {k: S.get(k) for k in STATE}
    st = _oracle_sync_rounds(oracle, x, y, K, job, [stream.reshape(epochs, n)], [0], kw, c)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k
    # unbinding restores the ordinary batched path on the same solver
    shard.close()
    S.set("w", np.zeros((K, p))); S.set("g_sum", np.zeros((K, p))); S.set("g_memory", np.zeros((K, n)))
    S.set("intercept", np.zeros(K)); S.set("g_sum_intercept", np.zeros(K))
    S.run(mode="batched", batch=batch, max_epochs=2, tol=0.0)
    st2 = oracle.new_state(K, p, n)
    oracle.saga(x, y, st2, family=family, penalty=penalty, gamma=0.004, alpha=1e-3, beta=2e-3,
                standardize=centre, x_center_scaled=c, max_iter=2, tol=0.0, stream=stream, batch=batch)
    for k in STATE:
        assert relerr(S.get(k), st2[k]) < TOL_BATCHED, k
    S.close()


def _gpu_sync_worker(rank, world, port, epochs, outdir, n, p, batch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch                      # before sgdnet_amd: one HIP runtime per process
    import torch.distributed as dist
    import sgdnet_amd as sa
    from sgdnet_amd import data as D
    from sgdnet_amd.parallel import HipSyncShard, SyncShardedSaga, shard_bounds
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_bounds(n, world, rank)
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=31, lo=lo, hi=hi)
    nl = hi - lo
    stream = sa.RRng(40 + rank).stream(nl, nl * epochs)
    S = sa.SagaSolver(D.as_scipy(pr), pr["y"], family="binomial", n_classes=1, n_total=n)
    S.set_penalty("elasticnet", 0.01, 1e-5, 1e-5)
    S.upload_stream(stream)
    shard = HipSyncShard(S, draws_per_epoch=nl, device=torch.device("cuda", 0), stage_on_host=True)
    job = SyncShardedSaga(shard, n, world, batch)
    for _ in range(epochs):
        job.epoch(rank)
    S.sync()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), stream=stream.reshape(epochs, nl),
             **{k: S.get(k) for k in STATE})
    S.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sync_mode_two_ranks_on_one_gpu_equal_the_single_process_iteration(sa, oracle, tmp_path):
    # two processes share cuda:0 and reduce through gloo (host staging); the replicated state must
    # equal the single-process batched iteration over the interleaved sample order
    import socket
    import torch.multiprocessing as mp
    from sgdnet_amd import data as D
    from sgdnet_amd.parallel import SyncShardedSaga, shard_bounds
    n, p, epochs, batch, world = 6001, 150, 3, 500, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gpu_sync_worker, args=(world, port, epochs, str(tmp_path), n, p, batch), nprocs=world, join=True)
    out = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=31)
    plan = SyncShardedSaga(None, n, world, batch)
    lows = [shard_bounds(n, world, r)[0] for r in range(world)]
    kw = dict(family="binomial", penalty="elasticnet", gamma=0.01, alpha=1e-5, beta=1e-5, n_total=n)
    st = _oracle_sync_rounds(oracle, D.as_scipy(pr), pr["y"], 1, plan, [o["stream"] for o in out], lows, kw)
    for k in ("w", "intercept", "g_sum", "g_sum_intercept"):
        assert np.array_equal(out[0][k], out[1][k]), k
        assert relerr(out[0][k], st[k]) < TOL_BATCHED, k
    mem = np.concatenate([o["g_memory"] for o in out], axis=1)
    assert relerr(mem, st["g_memory"]) < TOL_BATCHED


# ---------------------------------------------------------------------------------------------
# cv_sgdnet fan-out: independent fits through the same C-ABI entry point
# ---------------------------------------------------------------------------------------------
def test_cv_sgdnet_runs_the_reference_protocol(sa, oracle):
    from sgdnet_amd import data as D
    n, p = 600, 12
    pr = D.make_sparse_glm(n, p, 0.4, family="binomial", seed=5)
    X = np.asarray(D.as_scipy(pr).T.todense())
    y = pr["y"][0]
    cv = sa.cv_sgdnet(X, y, alpha=[0.5, 1.0], nfolds=3, family="binomial", type_measure="deviance",
                      seed=3, nlambda=8, thresh=1e-6)
    assert cv.name == "Binomial Deviance" and len(cv.cv_raw) == 2 and cv.cv_raw[0].shape == (3, 8)
    assert np.all(np.isfinite(cv.cv_summary)) and cv.cv_summary.shape == (16, 6)
    assert cv.alpha_min in (0.5, 1.0) and cv.lambda_1se >= cv.lambda_min
    assert sorted(np.unique(cv.foldid).tolist()) == [1, 2, 3]
    # the fold fits are the oracle's fits of the same sub-problems: fold 2 of alpha = 1, scored by hand
    train = cv.foldid == 2
    ref = oracle.fit(X[train], y[train], family="binomial", alpha=1.0, lambda_=cv.lambda_[1], thresh=1e-6, seed=9)
    lp = X[~train] @ ref["beta"][0] + ref["a0"][0]
    prob = np.clip(1 / (1 + np.exp(-lp)), 1e-5, 1 - 1e-5)
    t = y[~train][:, None]
    want = (-2 * (t * np.log(prob) + (1 - t) * np.log(1 - prob))).mean(axis=0)
    assert np.allclose(cv.cv_raw[1][1], want, rtol=2e-3)          # different sample order, same optimum
    # fan-out path (per-fit seeds, thread pool) gives the same table within the fit tolerance
    cv2 = sa.cv_sgdnet(X, y, alpha=[0.5, 1.0], nfolds=3, family="binomial", foldid=cv.foldid, devices=[0, 0],
                       nlambda=8, thresh=1e-6)
    assert np.allclose(cv2.cv_summary[:, 2], cv.cv_summary[:, 2], rtol=5e-3)


def test_cv_sgdnet_auc_draws_its_tie_breakers_from_the_shared_generator(sa):
    # R/cv_sgdnet.R with type.measure = "auc": every fold's score() calls auc() once per lambda, and auc()
    # orders equal probabilities by stats::runif(2 n) from R's global generator -- draws that also move the
    # stream the NEXT fold's fit starts from.  The protocol replayed by hand with one generator gives the table.
    rng0 = np.random.default_rng(6)
    n, p = 240, 5
    X = np.round(rng0.standard_normal((n, p)), 0)                # coarse x: many tied probabilities
    y = (rng0.random(n) < 1 / (1 + np.exp(-X[:, 0]))).astype(int)
    kw = dict(family="binomial", nlambda=5, thresh=1e-5)
    cv = sa.cv_sgdnet(X, y, alpha=1.0, nfolds=3, type_measure="auc", seed=11, **kw)
    assert cv.name == "AUC" and cv.cv_raw[0].shape == (3, 5)
    r = sa.RRng(11)
    full = sa.sgdnet(X, y, alpha=1.0, rng=r, **kw)
    from sgdnet_amd.cv import r_cut
    foldid = r_cut(r.sample(n), 3)
    assert np.array_equal(foldid, cv.foldid)
    for j in range(3):
        train = foldid == j + 1
        fit = sa.sgdnet(X[train], y[train], alpha=1.0, lambda_=full.lambda_, rng=r, family="binomial", thresh=1e-5)
        m = int((~train).sum())
        tie = r.unif(2 * m * 5).reshape(5, 2 * m).T
        want = sa.score(fit, X[~train], y[~train], "auc", tie_break=tie)
        assert np.allclose(cv.cv_raw[0][j], want, rtol=1e-12), j
    # on the device path too (sgdnet_auc_* with the same draws)
    cvd = sa.cv_sgdnet(X, y, alpha=1.0, nfolds=3, type_measure="auc", seed=11, **kw)
    assert np.allclose(cvd.cv_raw[0], cv.cv_raw[0], rtol=1e-12)


@pytest.mark.parametrize("family,K,penalty,batch", [
    ("binomial", 1, "elasticnet", 64), ("binomial", 1, "elasticnet", 6000), ("multinomial", 3, "elasticnet", 500),
    ("multinomial", 3, "elasticnet", 6000), ("multinomial", 10, "ridge", 500), ("mgaussian", 2, "grouplasso", 6000)])
def test_heavy_tailed_rows_walk_the_overflow_chains(sa, oracle, family, K, penalty, batch):
    # most samples have ~4 non-zeros (so the packed records are small), 3 % have 60-200: those
    # continue through several chained overflow records in every gather form and in exact mode
    rng = np.random.default_rng(33)
    n, p = 6000, 300
    z = np.maximum(1, rng.poisson(4, n))
    heavy = rng.random(n) < 0.03
    z[heavy] = rng.integers(60, 201, heavy.sum())
    z[:5] = [0, 1, 200, 21, 41]                                   # an empty row and the record boundaries
    ptr = np.concatenate([[0], np.cumsum(z)])
    idx = np.concatenate([np.sort(rng.choice(p, k, replace=False)) for k in z]).astype(np.int32)
    val = rng.standard_normal(ptr[-1])
    x = sp.csc_matrix((val, idx, ptr), shape=(p, n))              # column i = sample i
    lp = (x.T @ rng.standard_normal((p, K))) * 0.3
    if family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-lp[:, 0]))).astype(float).reshape(1, n)
    elif family == "multinomial":
        y = np.argmax(lp + rng.gumbel(size=lp.shape), axis=1).astype(float).reshape(1, n)
    else:
        y = np.asfortranarray((lp + 0.1 * rng.standard_normal(lp.shape)).T)
    kw = dict(family=family, K=K, penalty=penalty, gamma=0.002, alpha=1e-3,
              beta=0.0 if penalty == "ridge" else 2e-3)
    (ep, rc, st), (ep2, conv, got) = run_both(sa, oracle, x, y, epochs=2, mode="batched", batch=batch, **kw)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k
    if batch == 6000:                                             # once per family: the exact kernel too
        (ep, rc, st), (ep2, conv, got) = run_both(sa, oracle, x, y, epochs=1, mode="exact", **kw)
        for k in STATE:
            assert relerr(got[k], st[k]) < TOL_EXACT, k


def test_fit_with_heavy_tailed_rows_uses_the_device_packer(sa, oracle):
    # sgdnet_fit_sparse packs the records on the device (setup_device.hip): same data shape as
    # above, batched fit against the oracle's optimum
    rng = np.random.default_rng(34)
    n, p = 5000, 120
    z = np.maximum(1, rng.poisson(3, n))
    heavy = rng.random(n) < 0.04
    z[heavy] = rng.integers(50, 121, heavy.sum())
    ptr = np.concatenate([[0], np.cumsum(z)])
    idx = np.concatenate([np.sort(rng.choice(p, k, replace=False)) for k in z]).astype(np.int32)
    X = sp.csr_matrix((rng.standard_normal(ptr[-1]), idx, ptr), shape=(n, p)).tocsc()
    y = np.asarray(X @ (rng.standard_normal(p) * (rng.random(p) < 0.2))).ravel() + 0.1 * rng.standard_normal(n)
    kw = dict(family="gaussian", alpha=0.5, lambda_=[0.01], standardize=False, thresh=1e-10, maxit=2000)
    ref = oracle.fit(X, y, seed=3, **kw)
    for mode in ("exact", "batched"):
        fit = sa.sgdnet(X, y, seed=3, mode=mode, **kw)
        assert fit.return_codes[0] == 0
        assert relerr(fit.beta[:, 0], ref["beta"][0, :, 0]) < (1e-9 if mode == "exact" else 1e-7)


@pytest.mark.parametrize("family,K,n,p,dens,batch", [
    ("binomial", 1, 70, 1, 0.9, 16),            # one feature
    ("multinomial", 16, 900, 30, 0.2, 128),     # the largest class count of the batched kernels
    ("mgaussian", 16, 400, 5, 0.6, 64),
    ("gaussian", 1, 3, 2, 1.0, 2),              # fewer samples than lanes in a group
    ("binomial", 1, 300, 4000, 0.002, 200),     # p >> n, most columns empty
])
def test_extreme_shapes_batched_and_exact(sa, oracle, family, K, n, p, dens, batch):
    x, y = make_problem(family, K, n, p, dens, seed=41)
    penalty = "grouplasso" if family == "mgaussian" else "elasticnet"
    kw = dict(family=family, K=K, penalty=penalty, gamma=0.01, alpha=1e-3, beta=1e-3)
    (ep, rc, st), (ep2, conv, got) = run_both(sa, oracle, x, y, epochs=2, mode="batched", batch=batch, **kw)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_BATCHED, k
    (ep, rc, st), (ep2, conv, got) = run_both(sa, oracle, x, y, epochs=2, mode="exact", **kw)
    for k in STATE:
        assert relerr(got[k], st[k]) < TOL_EXACT, k


def test_more_than_sixteen_classes(sa, oracle):
    """17..64 classes: sparse x runs the binned form with a wavefront per draw (per-epoch parity with the
    batched oracle, then the fit driver against the exact optimum); dense x runs the class-lane form (round 4;
    until then: the binned form with every entry stored); K > 64 keeps the exact iteration behind mode = "batched"."""
    # kernels: K = 18 and K = 40, elastic net and group lasso, tail batch, small p (the table would fit LDS)
    for family, K, penalty in (("multinomial", 18, "elasticnet"), ("mgaussian", 40, "grouplasso")):
        x, y = make_problem(family, K, 3000, 50, 0.2, seed=12)
        ref, got = run_both(sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.02, alpha=1e-3, beta=1e-3,
                            epochs=2, mode="batched", batch=1300, seed=4)
        for name in STATE:
            assert relerr(got[2][name], ref[2][name]) < TOL_BATCHED, (K, name)
    rng = np.random.default_rng(8)
    n, p, K = 600, 6, 18
    X = sp.csc_matrix(rng.standard_normal((n, p)) * (rng.random((n, p)) < 0.7))
    y = rng.integers(0, K, n)
    y[:K] = np.arange(K)
    y[K:2 * K] = np.arange(K)                              # every class at least twice
    kw = dict(family="multinomial", alpha=0.5, lambda_=[0.01], standardize=False, maxit=3000)
    ref = oracle.fit(X, y, seed=2, thresh=1e-10, **kw)
    fit = sa.sgdnet(X, y, seed=2, mode="batched", batch=64, thresh=1e-10, **kw)      # sparse: batched kernels
    assert fit.return_codes[0] == 0
    for k in range(K):
        assert np.abs(fit.beta[k][:, 0] - ref["beta"][k, :, 0]).max() < 1e-7
    # dense x with 17..64 classes in batched mode: the class-lane form of the dense kernels (saga_dense_cl_gather_kernel;
    # rounds 2-3: the sparse binned form with every entry stored) -- same optimum, also with standardisation
    for std in (False, True):
        kw["standardize"] = std
        Xd = np.asarray(X.todense()) + (0.3 if std else 0.0)
        ref = oracle.fit(Xd, y, seed=2, thresh=1e-10, **{k_: v for k_, v in kw.items() if k_ != "thresh"})
        fit = sa.sgdnet(Xd, y, seed=2, mode="batched", batch=64, thresh=1e-10, **{k_: v for k_, v in kw.items() if k_ != "thresh"})
        assert fit.return_codes[0] == 0
        # (the batched kernels ran, not the exact iteration behind them: that one reproduces the oracle's bits)
        assert any(np.abs(fit.beta[k][:, 0] - ref["beta"][k, :, 0]).max() > 0 for k in range(K))
        for k in range(K):
            assert np.abs(fit.beta[k][:, 0] - ref["beta"][k, :, 0]).max() < 1e-7, (std, k)
        a0 = ref["a0"][:, 0] - ref["a0"][:, 0].mean()              # R/sgdnet.R:409-410 centres the class intercepts
        assert np.abs(fit.a0[:, 0] - a0).max() < 1e-7
    # more than 64 classes keep the exact iteration behind mode = "batched"
    K2 = 66
    y2 = rng.integers(0, K2, n)
    y2[:K2] = np.arange(K2)
    y2[K2:2 * K2] = np.arange(K2)
    kw2 = dict(family="multinomial", alpha=0.5, lambda_=[0.02], standardize=False, maxit=3000, thresh=1e-6)
    ref = oracle.fit(np.asarray(X.todense()), y2, seed=2, **kw2)
    fit = sa.sgdnet(np.asarray(X.todense()), y2, seed=2, mode="batched", **kw2)
    assert fit.npasses == ref["npasses"]


# ---------------------------------------------------------------------------------------------
# Periodic averaging of locally normalised shard runs (sgdnet_amd/parallel.py: ShardedSaga)
# ---------------------------------------------------------------------------------------------
def _gpu_avg_worker(rank, world, port, outdir, n, p, batch, max_epochs, tol, family="binomial", K=1, density=0.05,
                    reg=1e-5):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch                      # before sgdnet_amd: one HIP runtime per process
    import torch.distributed as dist
    import sgdnet_amd as sa
    from sgdnet_amd import data as D
    from sgdnet_amd.parallel import HipShard, ShardedSaga, merge_segments, shard_bounds
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_bounds(n, world, rank)
    pr = D.make_sparse_glm(n, p, density, family=family, n_classes=K, seed=31, lo=lo, hi=hi)
    nl = hi - lo
    S = sa.SagaSolver(D.as_scipy(pr), pr["y"], family=family, n_classes=K, n_total=nl)   # local normalisation
    S.set_penalty("elasticnet", 0.01, reg, reg)
    shard = HipShard(S, batch=batch, draws_per_epoch=nl, device=torch.device("cuda", 0), weight=nl / n,
                     stage_on_host=True)
    job = ShardedSaga(shard, world, merge_segments(nl, n, batch))
    rng = sa.RRng(50 + rank)
    S.convergence(tol)
    epochs, done = 0, False
    while not done and epochs < max_epochs:
        S.generate_stream(rng, nl)
        shard.offset = 0
        job.epoch()
        S.sync()
        flag = torch.tensor([1.0 if S.convergence(tol) else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        done = bool(flag[0] > 0.5)
        epochs += 1
    np.savez(os.path.join(outdir, f"a{rank}.npz"), w=S.get("w"), b=S.get("intercept"), epochs=epochs,
             form=S._L.sgdnet_solver_gather_form(S._h, batch))
    S.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_avg_mode_two_ranks_on_one_gpu_with_the_binned_multinomial_kernels(sa, oracle, tmp_path):
    # the config-5 code path under the multi-rank averaging: K = 10, K * p * 8 bytes beyond the LDS table
    # (binned gather/sweep, padded coefficient copy refreshed after every merge), two processes on cuda:0
    import socket
    import torch.multiprocessing as mp
    from sgdnet_amd import data as D
    n, p, K, batch, world, tol = 65536, 2000, 10, 4096, 2, 1e-10      # segments of whole batches
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gpu_avg_worker, args=(world, port, str(tmp_path), n, p, batch, 600, tol, "multinomial", K, 0.005, 1e-3),
             nprocs=world, join=True)
    out = [np.load(tmp_path / f"a{r}.npz") for r in range(world)]
    assert int(out[0]["form"]) == 2                                  # binned form
    assert np.array_equal(out[0]["w"], out[1]["w"]) and int(out[0]["epochs"]) < 600
    pr = D.make_sparse_glm(n, p, 0.005, family="multinomial", n_classes=K, seed=31)
    st = oracle.new_state(K, p, n)
    oracle.saga(D.as_scipy(pr), pr["y"], st, family="multinomial", penalty="elasticnet", gamma=0.01,
                alpha=1e-3, beta=1e-3, max_iter=3000, tol=tol, rng=oracle.Rng(3))
    assert np.abs(out[0]["w"] - st["w"]).max() < 1e-7 * np.abs(st["w"]).max()
    assert np.abs(out[0]["b"] - st["intercept"]).max() < 1e-7


@pytest.mark.timeout(900)
def test_avg_mode_two_ranks_on_one_gpu_reach_the_single_process_optimum(sa, oracle, tmp_path):
    # weakly regularised (the regime where summing globally normalised deltas once per epoch does
    # not converge): two processes share cuda:0, merge over gloo every n/32 draws per rank
    import socket
    import torch.multiprocessing as mp
    from sgdnet_amd import data as D
    n, p, batch, world, tol = 20000, 100, 128, 2, 1e-11
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gpu_avg_worker, args=(world, port, str(tmp_path), n, p, batch, 400, tol), nprocs=world, join=True)
    out = [np.load(tmp_path / f"a{r}.npz") for r in range(world)]
    assert np.array_equal(out[0]["w"], out[1]["w"]) and int(out[0]["epochs"]) < 400
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=31)
    st = oracle.new_state(1, p, n)
    ep1, _, _ = oracle.saga(D.as_scipy(pr), pr["y"], st, family="binomial", penalty="elasticnet", gamma=0.01,
                            alpha=1e-5, beta=1e-5, max_iter=2000, tol=tol, rng=oracle.Rng(3))
    assert relerr(out[0]["w"], st["w"]) < 1e-8 and abs(out[0]["b"][0] - st["intercept"][0]) < 1e-8
    # at thresh 1e-11 on this small problem the merged job needs 74 epochs against 43 of one
    # process (at the benchmark shapes and thresh 1e-6: 28-30 against 29-32)
    assert int(out[0]["epochs"]) <= 2 * ep1 + 5, (int(out[0]["epochs"]), ep1)


# ---------------------------------------------------------------------------------------------
# Virtual shards on one GPU (sgdnet_solver_set_virtual_shards): V locally normalised replicas,
# one launch per batch of all shards, averaged on the device every n / 32 draws per shard
# ---------------------------------------------------------------------------------------------
def _oracle_virtual_shards(oracle, x, y, V, batch, stream, epochs, kw):
    """Restatement with the oracle's batch halves: same shards, same streams, same merge points."""
    from sgdnet_amd.parallel import shard_bounds
    p, n = x.shape
    dps = n // V
    bounds = [shard_bounds(n, V, v) for v in range(V)]
    K = kw.pop("K", 1)
    sub = [(x[:, lo:hi], np.asfortranarray(y[:, lo:hi])) for lo, hi in bounds]
    mem = [np.zeros((K, hi - lo), order="F") for lo, hi in bounds]
    state = dict(w=np.zeros((K, p), order="F"), g_sum=np.zeros((K, p), order="F"), intercept=np.zeros(K),
                 g_sum_intercept=np.zeros(K))
    every = max(1, (n // 32) // batch)
    wts = [(hi - lo) / n for lo, hi in bounds]
    for e in range(epochs):
        reps = [dict(w=state["w"].copy(order="F"), g_sum=state["g_sum"].copy(order="F"),
                     intercept=state["intercept"].copy(), g_sum_intercept=state["g_sum_intercept"].copy(),
                     g_memory=mem[v]) for v in range(V)]
        ref = {k: state[k].copy(order="F") for k in state}
        nb = -(-dps // batch)
        for k in range(nb):
            for v in range(V):
                lo, hi = bounds[v]
                seg = stream[(e * V + v) * dps + k * batch:(e * V + v) * dps + min(dps, (k + 1) * batch)]
                local = (seg.astype(np.int64) - lo).astype(np.uint32)
                Dm, d0 = np.zeros((K, p), order="F"), np.zeros(K)
                oracle.batch_gather(sub[v][0], sub[v][1], reps[v], local, Dm, d0, n_total=hi - lo, **kw)
                oracle.batch_sweep((p, hi - lo), reps[v], local.size, Dm, d0, n_total=hi - lo, **kw)
            if k + 1 == nb or (k + 1) % every == 0:
                for key in ref:
                    ref[key] = ref[key] + sum(wts[v] * (reps[v][key] - ref[key]) for v in range(V))
                    for v in range(V):
                        reps[v][key][...] = ref[key]
        state = ref
    state["g_memory"] = np.concatenate(mem, axis=1)
    return state


@pytest.mark.parametrize("V,n,p,batch,family", [(2, 4000, 60, 64, "binomial"), (4, 6002, 90, 100, "gaussian"),
                                                (4, 40000, 64, 2500, "binomial")])
def test_virtual_shards_match_their_oracle_restatement(sa, oracle, V, n, p, batch, family):
    _virtual_shards_case(sa, oracle, V, n, p, batch, family, dense=False)


@pytest.mark.parametrize("V,n,p,batch,family", [(2, 3000, 40, 50, "binomial"), (8, 8003, 70, 90, "gaussian")])
def test_virtual_shards_on_dense_x(sa, oracle, V, n, p, batch, family):
    # the dense batched gather carries virtual shards too (16 wavefronts per workgroup, one LDS
    # copy of the accumulator); the restatement is the same as for sparse x
    _virtual_shards_case(sa, oracle, V, n, p, batch, family, dense=True)


@pytest.mark.parametrize("V,n,p,batch,family", [(2, 4000, 60, 64, "binomial"), (8, 8005, 50, 120, "gaussian")])
def test_virtual_shards_with_implicit_centring(sa, oracle, V, n, p, batch, family):
    # sparse standardize=TRUE (the package default) on shards: lp -= c.w_v per replica, recomputed
    # after every sweep and merge, and D_j -= c_j * sum(gc) of the shard in its sweep
    _virtual_shards_case(sa, oracle, V, n, p, batch, family, dense=False,
                         c=np.random.default_rng(4).normal(0.05, 0.1, p))


def _virtual_shards_case(sa, oracle, V, n, p, batch, family, dense, c=None):
    x, y = make_problem(family, 1, n, p, 0.5 if dense else 0.1, seed=29)
    kw = dict(family=family, penalty="elasticnet", gamma=0.005, alpha=1e-4, beta=1e-4)
    if dense:
        kw["dense_intercept"] = True                               # dense x: saga-dense.h's intercept step
    if c is not None:
        kw["x_center_scaled"] = c
    epochs = 3
    S = sa.SagaSolver(np.asfortranarray(x.toarray()) if dense else x, y, family=family, n_classes=1,
                      x_center_scaled=c)
    S.set_penalty("elasticnet", kw["gamma"], kw["alpha"], kw["beta"])
    S.set_virtual_shards(V)
    stream = S.sharded_stream([sa.RRng(60 + v) for v in range(V)], epochs)
    S.upload_stream(stream)
    draws = V * (n // V)
    ep, _ = S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=epochs, tol=0.0)
    assert ep == epochs
    st = _oracle_virtual_shards(oracle, x, y, V, batch, stream, epochs, kw)
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.set_virtual_shards(0)
    S.close()


@pytest.mark.parametrize("V,n,p,batch,family,K,penalty", [
    (2, 3000, 40, 50, "multinomial", 3, "elasticnet"), (4, 6002, 70, 90, "mgaussian", 2, "grouplasso"),
    (8, 8003, 33, 100, "multinomial", 10, "ridge"), (2, 3001, 25, 3001, "mgaussian", 16, "elasticnet")])
def test_virtual_shards_on_dense_x_with_several_classes(sa, oracle, V, n, p, batch, family, K, penalty):
    """Round 4: the dense batched gather carries virtual shards for 2..16 classes too (K-vector replicas, per-class
    intercept partials, first-occurrence claims per batch): the oracle's batch halves through the same shards, streams
    and merge points, saga-dense.h's intercept step."""
    x, y = make_problem(family, K, n, p, 0.5, seed=37)
    kw = dict(family=family, penalty=penalty, gamma=0.003, alpha=1e-4, beta=0.0 if penalty == "ridge" else 1e-4,
              dense_intercept=True)
    epochs = 3
    S = sa.SagaSolver(np.asfortranarray(x.toarray()), y, family=family, n_classes=K)
    S.set_penalty(penalty, kw["gamma"], kw["alpha"], kw["beta"])
    S.set_virtual_shards(V)
    stream = S.sharded_stream([sa.RRng(80 + v) for v in range(V)], epochs)
    S.upload_stream(stream)
    ep, _ = S.run(mode="batched", batch=batch, draws_per_epoch=V * (n // V), max_epochs=epochs, tol=0.0)
    assert ep == epochs
    st = _oracle_virtual_shards(oracle, x, y, V, min(batch, n // V), stream, epochs, dict(kw, K=K))
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.set_virtual_shards(0)
    S.close()


@pytest.mark.parametrize("V,batch,family", [(0, 9000, "binomial"), (0, 1500, "gaussian"), (2, 4500, "binomial"),
                                            (4, 1000, "gaussian")])
def test_compact_records_row_length_boundaries(sa, oracle, V, batch, family):
    # the K = 1 LDS gather reads two-plane compact records: 12 entries in the first plane, 12 in
    # the second (4 of them in registers, 8 on the tail path), the rest from the CSR arrays --
    # rows sit on every one of those boundaries, with and without virtual shards
    rng = np.random.default_rng(77)
    n, p = 9000, 400
    lengths = np.array([0, 1, 2, 11, 12, 13, 15, 16, 17, 23, 24, 25, 26, 40, 90])
    z = lengths[rng.integers(0, len(lengths), n)]
    z[:len(lengths)] = lengths
    ptr = np.concatenate([[0], np.cumsum(z)])
    idx = np.concatenate([np.sort(rng.choice(p, k, replace=False)) for k in z] + [np.zeros(0, int)]).astype(np.int32)
    val = rng.standard_normal(ptr[-1])
    x = sp.csc_matrix((val, idx, ptr), shape=(p, n))              # column i = sample i
    lp = np.asarray(x.T @ rng.standard_normal(p)) * 0.2
    if family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-lp))).astype(float).reshape(1, n)
    else:
        y = (lp + 0.1 * rng.standard_normal(n)).reshape(1, n)
    kw = dict(family=family, penalty="elasticnet", gamma=0.001, alpha=1e-4, beta=2e-4)
    epochs = 2
    if V == 0:
        (ep, rc, st), (ep2, conv, got) = run_both(sa, oracle, x, y, epochs=epochs, mode="batched", batch=batch, K=1,
                                                  **kw)
        for k in STATE:
            assert relerr(got[k], st[k]) < TOL_BATCHED, k
        return
    S = sa.SagaSolver(x, y, family=family, n_classes=1)
    S.set_penalty("elasticnet", kw["gamma"], kw["alpha"], kw["beta"])
    S.set_virtual_shards(V)
    stream = S.sharded_stream([sa.RRng(90 + v) for v in range(V)], epochs)
    S.upload_stream(stream)
    ep, _ = S.run(mode="batched", batch=batch, draws_per_epoch=V * (n // V), max_epochs=epochs, tol=0.0)
    assert ep == epochs
    st = _oracle_virtual_shards(oracle, x, y, V, batch, stream, epochs, kw)
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.set_virtual_shards(0)
    S.close()


def test_parallel_generators_reach_the_same_optimum(sa, oracle, monkeypatch):
    # batched fits with long epochs cut the sample stream into 8 segments with their own MT19937
    # (driver.cpp); forced on here for a small problem: same optimum as with the single R stream,
    # reproducible for a given seed, and the caller's generator ends in a defined state
    rng = np.random.default_rng(55)
    n, p = 6000, 30
    X = sp.csc_matrix(rng.standard_normal((n, p)) * (rng.random((n, p)) < 0.3))
    y = (rng.random(n) < 1 / (1 + np.exp(-np.asarray(X @ rng.standard_normal(p)).ravel()))).astype(float)
    kw = dict(family="binomial", alpha=0.5, lambda_=[0.002], standardize=False, thresh=1e-10, maxit=3000,
              mode="batched", batch=500)
    one = sa.sgdnet(X, y, seed=4, **kw)
    st_a, st_b = sa.RRng(4), sa.RRng(4)
    with sa.option("rng_generators", 8):
        a = sa.sgdnet(X, y, rng=st_a, **kw)
        b = sa.sgdnet(X, y, rng=st_b, **kw)
    assert a.return_codes[0] == 0 and abs(a.npasses - b.npasses) <= 1
    assert np.allclose(a.beta, b.beta, rtol=0, atol=1e-9)             # same streams, same fit (to summation order)
    assert np.abs(a.beta - one.beta).max() < 1e-6                     # another order, same optimum
    if a.npasses == b.npasses:                                        # then the generators ended in the same state
        assert np.array_equal(st_a.stream(n, 50), st_b.stream(n, 50))
    ref = sa.RRng(4)
    assert not np.array_equal(st_a.stream(n, 50), ref.stream(n, 50))  # the caller's generator moved


@pytest.mark.parametrize("workload,V", [("C3", 8), ("C4", 8)])
def test_bench_configuration_satisfies_the_kkt_conditions_at_convergence(sa, workload, V):
    """Full-size parity evidence for the headline mode (SURVEY.md 7 hard part 2: at sizes the CPU
    oracle cannot converge, check the optimality conditions of the reference's objective instead).
    bench.py's configuration -- 8 virtual shards, automatic window, compact records -- is run to
    convergence and the elastic-net KKT residual of  (1/n) sum_i logloss_i + a/2 |w|^2 + b |w|_1
    is evaluated on the host from x, y and the returned (w, intercept):
        g_j = x_j'(sigma - y)/n + a w_j ;  |g_j| <= b where w_j = 0,  g_j + b sign(w_j) = 0 elsewhere,
        sum_i (sigma_i - y_i) = 0 for the intercept.
    `a` is the L2 strength the iteration really applies: the reference shrinks by r = fl(1 - a gamma)
    per step (src/saga-sparse.h:234,297), and with a gamma ~ 2.5e-9 the rounding of r changes the
    effective `a` by ~4e-8 relative -- in the reference exactly as here."""
    from sgdnet_amd import data as D
    n, p, dens, seed = {"C3": (1_000_000, 1_000, 0.01, 3), "C4": (10_000_000, 10_000, 0.001, 4)}[workload]
    pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed)
    X = D.as_scipy(pr)
    Xt = X.T.tocsr()
    y = pr["y"].ravel()
    row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
    col_sq = np.bincount(pr["idx"], weights=pr["val"] ** 2, minlength=p)
    lam = 1.0 / n
    a_l2 = b_l1 = 0.5 * lam
    gamma = D.step_size(row_sq.max(), a_l2, True, "binomial", n)
    batch = sa.auto_batch(float(row_sq.max()), float(col_sq.max()) / n)
    S = sa.SagaSolver(X, pr["y"], family="binomial", n_classes=1)
    S.set_penalty("elasticnet", gamma, a_l2, b_l1)
    S.set("intercept", np.array([np.log(y.mean() / (1 - y.mean()))]))
    S.set_virtual_shards(V)
    rng = sa.RRng(seed + 1000)
    for _ in range(120):
        S.generate_stream(rng, n)
        S.enqueue_epochs(1, batch=batch, draws_per_epoch=V * (n // V))
    S.sync()
    w, b = S.get("w")[0], S.get("intercept")[0]
    S.set_virtual_shards(0)
    S.close()
    r = 1.0 / (1.0 + np.exp(-(Xt @ w + b))) - y
    a_eff = (1.0 - (1.0 - a_l2 * gamma)) / gamma

    def residual(a):
        g = (X @ r) / n + a * w
        return np.where(w == 0, np.maximum(np.abs(g) - b_l1, 0.0), np.abs(g + b_l1 * np.sign(w))).max()

    assert residual(a_eff) <= 1e-8 * lam
    assert residual(a_l2) <= 1e-6 * lam                    # nominal a: the rounding of 1 - a gamma shows
    assert abs(r.sum()) / n <= 1e-8 * lam


# ---- binned form (config 5's code path: K x p too large for any LDS copy) ----
@pytest.mark.parametrize("family,K,penalty,centre,heavy", [
    ("multinomial", 10, "elasticnet", False, False), ("mgaussian", 5, "grouplasso", False, False),
    ("multinomial", 6, "ridge", True, False), ("binomial", 1, "elasticnet", False, False),
    ("multinomial", 10, "elasticnet", False, True)])
def test_binned_form_matches_batched_oracle(sa, oracle, family, K, penalty, centre, heavy):
    """BASELINE config 5's shape (p = 100 000, 0.01 % non-zeros, alpha = 0.5) at a sample count the CPU
    oracle handles: the range-binned gather + per-range sweep against the oracle's restatement of the
    same batches, per epoch, including a tail batch, implicit centring, a group penalty, rows that
    continue in overflow records, and one feature present in a third of all samples."""
    n, p, dens, batch, epochs = 12_000, 100_000, 1e-4, 5_000, 2
    x, y = make_problem(family, K, n, p, dens, seed=33)
    if heavy:
        rng = np.random.default_rng(5)
        x = x.tolil()
        hot = rng.choice(n, n // 3, replace=False)
        x[77, hot] = rng.standard_normal(hot.size)                 # a bias-like feature
        for s_ in rng.choice(n, 40, replace=False):                 # rows of ~60 entries: overflow records
            x[rng.choice(p, 60, replace=False), s_] = rng.standard_normal(60)
        x = x.tocsc()
        x.sort_indices()
    c = None
    if centre:
        c = np.asarray(x.mean(axis=1)).ravel() * 3.0
    a, b = (1e-4, 0.0) if penalty == "ridge" else (5e-5, 5e-5)
    ref, got = run_both(sa, oracle, x, y, family=family, K=K, penalty=penalty, gamma=0.02, alpha=a, beta=b,
                        epochs=epochs, mode="batched", batch=batch, c=c, seed=9)
    for name in STATE:
        assert relerr(got[2][name], ref[2][name]) < TOL_BATCHED, name
    S = sa.SagaSolver(x, y, family=family, n_classes=K, x_center_scaled=c)
    S.set_penalty(penalty, 0.02, a, b)
    S.upload_stream(sa.RRng(1).stream(n, n))
    S.run(mode="batched", batch=batch, max_epochs=1, tol=0.0)
    assert S._L.sgdnet_solver_gather_form(S._h, batch) == 2        # the binned kernels are what ran
    S.close()


@pytest.mark.parametrize("family", ["binomial", "gaussian"])
def test_gradient_memory_moves_between_records_and_array_with_the_mode(sa, oracle, family):
    """One-response sparse fits keep the gradient memory inside the compact records while batched epochs run
    and in the K x n array for the exact kernels and the host (solver.cpp: m_to_record / m_to_array).  A
    solver that alternates batched epoch, exact epoch, host write, batched epoch must follow the oracle
    doing the same, and the host must read what the kernels wrote."""
    x, y = make_problem(family, 1, 6000, 300, 0.04, seed=12)
    p, n = x.shape
    kw = dict(family=family, penalty="elasticnet", gamma=0.004, alpha=1e-3, beta=2e-3)
    stream = oracle.Rng(8).stream(n, 4 * n)
    st = oracle.new_state(1, p, n)
    S = sa.SagaSolver(x, y, family=family, n_classes=1)
    S.set_penalty("elasticnet", kw["gamma"], kw["alpha"], kw["beta"])
    S.upload_stream(stream)
    plan = [("batched", 700), ("exact", 0), ("batched", 2500), ("batched", 2500)]
    for e, (mode, batch) in enumerate(plan):
        seg = stream[e * n:(e + 1) * n]
        oracle.saga(x, y, st, max_iter=1, tol=0.0, stream=seg, batch=batch, **kw)
        S.run(mode=mode, batch=batch, stream_offset=e * n, max_epochs=1, tol=0.0)
        for k in STATE:
            assert relerr(S.get(k), st[k]) < TOL_BATCHED, (e, k)
        if e == 2:                                   # the host rewrites the gradient memory between two batched epochs
            M = S.get("g_memory") * 0.5
            S.set("g_memory", M)
            st["g_memory"][:] = M
            G = (x @ M.T).T / n                      # keep the invariant g_sum = (1/n) sum_i x_i M_i
            S.set("g_sum", G)
            st["g_sum"][:] = np.asfortranarray(G)
            gb = M.sum(axis=1) / n
            S.set("g_sum_intercept", gb)
            st["g_sum_intercept"][:] = gb
    S.close()


@pytest.mark.parametrize("V,n,p,batch,family,K,penalty,centre", [
    (2, 4000, 60, 64, "multinomial", 3, "elasticnet", False), (4, 6002, 90, 100, "mgaussian", 2, "grouplasso", False),
    (8, 8005, 50, 120, "multinomial", 4, "ridge", True), (2, 3001, 40, 3001, "multinomial", 3, "elasticnet", True),
    (4, 6003, 70, 90, "multinomial", 10, "elasticnet", False), (2, 4001, 30, 77, "mgaussian", 6, "grouplasso", True),
    (8, 8008, 40, 100, "multinomial", 16, "elasticnet", True)])
def test_virtual_shards_with_several_classes(sa, oracle, V, n, p, batch, family, K, penalty, centre):
    """Round 3: virtual shards for 2..16 classes of sparse x (the LDS gather of the 16-lane draw, or of the class-lane
    draw from 5 classes on, against the shard's replica; a sweep that owns K-vectors; per-class intercept partials): the oracle's batch halves driven through the
    same shards, streams and merge points, including a group penalty, implicit centring, n not divisible by V and a
    window as long as a shard's epoch."""
    x, y = make_problem(family, K, n, p, 0.1, seed=31)
    c = np.asarray(x.mean(axis=1)).ravel() * 2.0 if centre else None
    kw = dict(family=family, penalty=penalty, gamma=0.004, alpha=1e-4, beta=0.0 if penalty == "ridge" else 1e-4)
    if c is not None:
        kw["x_center_scaled"] = c
    epochs = 3
    S = sa.SagaSolver(x, y, family=family, n_classes=K, x_center_scaled=c)
    S.set_penalty(penalty, kw["gamma"], kw["alpha"], kw["beta"])
    S.set_virtual_shards(V)
    stream = S.sharded_stream([sa.RRng(70 + v) for v in range(V)], epochs)
    S.upload_stream(stream)
    draws = V * (n // V)
    ep, _ = S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=epochs, tol=0.0)
    assert ep == epochs
    st = _oracle_virtual_shards(oracle, x, y, V, min(batch, n // V), stream, epochs, dict(kw, K=K))
    for k in STATE:
        assert relerr(S.get(k), st[k]) < TOL_BATCHED, k
    S.set_virtual_shards(0)
    S.close()
