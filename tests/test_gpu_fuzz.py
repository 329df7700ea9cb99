"""Differential sweep: random small problems through every kernel family (exact: register-resident, workgroup,
one-wavefront, sparse; batched: LDS, atomic, binned, dense, tiled) against the CPU oracle's restatement of the
same iteration on the same sample stream.  Shapes are drawn around the dispatch boundaries (K*p = 64, p = 64 /
512, K = 16 / 17, batch = 1 / n)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
STATE = ("w", "intercept", "g_sum", "g_memory", "g_sum_intercept")


@pytest.fixture(scope="module")
def sa():
    import torch  # noqa: F401
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the backend has no CPU fallback")
    return sgdnet_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle as po
    return po


def _case(seed):
    r = np.random.default_rng(1000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    K = 1 if family in ("gaussian", "binomial") else int(r.choice([2, 3, 4, 5, 9, 16, 17, 20]))
    p = int(r.choice([1, 2, 3, 4, 7, 16, 21, 22, 63, 64, 65, 130, 511, 513, 700]))
    if K * p > 6000:
        p = max(1, 6000 // K)
    n = int(r.choice([5, 17, 64, 150, 400]))
    dense = bool(r.random() < 0.55)
    penalty = "grouplasso" if family == "mgaussian" and r.random() < 0.6 else str(r.choice(["ridge", "elasticnet"]))
    fit_intercept = bool(r.random() < 0.8)
    mode = "exact" if r.random() < 0.5 else "batched"
    batch = int(r.choice([1, 3, 16, 64, n, 4 * n])) if mode == "batched" else 0
    centre = bool((not dense) and r.random() < 0.3)              # sparse standardize = TRUE: implicit centring
    heavy_ridge = bool(r.random() < 0.15)                        # alpha * gamma = 0.3: the w_scale reset fires
    return dict(family=family, K=K, p=p, n=n, dense=dense, penalty=penalty, fit_intercept=fit_intercept, mode=mode,
                batch=batch, centre=centre, heavy_ridge=heavy_ridge, seed=seed)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SGDNET_FUZZ_KERNEL_CASES", 600))))
def test_random_problem_matches_the_oracle(sa, oracle, seed):
    c = _case(seed)
    r = np.random.default_rng(seed)
    n, p, K, family = c["n"], c["p"], c["K"], c["family"]
    dens = 1.0 if c["dense"] else float(r.choice([0.05, 0.3, 0.9]))
    X = r.standard_normal((p, n)) * (r.random((p, n)) < dens)
    if not c["dense"]:
        X[:, X.any(axis=0) == 0] = 0.0
        X[r.integers(0, p, n), np.arange(n)] += 0.5             # no empty sample
    B = r.standard_normal((K, p)) * 0.5
    lp = B @ X
    if family == "gaussian":
        y = lp[0:1] + 0.1 * r.standard_normal((1, n))
    elif family == "binomial":
        y = (r.random((1, n)) < 1 / (1 + np.exp(-lp[0:1]))).astype(float)
    elif family == "multinomial":
        y = np.argmax(lp + r.gumbel(size=lp.shape), axis=0).astype(float).reshape(1, n)
    else:
        y = lp + 0.1 * r.standard_normal(lp.shape)
    x = np.asfortranarray(X) if c["dense"] else sp.csc_matrix(X)
    y = np.asfortranarray(y)
    L = float((X ** 2).sum(axis=0).max()) + 1.0
    gamma = 0.3 / L
    a, b = (2e-3, 0.0) if c["penalty"] == "ridge" else (1e-3, 2e-3)
    if c["heavy_ridge"]:
        a = 0.3 / gamma
    cvec = r.normal(0, 0.05, p) if c["centre"] else None
    epochs = 2
    stream = oracle.Rng(seed).stream(n, n * epochs)
    if n > 8:
        stream[3:6] = stream[3]                                    # a sample drawn three times in a row
    st = oracle.new_state(K, p, n)
    ep_ref, _, _ = oracle.saga(x if not c["dense"] else sp.csc_matrix(X) if c["mode"] == "batched" else x, y, st, family=family,
                penalty=c["penalty"], gamma=gamma, alpha=a, beta=b, fit_intercept=c["fit_intercept"], max_iter=epochs,
                tol=0.0, stream=stream, batch=c["batch"] if c["mode"] == "batched" else 0,
                standardize=cvec is not None, x_center_scaled=cvec, dense_intercept=c["dense"])
    S = sa.SagaSolver(x, y, family=family, n_classes=K, fit_intercept=c["fit_intercept"], x_center_scaled=cvec)
    S.set_penalty(c["penalty"], gamma, a, b)
    S.upload_stream(stream)
    ep, _ = S.run(mode=c["mode"], batch=c["batch"], max_epochs=epochs, tol=0.0)
    assert ep == ep_ref, c                       # (a solution that is exactly zero and stays zero stops early in both)
    tol = 1e-9 if c["mode"] == "batched" else 1e-10
    for name in STATE:
        got, want = S.get(name), st[name]
        err = float(np.abs(np.asarray(got) - np.asarray(want)).max() / max(1e-300, np.abs(want).max()))
        assert err < tol or np.abs(want).max() < 1e-300, (c, name, err)
    S.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SGDNET_FUZZ_EXACT_FITS", 32))))
def test_random_fit_matches_the_oracle_fit(sa, oracle, seed):
    # the lambda-path driver in its default (exact) mode against the oracle's fit under the same set.seed():
    # random family / storage / standardisation / intercept / mixing / shape, a short path below lambda_max
    r = np.random.default_rng(5000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n = int(r.choice([40, 120, 300]))
    p = int(r.choice([2, 5, 12, 30, 70, 140]))
    sparse = bool(r.random() < 0.5)
    x = r.standard_normal((n, p)) * r.uniform(0.5, 2.0, p) * (r.random((n, p)) < (0.4 if sparse else 1.0))
    x[np.arange(n), r.integers(0, p, n)] += 0.7
    z = x[:, : min(p, 4)] @ r.uniform(-1, 1, (min(p, 4), 3)) + 0.2
    y = {"gaussian": z[:, 0] + 0.1 * r.standard_normal(n),
         "binomial": (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.1 * r.standard_normal((n, 2))}[family]
    if family in ("binomial", "multinomial"):
        y[: 3] = [0, 1, 2 if family == "multinomial" else 1]     # every class present
    xx = sp.csc_matrix(x) if sparse else x
    kw = dict(family=family, alpha=float(r.choice([0.0, 0.3, 1.0])), thresh=float(r.choice([1e-3, 1e-5])),
              standardize=bool(r.random() < 0.5), intercept=bool(r.random() < 0.8))
    if family in ("gaussian", "mgaussian") and r.random() < 0.4:
        kw["standardize_response"] = True
    if r.random() < 0.25:
        kw["maxit"] = int(r.choice([1, 3, 10]))                 # lambdas that stop at max_iter: return_code 1
    ref0 = oracle.fit(xx, y, seed=seed, nlambda=6, **{**kw, "maxit": 1})
    lam = ref0["lambda"][1:]
    if r.random() < 0.25:
        lam = lam[r.permutation(lam.size)]                      # a user-supplied sequence in any order
    fit = sa.sgdnet(xx, y, seed=seed, lambda_=lam, **kw)
    ref = oracle.fit(xx, y, seed=seed, lambda_=lam, **kw)
    assert fit.npasses == ref["npasses"], (kw, n, p, sparse)
    assert np.array_equal(np.asarray(fit.return_codes), np.asarray(ref["return_codes"])), (kw, n, p, sparse)
    beta = np.stack(fit.beta) if isinstance(fit.beta, list) else fit.beta[None]
    scale = max(np.abs(ref["beta"]).max(), 1e-12)
    assert np.abs(beta - ref["beta"]).max() < 1e-8 * scale, (kw, n, p, sparse)
    a0 = ref["a0"] - ref["a0"].mean(axis=0, keepdims=True) if family == "multinomial" else ref["a0"]
    assert np.abs(np.atleast_2d(fit.a0) - a0).max() < 1e-8 * max(np.abs(a0).max(), 1.0)
    assert np.allclose(fit.dev_ratio, ref["dev_ratio"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("registers", [0, 2, 3, 4])
def test_random_fits_with_every_exact_kernel_forced(sa, oracle, registers):
    # the same random fits with option exact_row_registers forcing each family of exact kernels wherever it is
    # legal -- 3 puts the multi-wavefront kernels onto the crowded rows (p = 2 ... 30: every draw in flight shares
    # features) and the tiny epochs the default rule keeps away from them, 0 / 2 / 4 the general, memory-state and
    # one-consumer kernels; the sparse fits are what the option changes (round 3's 700-fit sweep,
    # scripts/dev/exact_fuzz_forced.py, as a bounded test: 12 fits per value, ~10 s)
    fn = getattr(test_random_fit_matches_the_oracle_fit, "__wrapped__", test_random_fit_matches_the_oracle_fit)
    with sa.option("exact_row_registers", registers):
        for seed in range(200 + 12 * registers, 212 + 12 * registers):
            fn(sa, oracle, seed)


# seeds 88 and 91 (few strongly scaled dense features): the rule's 64-draw floor blows up; the driver restarts
# with a 16-draw window, and mode = "auto" would rerun the fit in exact mode if that failed too
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("SGDNET_FUZZ_BATCHED_FITS", 24)))) | {88, 91}))
def test_random_fit_in_batched_mode_reaches_the_oracle_optimum(sa, oracle, seed):
    # mode = "auto" (automatic window, virtual shards off at these sizes): another trajectory, the same optimum
    r = np.random.default_rng(7000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n = int(r.choice([300, 1000, 2000]))
    p = int(r.choice([5, 20, 60, 150]))
    sparse = bool(r.random() < 0.6)
    x = r.standard_normal((n, p)) * r.uniform(0.5, 2.0, p) * (r.random((n, p)) < (0.3 if sparse else 1.0))
    x[np.arange(n), r.integers(0, p, n)] += 0.7
    z = x[:, : min(p, 4)] @ r.uniform(-1, 1, (min(p, 4), 3)) + 0.2
    y = {"gaussian": z[:, 0] + 0.1 * r.standard_normal(n),
         "binomial": (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.1 * r.standard_normal((n, 2))}[family]
    if family in ("binomial", "multinomial"):
        y[: 3] = [0, 1, 2 if family == "multinomial" else 1]
    xx = sp.csc_matrix(x) if sparse else x
    kw = dict(family=family, alpha=float(r.choice([0.2, 0.7, 1.0])), standardize=bool(r.random() < 0.5))
    ref0 = oracle.fit(xx, y, seed=seed, nlambda=5, maxit=1, **kw)
    lam = ref0["lambda"][[1, 3]]
    ref = oracle.fit(xx, y, seed=seed, lambda_=lam, thresh=1e-9, maxit=3000, **kw)
    if list(ref["return_codes"]) != [0, 0]:
        pytest.skip("the exact iteration itself does not reach 1e-9 in 3000 epochs on this problem")
    # (the batched iteration is the sparse one, dense x included: its intercept step carries the reference's
    # 0.01 decay of saga-sparse.h:300-304, so intercept-dominated dense fits take up to ~2x the epochs of the
    # dense exact iteration: more room in maxit)
    fit = sa.sgdnet(xx, y, seed=seed, lambda_=lam, thresh=1e-9, mode="auto", maxit=12000, **kw)
    assert list(fit.return_codes) == [0, 0], (kw, n, p, sparse, fit.npasses, ref["npasses"])
    beta = np.stack(fit.beta) if isinstance(fit.beta, list) else fit.beta[None]
    scale = max(np.abs(ref["beta"]).max(), 1e-12)
    assert np.abs(beta - ref["beta"]).max() < 1e-5 * scale, (kw, n, p, sparse)


@pytest.mark.parametrize("seed", [1, 13, 33, 36, 37])
def test_auto_mode_agrees_with_exact_on_dense_paths(sa, seed):
    # 30-lambda paths at the default thresh on dense x (a dominant common factor in three of the five): these
    # were off by 0.006-0.6 in deviance ratio while the batched iteration carried the sparse intercept decay and a
    # 64-draw window floor (scripts/dev/auto_vs_exact_paths.py)
    r = np.random.default_rng(9000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n = int(r.choice([500, 2000, 5000]))
    p = int(r.choice([8, 40, 200, 1000]))
    sparse = bool(r.random() < 0.6)
    corr = float(r.choice([0.0, 0.0, 0.7, 0.95]))
    f = r.standard_normal((n, 1))
    x = (np.sqrt(1 - corr) * r.standard_normal((n, p)) + np.sqrt(corr) * f) * r.uniform(0.3, 3.0, p)
    assert not sparse
    x = x + r.uniform(-2, 2, p)
    k0 = min(p, 6)
    z = x[:, :k0] @ r.uniform(-1, 1, (k0, 3)) + 0.2
    y = {"gaussian": z[:, 0] + 0.3 * r.standard_normal(n),
         "binomial": (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
         "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
         "mgaussian": z[:, :2] + 0.3 * r.standard_normal((n, 2))}[family]
    kw = dict(family=family, alpha=float(r.choice([0.0, 0.5, 1.0])), standardize=bool(r.random() < 0.6), nlambda=30)
    ex = sa.sgdnet(x, y, seed=seed, **kw)
    au = sa.sgdnet(x, y, seed=seed, mode="auto", **kw)
    assert not np.any(np.asarray(au.return_codes) != 0)
    assert np.abs(np.asarray(au.dev_ratio) - np.asarray(ex.dev_ratio)).max() < 3e-3
    assert au.npasses < 2.0 * ex.npasses + 50


@pytest.mark.parametrize("seed", range(24))
def test_random_device_predict_and_score_match_the_host_mirror(sa, seed):
    # score.hip against the numpy mirror of R/predict.sgdnet.R / R/score.R: random family, storage, shape,
    # every measure of the family (AUC with per-lambda tie breakers), whole path and interpolated lambdas
    from sgdnet_amd.score import _MEASURES
    r = np.random.default_rng(11000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    n, p = int(r.choice([60, 300, 1500])), int(r.choice([1, 3, 17, 70, 300]))
    K = int(r.choice([3, 5, 12])) if family == "multinomial" else (int(r.choice([2, 4])) if family == "mgaussian" else 1)
    sparse = bool(r.random() < 0.5)
    x = np.round(r.standard_normal((n, p)), 1) * (r.random((n, p)) < (0.3 if sparse else 1.0))
    x[np.arange(n), r.integers(0, p, n)] += 0.5
    z = x[:, : min(p, 4)] @ r.uniform(-1, 1, (min(p, 4), max(K, 2))) + 0.1
    if family == "gaussian":
        y = z[:, 0] + 0.2 * r.standard_normal(n)
    elif family == "binomial":
        y = (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(int)
        y[:2] = [0, 1]
    elif family == "multinomial":
        y = np.argmax(z[:, :K] + r.gumbel(size=(n, K)), axis=1) if z.shape[1] >= K else r.integers(0, K, n)
        y[:K] = np.arange(K)
    else:
        y = z[:, :K] + 0.2 * r.standard_normal((n, K)) if z.shape[1] >= K else r.standard_normal((n, K))
    xin = sp.csc_matrix(x) if sparse else x
    ntr = n // 2
    fit = sa.sgdnet(xin[:ntr], y[:ntr], family=family, alpha=0.6, nlambda=9, thresh=1e-4, mode="auto")
    xt, yt = xin[ntr:], y[ntr:]
    if family in ("binomial", "multinomial") and len(np.unique(yt)) < len(np.unique(y[:ntr])):
        pytest.skip("a class is missing from the held-out half")
    for typ in ("link", "response"):
        a, b = sa.predict(fit, xt, type=typ), sa.predict(fit, xt, type=typ, device=0)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-10, atol=1e-12), typ
    L = len(fit.lambda_)
    for sel in (None, fit.lambda_[[0, L // 2]] * 0.93):
        Ls = L if sel is None else 2
        for measure in _MEASURES[family]:
            tb = r.random((2 * (n - ntr), Ls)) if measure == "auc" else None
            host = sa.score(fit, xt, yt, measure, s=sel, tie_break=tb)
            dev = sa.score(fit, xt, yt, measure, s=sel, device=0, tie_break=tb)
            assert host.shape == dev.shape and np.allclose(host, dev, rtol=1e-9, atol=1e-12), (measure, host, dev)


@pytest.mark.parametrize("seed", range(24))
def test_random_wide_dense_rows_in_exact_mode(sa, oracle, seed):
    # the workgroup exact kernel on rows of thousands of features: LDS-staged state (padded to whole chunks),
    # state in global memory (2 K p doubles beyond the LDS), 1 / 2..4 / 5..16 class variants, several batches of
    # chunks per thread, few samples (repeats back to back)
    r = np.random.default_rng(23000 + seed)
    family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
    K = 1 if family in ("gaussian", "binomial") else int(r.choice([2, 4, 5, 16]))
    p = int(r.choice([1500, 4100, 9000, 12500])) // (1 if K == 1 else (2 if K <= 4 else K))
    n = int(r.choice([12, 40, 90]))
    X = r.standard_normal((p, n))
    B = r.standard_normal((K, p)) * (r.random((K, p)) < 0.02)
    lp = B @ X
    if family == "gaussian":
        y = lp[0:1] + 0.1 * r.standard_normal((1, n))
    elif family == "binomial":
        y = (r.random((1, n)) < 1 / (1 + np.exp(-lp[0:1]))).astype(float)
    elif family == "multinomial":
        y = np.argmax(lp + r.gumbel(size=lp.shape), axis=0).astype(float).reshape(1, n)
    else:
        y = lp + 0.1 * r.standard_normal(lp.shape)
    x, y = np.asfortranarray(X), np.asfortranarray(y)
    penalty = "grouplasso" if family == "mgaussian" and r.random() < 0.6 else str(r.choice(["ridge", "elasticnet"]))
    gamma = 0.3 / (float((X ** 2).sum(axis=0).max()) + 1.0)
    a, b = (2e-3, 0.0) if penalty == "ridge" else (1e-3, 2e-3)
    fit_intercept = bool(r.random() < 0.8)
    stream = oracle.Rng(seed).stream(n, 3 * n)
    stream[2:5] = stream[2]
    st = oracle.new_state(K, p, n)
    ep_ref, _, _ = oracle.saga(x, y, st, family=family, penalty=penalty, gamma=gamma, alpha=a, beta=b,
                               fit_intercept=fit_intercept, max_iter=3, tol=0.0, stream=stream)
    S = sa.SagaSolver(x, y, family=family, n_classes=K, fit_intercept=fit_intercept)
    S.set_penalty(penalty, gamma, a, b)
    S.upload_stream(stream)
    ep, _ = S.run(mode="exact", max_epochs=3, tol=0.0)
    assert ep == ep_ref
    for name in STATE:
        got, want = S.get(name), st[name]
        err = float(np.abs(np.asarray(got) - np.asarray(want)).max() / max(1e-300, np.abs(want).max()))
        assert err < 1e-10 or np.abs(want).max() < 1e-300, (family, K, p, n, penalty, name, err)
    S.close()


def test_concurrent_fits_from_threads_equal_the_sequential_ones(sa):
    # cv_sgdnet fans fits out over threads (one per entry of `devices`, several may share a GPU): the library keeps
    # per-call state only (per-device kernel attributes, a mutex around the jump polynomials, thread-local error
    # text).  Exact-mode fits are deterministic, so 24 fits run by 6 threads must equal the same fits run in turn.
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for seed in range(24):
        r = np.random.default_rng(31000 + seed)
        family = ["gaussian", "binomial", "multinomial", "mgaussian"][seed % 4]
        n, p = int(r.choice([80, 300])), int(r.choice([4, 30, 120]))
        sparse = bool(r.random() < 0.5)
        x = r.standard_normal((n, p)) * (r.random((n, p)) < (0.4 if sparse else 1.0))
        x[np.arange(n), r.integers(0, p, n)] += 0.7
        z = x[:, : min(p, 4)] @ r.uniform(-1, 1, (min(p, 4), 3)) + 0.2
        y = {"gaussian": z[:, 0] + 0.1 * r.standard_normal(n),
             "binomial": (r.random(n) < 1 / (1 + np.exp(-z[:, 0]))).astype(float),
             "multinomial": np.argmax(z + r.gumbel(size=z.shape), axis=1).astype(float),
             "mgaussian": z[:, :2] + 0.1 * r.standard_normal((n, 2))}[family]
        if family in ("binomial", "multinomial"):
            y[:3] = [0, 1, 2 if family == "multinomial" else 1]
        jobs.append((sp.csc_matrix(x) if sparse else x, y, dict(family=family, alpha=0.5, nlambda=6, seed=seed)))

    def run(job):
        x, y, kw = job
        f = sa.sgdnet(x, y, **kw)
        return np.concatenate([np.ravel(b) for b in (f.beta if isinstance(f.beta, list) else [f.beta])]), np.ravel(f.a0), f.npasses

    want = [run(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=6) as pool:
        got = list(pool.map(run, jobs))
    for (b0, a0, n0), (b1, a1, n1) in zip(want, got):
        assert n0 == n1 and np.array_equal(b0, b1) and np.array_equal(a0, a1)
