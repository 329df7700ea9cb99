"""The fused epoch of the virtual shards (saga_vs_epoch_kernel, round 4): ONE launch per epoch whose
workgroups synchronise shard by shard, sweep their own feature slices, average the replicas slice by slice,
turn the generators' raw words into draws and produce the next epoch's sample order on spare workgroups.

It runs the same iteration as the separate launches (src/saga-sparse.h:258-348 in batches, per replica;
DESIGN.md 8 "virtual shards"): the state after every epoch must agree with them to rounding (the slab sums
are taken in another order), and with the CPU restatement (tests/test_gpu_parity.py runs through it
whenever the shapes allow: sparse x, one response, an even number of features).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATE = ("w", "intercept", "g_sum", "g_sum_intercept", "g_memory")


@pytest.fixture(scope="module")
def sa():
    import sgdnet_amd
    sgdnet_amd.load()
    return sgdnet_amd


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def _problem(n, p, family, seed):
    from sgdnet_amd import data as D
    pr = D.make_sparse_glm(n, p, 0.08, family=family, seed=seed)
    return D.as_scipy(pr), pr["y"]


def _run(sa, x, y, family, V, batch, epochs, fused, c=None, penalty="elasticnet"):
    with sa.option("fused_epoch", fused):
        S = sa.SagaSolver(x, y, family=family, n_classes=1, x_center_scaled=c)
        S.set_penalty(penalty, 0.004, 1e-4, 0.0 if penalty == "ridge" else 1e-4)
        S.set_virtual_shards(V)
        S.upload_stream(S.sharded_stream([sa.RRng(70 + v) for v in range(V)], epochs))
        draws = V * (S.n // V)
        form = S._L.sgdnet_solver_gather_form(S._h, batch)
        ep, _ = S.run(mode="batched", batch=batch, draws_per_epoch=draws, max_epochs=epochs, tol=0.0)
        assert ep == epochs
        st = {k: S.get(k) for k in STATE}
        S.set_virtual_shards(0)
        S.close()
    return st, form


@pytest.mark.parametrize("V,n,p,batch,family,penalty,centred", [
    (8, 40_000, 64, 1200, "binomial", "elasticnet", False),
    (8, 40_003, 200, 1700, "gaussian", "ridge", False),       # n not divisible by V, a tail batch
    (4, 30_000, 90, 2500, "binomial", "elasticnet", True),    # implicit centring; two XCDs per shard
    (2, 20_000, 50, 900, "gaussian", "elasticnet", True),
])
def test_fused_epoch_equals_the_separate_launches(sa, V, n, p, batch, family, penalty, centred):
    x, y = _problem(n, p, family, seed=5)
    c = np.random.default_rng(9).normal(0.05, 0.1, p) if centred else None
    one, form1 = _run(sa, x, y, family, V, batch, 3, fused=1, c=c, penalty=penalty)
    sep, form0 = _run(sa, x, y, family, V, batch, 3, fused=0, c=c, penalty=penalty)
    wt, _ = _run(sa, x, y, family, V, batch, 3, fused=2, c=c, penalty=penalty)
    assert form1 == 3 and form0 == 1                     # the fused kernel really ran / really did not
    for k in STATE:
        assert relerr(one[k], sep[k]) < 1e-11, k
        assert relerr(wt[k], one[k]) < 1e-13, k           # the two hand-off flavours sum in the same order


def test_an_odd_number_of_features_keeps_the_separate_launches(sa):
    x, y = _problem(20_000, 51, "binomial", seed=6)
    _, form = _run(sa, x, y, "binomial", 4, 800, 1, fused=1)
    assert form == 1


def test_fit_with_the_generators_inside_the_epoch_kernel(sa):
    """sgdnet() on a problem large enough for 8 virtual shards and several generators: with the fused kernel
    the sample order of the next epoch is produced by its spare workgroups and converted by the
    workgroups that read it.  Same draws (R's one stream), same path as with separate launches, and R's generator ends
    where the reference leaves it."""
    from sgdnet_amd import data as D
    n, p = 240_000, 100
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=23)
    x, y = D.as_scipy(pr).T.tocsc(), pr["y"].ravel()
    kw = dict(family="binomial", alpha=0.5, lambda_=[3e-3, 1.5e-3, 7e-4], standardize=True, thresh=1e-7, maxit=80,
              mode="batched")
    fits, rngs = {}, {}
    for fused in (1, 0):
        with sa.option("fused_epoch", fused):
            rngs[fused] = sa.RRng(5)
            fits[fused] = sa.sgdnet(x, y, rng=rngs[fused], **kw)
    a, b = fits[1], fits[0]
    assert a.npasses == b.npasses and a.draws_used == b.draws_used
    assert np.abs(a.beta - b.beta).max() <= 1e-9 * np.abs(b.beta).max()
    assert np.abs(a.a0 - b.a0).max() <= 1e-9
    host = sa.RRng(5)
    host.stream(n, a.draws_used)
    want = host.unif(32)
    assert np.array_equal(want, rngs[1].unif(32)) and np.array_equal(want, rngs[0].unif(32))
