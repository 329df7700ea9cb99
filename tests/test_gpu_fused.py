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


@pytest.mark.parametrize("ranks,V,centred", [(2, 4, False), (2, 4, True)])
def test_linked_solvers_average_their_replicas_inside_the_epoch_kernel(sa, ranks, V, centred):
    """sgdnet_solver_link_peers (the multi-GPU merge of SURVEY.md 8e, round 4): `ranks` sample-sharded solvers with V
    virtual shards each run as ONE ranks * V-way average -- every rank's epoch kernel adds to the slice counters of
    all ranks and reads their exchange buffers directly.  On this one-GPU box the ranks share the device (a CU
    budget each, ordinary pointers instead of peer pointers); the arithmetic is that of one solver holding all
    samples with ranks * V shards, in the same order: equal to rounding.
    (Two ranks only: the ranks' kernels wait for each other, so they must RUN side by side; four solver streams
    on one device can end up behind each other in one of the process's four hardware queues.)"""
    n, p, batch, epochs, family = 48_000, 64, 700, 3, "binomial"
    G = ranks * V
    x, y = _problem(n, p, family, seed=8)                  # x: (p, n), sample i = column i
    c = np.random.default_rng(3).normal(0.05, 0.1, p) if centred else None
    period = 2 * batch                                     # draws per shard between merges, the same for every solver
    one = sa.SagaSolver(x, y, family=family, n_classes=1, x_center_scaled=c)
    one.set_penalty("elasticnet", 0.004, 1e-4, 1e-4)
    one.set_virtual_shards(G)
    one.set_merge_period(period)
    one.upload_stream(one.sharded_stream([sa.RRng(90 + v) for v in range(G)], epochs))
    one.run(mode="batched", batch=batch, draws_per_epoch=G * (n // G), max_epochs=epochs, tol=0.0)
    want = {k: one.get(k) for k in STATE}
    one.close()

    nl = n // ranks
    solvers = []
    for q in range(ranks):
        S = sa.SagaSolver(x[:, q * nl:(q + 1) * nl].tocsc(), y[..., q * nl:(q + 1) * nl], family=family, n_classes=1,
                          x_center_scaled=c)
        S.set_penalty("elasticnet", 0.004, 1e-4, 1e-4)
        S.set_cu_budget(256 // ranks)
        S.set_virtual_shards(V)
        S.set_merge_period(period)
        S.upload_stream(S.sharded_stream([sa.RRng(90 + q * V + u) for u in range(V)], epochs))
        solvers.append(S)
    sa.link_peers(solvers)
    for S in solvers:                                      # every rank's epochs are enqueued before any is waited for
        assert S._L.sgdnet_solver_gather_form(S._h, batch) == 3
        S.enqueue_epochs(epochs, batch=batch, stream_offset=0, draws_per_epoch=V * (nl // V))
    for S in solvers:
        S.sync()
    for k in ("w", "intercept", "g_sum", "g_sum_intercept"):
        for S in solvers:
            assert relerr(S.get(k), want[k]) < 1e-13, k
    gm = np.concatenate([S.get("g_memory") for S in solvers], axis=1)
    assert relerr(gm, want["g_memory"]) < 1e-13
    sa.link_peers(solvers[:1])
    for S in solvers:
        S.close()


def test_fit_sharded_over_two_ranks_equals_the_fit_on_one_gpu(sa):
    """control.n_gpus (ABI 4): sgdnet_fit_sparse cuts the samples into one range per GPU, gives every rank its own
    solver, virtual shards and generators on R's ONE stream (rank q's start lo_q draws in, all of them moving n draws
    per epoch), links the solvers and runs the path -- here with both ranks on this box's one GPU.  The job is the
    same 8-way average over the same draws as the single-GPU fit: same epochs, same coefficients, and R's generator
    ends on the same draw."""
    from sgdnet_amd import data as D
    n, p = 240_000, 100
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=31)
    x, y = D.as_scipy(pr).T.tocsc(), pr["y"].ravel()
    kw = dict(family="binomial", alpha=0.5, lambda_=[3e-3, 1.5e-3, 7e-4], standardize=True, thresh=1e-7, maxit=80,
              mode="batched")
    r1, r2 = sa.RRng(12), sa.RRng(12)
    with sa.option("host_setup", 1):                       # the sharded fit prepares x on the host (it cuts that copy into
        one = sa.sgdnet(x, y, rng=r1, **kw)                # the ranks' ranges): the same window rule as the reference fit
    two = sa.sgdnet(x, y, rng=r2, devices=[0, 0], **kw)
    assert two.npasses == one.npasses and two.draws_used == one.draws_used
    assert np.abs(two.beta - one.beta).max() <= 1e-11 * np.abs(one.beta).max()
    assert np.abs(two.a0 - one.a0).max() <= 1e-11
    assert np.abs(np.asarray(two.dev_ratio) - np.asarray(one.dev_ratio)).max() < 1e-12
    assert np.array_equal(r1.unif(32), r2.unif(32))
    with pytest.raises(Exception, match="n_gpus"):          # what the sharded fit does not cover says so
        sa.sgdnet(x, y, rng=sa.RRng(1), devices=[0, 0], **dict(kw, mode="exact"))


def test_two_processes_on_one_gpu_fall_back_to_separate_launches(sa, tmp_path):
    """The epoch kernel's workgroups wait for each other, so a launch needs (nearly) the whole GPU at once; two
    processes that fit at the same time on ONE GPU (R workers, say) cannot both have it.  The start barrier notices
    (nothing has been modified yet), the solver runs that epoch -- and the rest of its life -- as separate launches, and
    the fit comes out as if nothing had happened: both processes return the path a lone process computes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(f'''
import sys, numpy as np
sys.path.insert(0, {root!r})
import sgdnet_amd as sa
from sgdnet_amd import data as D
n, p = 400_000, 200
pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=41)
x, y = D.as_scipy(pr).T.tocsc(), pr["y"].ravel()
kw = dict(family="binomial", alpha=0.5, nlambda=12, standardize=False, thresh=1e-6, maxit=60, mode="batched")
if sys.argv[1] == "ref":
    sa.set_option("fused_epoch", 0)
else:                                  # both workers have their data: start fitting together
    import os, time
    open(sys.argv[2] + ".ready", "w").close()
    t0 = time.time()
    while not os.path.exists(sys.argv[3] + ".ready") and time.time() - t0 < 120:
        time.sleep(0.001)
fits = [sa.sgdnet(x, y, seed=3, **kw) for _ in range(1 if sys.argv[1] == "ref" else 12)]
np.save(sys.argv[2], np.stack([f.beta for f in fits]))
print("passes", [f.npasses for f in fits])
''')
    env = dict(os.environ, SGDNET_TRACE="1")
    ref = tmp_path / "ref.npy"
    subprocess.run([sys.executable, str(script), "ref", str(ref)], check=True, env=env, capture_output=True, timeout=600)
    outs = [tmp_path / f"w{i}.npy" for i in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script), "work", str(outs[i]), str(outs[1 - i])], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(2)]
    logs = [pr.communicate(timeout=600) for pr in procs]
    for pr, (so, se) in zip(procs, logs):
        assert pr.returncode == 0, se[-2000:]
    fell_back = sum("could not become resident" in se for _, se in logs)
    print(f"workers that fell back to separate launches: {fell_back} of 2")
    want = np.load(ref)[0]
    for o in outs:
        got = np.load(o)
        for k in range(got.shape[0]):
            assert np.abs(got[k] - want).max() <= 1e-8 * np.abs(want).max()
