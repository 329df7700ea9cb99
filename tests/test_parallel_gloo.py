"""World-size-2 tests of the sample-sharded SAGA drivers on CPU (gloo).

The product's merge logic (sgdnet_amd/parallel.py) runs unchanged; the per-rank local solver is
an oracle-backed stand-in with the same methods as the HIP shard, which is the only place tests
may use the oracle.

ShardedSaga (periodic averaging of locally normalised shard runs): (1) the two-process result
equals an in-process emulation of the same algorithm up to summation order, (2) on a WEAKLY
regularised problem -- the regime where summing globally normalised deltas once per epoch does
not converge -- it reaches the single-process optimum in about as many epochs as one process.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, P, DENS, SEED = 6000, 80, 0.06, 17
GAMMA, A_L2, B_L1, BATCH = 0.08, 2e-5, 1e-5, 32
PERIOD = N // 32                                       # draws per rank between merges
KW = dict(family="binomial", penalty="elasticnet", gamma=GAMMA, alpha=A_L2, beta=B_L1)


class OracleShard:
    def __init__(self, rank, world):
        import torch
        from oracle import pyoracle as po
        from sgdnet_amd import data as D
        from sgdnet_amd.parallel import shard_bounds
        self.po, self.torch = po, torch
        lo, hi = shard_bounds(N, world, rank)
        pr = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED, lo=lo, hi=hi)
        self.X, self.y, self.n = D.as_scipy(pr), pr["y"], hi - lo
        self.weight = self.n / N
        self.st = po.new_state(1, P, self.n)
        self.rng = po.Rng(SEED + rank)
        self.D, self.d0 = np.zeros((1, P), order="F"), np.zeros(1)
        self.ref = None

    def _pack(self):
        s = self.st
        return np.concatenate([s["g_sum"].ravel("F"), s["w"].ravel("F"), s["g_sum_intercept"],
                               s["intercept"]])

    def snapshot(self):
        self.ref = self._pack()

    def local_run(self, draws):
        # batched SAGA on the shard with local normalisation (n_total = n_local)
        stream = self.rng.stream(self.n, draws)
        for t0 in range(0, draws, BATCH):
            seg = stream[t0:t0 + BATCH]
            self.po.batch_gather(self.X, self.y, self.st, seg, self.D, self.d0, n_total=self.n, **KW)
            self.po.batch_sweep((P, self.n), self.st, seg.size, self.D, self.d0, n_total=self.n, **KW)

    def export_delta(self):
        return self.torch.from_numpy(self.weight * (self._pack() - self.ref))

    def apply_merged(self, buf):
        v = self.ref + buf.numpy()
        self.ref = v                          # the merged state is the next local run's snapshot
        s = self.st
        s["g_sum"][:] = v[:P].reshape(1, P)
        s["w"][:] = v[P:2 * P].reshape(1, P)
        s["g_sum_intercept"][:] = v[2 * P:2 * P + 1]
        s["intercept"][:] = v[2 * P + 1:]


def _segments(n_local):
    from sgdnet_amd.parallel import merge_segments
    return merge_segments(n_local, N, BATCH, period=PERIOD)


def _worker(rank, world, port, epochs, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sgdnet_amd.parallel import ShardedSaga
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    shard = OracleShard(rank, world)
    job = ShardedSaga(shard, world, _segments(shard.n))
    for _ in range(epochs):
        job.epoch()
    np.save(os.path.join(outdir, f"w{rank}.npy"), np.r_[shard.st["w"].ravel(), shard.st["intercept"]])
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _emulate(world, epochs, tol=None):
    shards = [OracleShard(r, world) for r in range(world)]
    segs = _segments(shards[0].n)
    assert all(_segments(s.n) == segs for s in shards)
    prev, used = None, epochs
    for e in range(epochs):
        for s in shards:
            s.snapshot()
        for seg in segs:
            for s in shards:
                s.local_run(seg)
            total = sum(s.export_delta() for s in shards)
            for s in shards:
                s.apply_merged(total.clone())
        w = shards[0].st["w"].ravel().copy()
        if tol is not None and prev is not None and np.abs(w - prev).max() <= tol * np.abs(w).max():
            used = e + 1
            break
        prev = w
    return np.r_[shards[0].st["w"].ravel(), shards[0].st["intercept"]], used


def test_merge_segments_cover_the_epoch():
    from sgdnet_amd.parallel import merge_segments
    for n_local, n_total, batch in [(1_250_000, 10_000_000, 131072), (5_000_000, 10_000_000, 131072),
                                    (3000, 6000, 32), (10, 80, 64), (7, 7, 1)]:
        segs = merge_segments(n_local, n_total, batch)
        assert sum(segs) == n_local and all(s > 0 for s in segs)
        assert len(set(segs[:-1])) <= 1 and segs[-1] <= segs[0]
        b = min(batch, n_local)
        assert segs[0] % b == 0 or len(segs) == 1
    assert merge_segments(1_250_000, 10_000_000, 131072) == [262144, 262144, 262144, 262144, 201424]


@pytest.mark.timeout(600)
def test_two_rank_gloo_matches_emulation_and_converges(tmp_path):
    import torch.multiprocessing as mp
    epochs = 12
    mp.spawn(_worker, args=(2, _free_port(), epochs, str(tmp_path)), nprocs=2, join=True)
    w0 = np.load(tmp_path / "w0.npy")
    w1 = np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1), "ranks must hold identical merged state"
    emu, _ = _emulate(2, epochs)
    np.testing.assert_allclose(w0, emu, rtol=0, atol=1e-13)

    # weakly regularised: the single-process optimum (exact reference iteration) is reached by the
    # 2- and 4-rank jobs in about as many epochs as one process needs
    from oracle import pyoracle as po
    from sgdnet_amd import data as D
    pr = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED)
    st = po.new_state(1, P, N)
    ep1, _, _ = po.saga(D.as_scipy(pr), pr["y"], st, max_iter=2000, tol=1e-11, rng=po.Rng(3), **KW)
    opt = np.r_[st["w"].ravel(), st["intercept"]]
    for world in (2, 4):
        got, used = _emulate(world, 2000, tol=1e-11)
        assert np.abs(got - opt).max() / np.abs(opt).max() < 1e-8
        assert used <= 1.5 * ep1 + 5, (world, used, ep1)


# ---------------------------------------------------------------------------------------------
# Synchronous mode: global batches split across ranks, one all-reduce per batch
# (sgdnet_amd/parallel.py: SyncShardedSaga).  Must reproduce the single-process batched
# iteration over the interleaved sample order, not merely its fixed point.
# ---------------------------------------------------------------------------------------------
S_LAM = dict(gamma=0.08, alpha=2e-5, beta=1e-5)       # weakly regularised: the regime where the
S_BATCH = 96                                           # per-epoch averaged merge oscillates


class OracleSyncShard:
    def __init__(self, rank, world, family="binomial", K=1):
        import torch
        from oracle import pyoracle as po
        from sgdnet_amd import data as D
        from sgdnet_amd.parallel import shard_bounds
        self.po, self.torch, self.family, self.K = po, torch, family, K
        self.lo, hi = shard_bounds(N, world, rank)
        pr = D.make_sparse_glm(N, P, DENS, family=family, n_classes=K, seed=SEED, lo=self.lo, hi=hi)
        self.X, self.y, self.n = D.as_scipy(pr), pr["y"], hi - self.lo
        self.st = po.new_state(K, P, self.n)
        self.rng = po.Rng(SEED + rank)
        self.buf = np.zeros(K * P + K)
        self.D = self.buf[:K * P].reshape((K, P), order="F")
        self.d0 = self.buf[K * P:]
        self.kw = dict(family=family, penalty="elasticnet", n_total=N, **S_LAM)
        self.history = []

    def sync_begin(self):
        self.stream = self.rng.stream(self.n, self.n)
        self.history.append(self.stream)

    def sync_gather(self, t0, m, rnd):
        self.po.batch_gather(self.X, self.y, self.st, self.stream[t0:t0 + m], self.D, self.d0, **self.kw)

    def sync_reduce(self, group):
        import torch.distributed as dist
        dist.all_reduce(self.torch.from_numpy(self.buf), op=dist.ReduceOp.SUM, group=group)

    def sync_sweep(self, m_global, m_local, rnd):
        self.po.batch_sweep((P, self.n), self.st, m_global, self.D, self.d0, **self.kw)

    def sync_end(self, rounds):
        pass


def _sync_worker(rank, world, port, epochs, outdir, family, K):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sgdnet_amd.parallel import SyncShardedSaga
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    shard = OracleSyncShard(rank, world, family, K)
    job = SyncShardedSaga(shard, N, world, S_BATCH)
    for _ in range(epochs):
        job.epoch(rank)
    np.save(os.path.join(outdir, f"s{rank}.npy"),
            np.r_[shard.st["w"].ravel("F"), shard.st["intercept"], shard.st["g_sum"].ravel("F")])
    np.save(os.path.join(outdir, f"h{rank}.npy"), np.stack(shard.history))
    dist.destroy_process_group()


def _sync_single_process(world, epochs, histories, family, K):
    """The same global batches on ONE process holding all samples."""
    from oracle import pyoracle as po
    from sgdnet_amd import data as D
    from sgdnet_amd.parallel import SyncShardedSaga, round_share, shard_bounds
    pr = D.make_sparse_glm(N, P, DENS, family=family, n_classes=K, seed=SEED)
    X, y = D.as_scipy(pr), pr["y"]
    st = po.new_state(K, P, N)
    plan = SyncShardedSaga(None, N, world, S_BATCH)
    Dm, d0 = np.zeros((K, P), order="F"), np.zeros(K)
    kw = dict(family=family, penalty="elasticnet", n_total=N, **S_LAM)
    for e in range(epochs):
        for k in range(plan.rounds):
            draws = []
            for r in range(world):
                lo_r = shard_bounds(N, world, r)[0]
                a, b = round_share(plan.sizes[r], plan.rounds, k)
                draws.append(histories[r][e][a:b].astype(np.int64) + lo_r)
            draws = np.concatenate(draws).astype(np.uint32)
            assert draws.size == plan.m_global[k]
            po.batch_gather(X, y, st, draws, Dm, d0, **kw)
            po.batch_sweep((P, N), st, draws.size, Dm, d0, **kw)
    return np.r_[st["w"].ravel("F"), st["intercept"], st["g_sum"].ravel("F")], (X, y)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("family,K", [("binomial", 1), ("multinomial", 3)])
def test_two_rank_sync_mode_is_the_single_process_batched_iteration(tmp_path, family, K):
    import torch.multiprocessing as mp
    epochs = 6
    mp.spawn(_sync_worker, args=(2, _free_port(), epochs, str(tmp_path), family, K), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert np.array_equal(s0, s1), "ranks must hold identical replicated state"
    hist = [np.load(tmp_path / f"h{r}.npy") for r in range(2)]
    one, _ = _sync_single_process(2, epochs, hist, family, K)
    np.testing.assert_allclose(s0, one, rtol=0, atol=1e-12)


def test_sync_round_plan_covers_every_draw_once():
    from sgdnet_amd.parallel import SyncShardedSaga, round_share
    for n_total, world, batch in [(1000, 3, 64), (17, 4, 1), (10_000_000, 8, 131072), (5, 2, 100)]:
        plan = SyncShardedSaga(None, n_total, world, batch)
        assert sum(plan.m_global) == n_total
        assert min(plan.m_global) >= world              # every rank contributes to every round
        for nr in plan.sizes:
            cuts = [round_share(nr, plan.rounds, k) for k in range(plan.rounds)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nr
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
