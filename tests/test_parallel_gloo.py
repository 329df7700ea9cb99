"""World-size-2 test of the sample-sharded SAGA driver on CPU (gloo).

The product's merge logic (sgdnet_amd/parallel.py: snapshot -> local epoch -> one
all-reduce of the packed deltas -> apply) runs unchanged; the per-rank local solver is
an oracle-backed stand-in with the same four methods as the HIP shard, which is the only
place tests may use the oracle.  Checks: (1) the two-process result equals an in-process
emulation of the same algorithm bit for bit up to summation order, (2) the merged
iteration converges to the single-process optimum (the fixed point is preserved).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, P, DENS, SEED = 6000, 80, 0.06, 17
GAMMA, A_L2, B_L1, BATCH = 0.08, 2e-3, 1e-3, 32


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
        self.st = po.new_state(1, P, self.n)
        self.rng = po.Rng(SEED + rank)
        self.ref = None

    def _pack(self):
        s = self.st
        return np.concatenate([s["g_sum"].ravel("F"), s["w"].ravel("F"), s["g_sum_intercept"],
                               s["intercept"]])

    def snapshot(self):
        self.ref = self._pack()

    def local_epoch(self):
        stream = self.rng.stream(self.n, self.n)
        self.po.saga(self.X, self.y, self.st, family="binomial", penalty="elasticnet", gamma=GAMMA,
                     alpha=A_L2, beta=B_L1, max_iter=1, tol=0.0, stream=stream, batch=BATCH,
                     n_total=N)

    def export_delta(self):
        return self.torch.from_numpy(self._pack() - self.ref)

    def apply_merged(self, buf, w_weight):
        m = buf.numpy()
        s, KP = self.st, P
        s["g_sum"][:] = (self.ref[:KP] + m[:KP]).reshape(1, P)
        s["w"][:] = (self.ref[KP:2 * KP] + w_weight * m[KP:2 * KP]).reshape(1, P)
        s["g_sum_intercept"][:] = self.ref[2 * KP:2 * KP + 1] + m[2 * KP:2 * KP + 1]
        s["intercept"][:] = self.ref[2 * KP + 1:] + w_weight * m[2 * KP + 1:]


def _worker(rank, world, port, epochs, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sgdnet_amd.parallel import ShardedSaga
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    shard = OracleShard(rank, world)
    job = ShardedSaga(shard, world)
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


def _emulate(world, epochs):
    shards = [OracleShard(r, world) for r in range(world)]
    for _ in range(epochs):
        for s in shards:
            s.snapshot()
            s.local_epoch()
        total = sum(s.export_delta() for s in shards)
        for s in shards:
            s.apply_merged(total.clone(), 1.0 / world)
    return np.r_[shards[0].st["w"].ravel(), shards[0].st["intercept"]]


@pytest.mark.timeout(600)
def test_two_rank_gloo_matches_emulation_and_converges(tmp_path):
    import torch.multiprocessing as mp
    epochs = 150
    mp.spawn(_worker, args=(2, _free_port(), epochs, str(tmp_path)), nprocs=2, join=True)
    w0 = np.load(tmp_path / "w0.npy")
    w1 = np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1), "ranks must hold identical merged state"
    emu = _emulate(2, epochs)
    np.testing.assert_allclose(w0, emu, rtol=0, atol=1e-13)

    # the single-process optimum of the same problem (exact reference iteration)
    from oracle import pyoracle as po
    from sgdnet_amd import data as D
    pr = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED)
    st = po.new_state(1, P, N)
    po.saga(D.as_scipy(pr), pr["y"], st, family="binomial", penalty="elasticnet", gamma=GAMMA,
            alpha=A_L2, beta=B_L1, max_iter=400, tol=1e-12, rng=po.Rng(3))
    opt = np.r_[st["w"].ravel(), st["intercept"]]
    assert np.abs(w0 - opt).max() / np.abs(opt).max() < 1e-8


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
