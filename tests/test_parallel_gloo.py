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
