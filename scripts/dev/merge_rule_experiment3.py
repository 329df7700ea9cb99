"""CPU emulation (oracle batch halves): per-epoch-style exchange with LOCAL normalisation and a
merge period.  Every rank runs SAGA on its shard as if the shard were the data set (g_sum
increments weighted 1/n_local) starting from the merged (w, g_sum); every `period` draws per
rank the deltas are averaged with shard-size weights.  Question: does a bounded local run
length make the scheme converge for every world size?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from sgdnet_amd import data as D
from sgdnet_amd.parallel import shard_bounds

N, P, DENS, SEED = 400_000, 2000, 0.005, 17
pr = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
LAM = float(os.environ.get("LAM_MULT", "1")) / N
A_L2 = B_L1 = 0.5 * LAM
GAMMA = D.step_size(row_sq.max(), A_L2, True, "binomial", N)
BATCH, TOL = 4000, 1e-6
KW = dict(family="binomial", penalty="elasticnet", gamma=GAMMA, alpha=A_L2, beta=B_L1)


class Shard:
    def __init__(self, rank, world):
        lo, hi = shard_bounds(N, world, rank)
        p2 = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED, lo=lo, hi=hi)
        self.X, self.y, self.n = D.as_scipy(p2), p2["y"], hi - lo
        self.st = po.new_state(1, P, self.n)
        self.rng = po.Rng(SEED + rank)
        self.D, self.d0 = np.zeros((1, P), order="F"), np.zeros(1)

    def pack(self):
        s = self.st
        return np.concatenate([s["g_sum"].ravel(), s["w"].ravel(), s["g_sum_intercept"], s["intercept"]])

    def run(self, draws):
        stream = self.rng.stream(self.n, draws)
        for t0 in range(0, draws, BATCH):
            seg = stream[t0:t0 + BATCH]
            po.batch_gather(self.X, self.y, self.st, seg, self.D, self.d0, n_total=self.n, **KW)
            po.batch_sweep((P, self.n), self.st, seg.size, self.D, self.d0, n_total=self.n, **KW)

    def set(self, v):
        s = self.st
        s["g_sum"][:] = v[:P].reshape(1, P); s["w"][:] = v[P:2 * P].reshape(1, P)
        s["g_sum_intercept"][:] = v[2 * P:2 * P + 1]; s["intercept"][:] = v[2 * P + 1:]


for world, period in [(int(a), int(b)) for a, b in (x.split(":") for x in sys.argv[1:])]:
    shards = [Shard(r, world) for r in range(world)]
    wts = np.array([s.n / N for s in shards])
    per_epoch = max(1, (N // world) // period)
    wprev, ep, t, res = np.zeros(P), 0, time.time(), "max epochs"
    while ep < 150:
        for _ in range(per_epoch):
            ref = shards[0].pack()
            for s in shards:
                s.run(min(period, s.n))
            merged = ref + sum(wt * (s.pack() - ref) for wt, s in zip(wts, shards))
            for s in shards:
                s.set(merged)
        w = shards[0].st["w"].ravel().copy(); ep += 1
        if not np.all(np.isfinite(w)) or np.abs(w).max() > 1e6:
            res = "diverged"; break
        if np.abs(w).max() > 0 and np.abs(w - wprev).max() / np.abs(w).max() <= TOL:
            res = "converged"; break
        wprev = w
    print(f"world={world} period={period} draws/rank ({per_epoch} merges per epoch): {res} after {ep} epochs, "
          f"max|w|={np.abs(w).max():.4f} ({time.time()-t:.0f}s)", flush=True)
