"""CPU emulation (oracle) of the sample-sharded SAGA merge: epochs-to-tolerance vs world size
and merge weight.  Informs the default w_weight of sgdnet_amd/parallel.py."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from sgdnet_amd import data as D
from sgdnet_amd.parallel import shard_bounds

N, P, DENS, SEED = 400_000, 2000, 0.005, 17
pr = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
A_L2 = B_L1 = 0.5 / N
GAMMA = D.step_size(row_sq.max(), A_L2, True, "binomial", N)
BATCH = 4000
TOL = 1e-6

class Shard:
    def __init__(self, rank, world):
        lo, hi = shard_bounds(N, world, rank)
        p2 = D.make_sparse_glm(N, P, DENS, family="binomial", seed=SEED, lo=lo, hi=hi)
        self.X, self.y, self.n = D.as_scipy(p2), p2["y"], hi - lo
        self.st = po.new_state(1, P, self.n); self.rng = po.Rng(SEED + rank)
    def pack(self):
        s = self.st
        return np.concatenate([s["g_sum"].ravel(), s["w"].ravel(), s["g_sum_intercept"], s["intercept"]])
    def epoch(self):
        stream = self.rng.stream(self.n, self.n)
        po.saga(self.X, self.y, self.st, family="binomial", penalty="elasticnet", gamma=GAMMA, alpha=A_L2,
                beta=B_L1, max_iter=1, tol=0.0, stream=stream, batch=min(BATCH, self.n), n_total=N)
    def apply(self, ref, m, ww):
        s = self.st
        s["g_sum"][:] = (ref[:P] + m[:P]).reshape(1, P); s["w"][:] = (ref[P:2*P] + ww * m[P:2*P]).reshape(1, P)
        s["g_sum_intercept"][:] = ref[2*P:2*P+1] + m[2*P:2*P+1]; s["intercept"][:] = ref[2*P+1:] + ww * m[2*P+1:]

for world in (1, 2, 4, 8):
    for ww in ([1.0] if world == 1 else [1.0 / world, 2.0 / world, 1.0 / np.sqrt(world), 1.0]):
        shards = [Shard(r, world) for r in range(world)]
        wprev = np.zeros(P); ep = 0; t = time.time()
        while ep < 300:
            refs = [s.pack() for s in shards]
            for s in shards: s.epoch()
            tot = sum(s.pack() - r for s, r in zip(shards, refs))
            for s, r in zip(shards, refs): s.apply(r, tot, ww)
            w = shards[0].st["w"].ravel().copy(); ep += 1
            if not np.all(np.isfinite(w)) or np.abs(w).max() > 1e6: ep = -ep; break
            if np.abs(w).max() > 0 and np.abs(w - wprev).max() / np.abs(w).max() <= TOL: break
            wprev = w
        print(f"world={world} w_weight={ww:.3f}: epochs={ep} ({time.time()-t:.0f}s)", flush=True)
