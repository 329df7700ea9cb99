"""CPU emulation (oracle) of a per-epoch merge with LOCAL normalisation: every rank runs a SAGA
epoch on its shard as if the shard were the whole data set (g_sum increments weighted 1/n_local),
starting from the merged (w, g_sum); merge = shard-size-weighted average of both deltas.
Expected direction on rank r: grad f_r(w) - grad f_r(w0) + g_sum0  (DANE-style correction)."""
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
TOL = float(os.environ.get("TOL", "1e-6"))

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
                beta=B_L1, max_iter=1, tol=0.0, stream=stream, batch=min(BATCH, self.n), n_total=self.n)
    def set(self, v):
        s = self.st
        s["g_sum"][:] = v[:P].reshape(1, P); s["w"][:] = v[P:2*P].reshape(1, P)
        s["g_sum_intercept"][:] = v[2*P:2*P+1]; s["intercept"][:] = v[2*P+1:]

final = {}
for world in (1, 2, 4, 8):
    shards = [Shard(r, world) for r in range(world)]
    wts = np.array([s.n / N for s in shards])
    wprev = np.zeros(P); ep = 0; t = time.time()
    while ep < 300:
        ref = shards[0].pack()
        for s in shards: s.epoch()
        merged = ref + sum(wt * (s.pack() - ref) for wt, s in zip(wts, shards))
        for s in shards: s.set(merged)
        w = shards[0].st["w"].ravel().copy(); ep += 1
        if not np.all(np.isfinite(w)) or np.abs(w).max() > 1e6: ep = -ep; break
        if np.abs(w).max() > 0 and np.abs(w - wprev).max() / np.abs(w).max() <= TOL: break
        wprev = w
    final[world] = np.r_[w, shards[0].st["intercept"]]
    err = np.abs(final[world] - final[1]).max() / np.abs(final[1]).max()
    print(f"world={world} local-normalised merge: epochs={ep} max|w|={np.abs(w).max():.4f} "
          f"rel diff vs world=1: {err:.2e} ({time.time()-t:.0f}s)", flush=True)
