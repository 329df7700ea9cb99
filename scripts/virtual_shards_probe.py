"""Probe: V virtual shards on ONE GPU (V solvers, V streams, grids of 256/V workgroups) with the
periodic-averaging merge, against one solver -- does concurrency amortise the per-batch fixed cost?"""
import sys, os, time
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2
os.environ["SGDNET_LDS_GRID"] = str(256 // V)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sgdnet_amd as sa
from sgdnet_amd import data as D
from sgdnet_amd.parallel import merge_segments, shard_bounds
n, p, dens, seed = 10_000_000, 10_000, 0.001, 4
batch = 131072
epochs = 4
solvers, bufs, segs = [], [], None
for v in range(V):
    lo, hi = shard_bounds(n, V, v)
    pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=seed, lo=lo, hi=hi)
    nl = hi - lo
    S = sa.SagaSolver(D.as_scipy(pr), pr["y"], family="binomial", n_classes=1, n_total=nl)
    if v == 0:
        row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
        gamma = D.step_size(row_sq.max() * 1.2, 0.5 / n, True, "binomial", n)
    S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n)
    S.upload_stream(sa.RRng(seed + v).stream(nl, nl * (epochs + 1)))
    solvers.append((S, nl))
    bufs.append(torch.zeros(S.delta_len(), dtype=torch.float64, device="cuda"))
    segs = merge_segments(nl, n, batch) if V > 1 else [nl]
torch.cuda.synchronize()
def epoch(e):
    off = [e * nl for _, nl in solvers]
    for (S, nl) in solvers:
        S.snapshot()
    for seg in segs:
        for i, (S, nl) in enumerate(solvers):
            S.enqueue_epochs(1, batch=min(batch, seg), stream_offset=off[i], draws_per_epoch=seg)
            off[i] += seg
        if V > 1:
            for i, (S, nl) in enumerate(solvers):
                S.export_delta_async(bufs[i].data_ptr(), nl / n)
            for S, _ in solvers:
                S.sync()
            tot = bufs[0].clone()
            for b in bufs[1:]:
                tot += b
            torch.cuda.synchronize()
            for S, _ in solvers:
                S.apply_merged_async(tot.data_ptr(), 1.0)
            for S, _ in solvers:
                S.sync()
epoch(0)
for S, _ in solvers: S.sync()
t = time.time()
for e in range(1, epochs):
    epoch(e)
for S, _ in solvers: S.sync()
dt = (time.time() - t) / (epochs - 1)
print(f"V={V}: {dt*1e3:.2f} ms/epoch ({1/dt:.0f} epochs/s), {len(segs)} merges per epoch, grid {256//V} per shard", flush=True)
