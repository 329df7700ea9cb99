"""Experiment: the 8 shards of C4 as 8 independent solvers on 8 streams (32 workgroups each), no merges,
against the fused launch of 8 virtual shards -- does inter-shard asynchrony hide the per-launch fixed cost?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["SGDNET_LDS_GRID"] = os.environ.get("SHARD_GRID", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import sgdnet_amd as sa
from sgdnet_amd import data as D
from sgdnet_amd.parallel import shard_bounds

n, p, dens, V, E, batch = 10_000_000, 10_000, 0.001, 8, 12, 131072
solvers = []
for v in range(V):
    lo, hi = shard_bounds(n, V, v)
    pr = D.make_sparse_glm(n, p, dens, family="binomial", seed=4, lo=lo, hi=hi)
    S = sa.SagaSolver(D.as_scipy(pr), pr["y"], family="binomial", n_classes=1, n_total=hi - lo)
    S.set_penalty("elasticnet", 0.02, 0.5 / n, 0.5 / n)
    S.upload_stream(sa.RRng(1 + v).stream(hi - lo, (hi - lo) * (E + 2)))
    solvers.append((S, hi - lo))
for S, nl in solvers:
    S.enqueue_epochs(1, batch=batch, draws_per_epoch=nl)
for S, nl in solvers:
    S.sync()
for rep in range(2):
    t = time.time()
    for e in range(E):
        for S, nl in solvers:
            S.enqueue_epochs(1, batch=batch, stream_offset=(e + 1) * nl, draws_per_epoch=nl)
    for S, nl in solvers:
        S.sync()
    dt = (time.time() - t) / E
    print(f"8 solvers x {os.environ['SGDNET_LDS_GRID']} workgroups on their own streams: {dt * 1e3:.3f} ms per epoch of all shards "
          f"({1 / dt:.0f} epochs/s), no merges", flush=True)
# one solver at a time (no overlap), for scale
t = time.time()
for S, nl in solvers:
    S.enqueue_epochs(E, batch=batch, stream_offset=nl, draws_per_epoch=nl)
    S.sync()
print(f"the same solvers one after the other: {(time.time() - t) / E * 1e3:.3f} ms per epoch of all shards", flush=True)
for S, nl in solvers:
    S.close()
