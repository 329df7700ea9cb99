"""Epochs to tolerance of the two-rank averaging job on the binned multinomial kernels (one GPU, gloo)."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def worker(rank, world, port, n, p, K, batch, tol, gamma):
    import torch
    import torch.distributed as dist
    import sgdnet_amd as sa
    from sgdnet_amd import data as D
    from sgdnet_amd.parallel import HipShard, ShardedSaga, merge_segments, shard_bounds
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_bounds(n, world, rank)
    pr = D.make_sparse_glm(n, p, 0.005, family="multinomial", n_classes=K, seed=31, lo=lo, hi=hi)
    nl = hi - lo
    S = sa.SagaSolver(D.as_scipy(pr), pr["y"], family="multinomial", n_classes=K, n_total=nl)
    S.set_penalty("elasticnet", gamma, 1e-5, 1e-5)
    shard = HipShard(S, batch=batch, draws_per_epoch=nl, device=torch.device("cuda", 0), weight=nl / n, stage_on_host=True)
    job = ShardedSaga(shard, world, merge_segments(nl, n, batch))
    rng = sa.RRng(50 + rank)
    S.convergence(tol)
    prev = None
    for e in range(400):
        S.generate_stream(rng, nl)
        shard.offset = 0
        job.epoch()
        S.sync()
        w = S.get("w")
        if prev is not None and rank == 0 and (e % 20 == 0):
            print(world, "epoch", e, "max change rel", np.abs(w - prev).max() / np.abs(w).max(), flush=True)
        prev = w.copy()
    S.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    for world in (1, 2):
        for gamma in (0.01, 0.003):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            print("world", world, "gamma", gamma, flush=True)
            mp.spawn(worker, args=(world, port, 65536, 2000, 10, 4096, 1e-10, gamma), nprocs=world, join=True)
