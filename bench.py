#!/usr/bin/env python3
"""Benchmark of the SAGA hot path: epochs/s + achieved HBM GB/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step is one SAGA epoch (n inner iterations over the whole job) of the batched
HIP path on synthetic data of BASELINE config 4 -- CSC 10M x 10k, 0.1 % nnz,
family=binomial, alpha=0.5, lambda=1/n, intercept, standardize=FALSE -- with the
sample order, the matrix, y and the solver state resident in HBM before the
timed region.  For N > 1 the driver launches one process per GPU
(torch.distributed.run) and the samples are sharded; the total work is fixed as N
grows ("strong" scaling).  --merge selects the exchange (sgdnet_amd/parallel.py):
  avg   (default) every rank runs the batched iteration on its own shard with local
        normalisation; every n/32 draws per rank one RCCL all-reduce averages the weighted
        state deltas [dG | dw | dgb | db].  Same fixed point, epochs-to-tolerance close to the
        single-GPU run's (reported in `convergence`);
  sync  every global batch is split across the ranks and its scatter accumulator is
        all-reduced before the sweep: exactly the single-GPU iterates, one collective per
        batch, no speed-up (--alt-merge reports the scheme that was not selected as `alt_merge`).

Inside every GPU the same averaging runs over up to 8 *virtual* shards (replicas over sample
ranges, one launch per batch of all shards, merged on the device; --vshards): the
per-GPU count follows the library's rule (every shard keeps 100 samples per feature), so the job
is an 8-way average at C4 for every N, and `convergence` reports its epochs to tolerance.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel
(the batched gather kernel) by the algorithmic bytes of SURVEY.md 8d divided by
its average dispatch duration (HIP events bound to every launch); `cpu_baseline` times the CPU
oracle (single-threaded restatement of the reference loop) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n, p, density, family, K, seed)
    "C4": (10_000_000, 10_000, 0.001, "binomial", 1, 4),
    "C3": (1_000_000, 1_000, 0.01, "binomial", 1, 3),
    # BASELINE config 5 (one GPU holds it: ~25 GB resident) and a 25x smaller problem of the same shape
    "C5": (50_000_000, 100_000, 0.0001, "multinomial", 10, 5),
    "C5s": (2_000_000, 100_000, 0.0001, "multinomial", 10, 5),
    "tiny": (100_000, 1_000, 0.01, "binomial", 1, 7),
}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def self_launch(n_gpus):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main(merge_override=None, note_override=""):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="staleness window; 0 = automatic (sgdnet_auto_batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-epochs", type=int, default=2)
    ap.add_argument("--no-convergence", action="store_true",
                    help="skip the epochs-to-tolerance leg (outside the timed region)")
    ap.add_argument("--conv-thresh", type=float, default=1e-6)
    ap.add_argument("--conv-max-epochs", type=int, default=400)
    ap.add_argument("--vshards", type=int, default=-1,
                    help="virtual shards per GPU (DESIGN.md 8); -1 = the library's rule, 0/1 = off")
    ap.add_argument("--fused-epoch", type=int, default=1, choices=[0, 1, 2],
                    help="the library's option fused_epoch -- 1 (default): an epoch on virtual shards is ONE launch; "
                         "2: the same with write-through hand-offs always; 0: one gather and one sweep launch per "
                         "batch (round 3's structure, for A/B runs)")
    ap.add_argument("--alt-merge", action="store_true",
                    help="N > 1: also measure the exchange scheme that --merge did not select")
    ap.add_argument("--no-alt-merge", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: every rank holds a full copy of the workload's sample count (n = N x n_workload, "
                         "lambda = 1/n); the default is BASELINE's strong scaling of the fixed problem")
    ap.add_argument("--merge", default="peers", choices=["peers", "avg", "sync"],
                    help="N > 1: peers (default) = the ranks' replicas are averaged INSIDE their epoch kernels by direct loads "
                         "from the other GPUs' exchange buffers (hipIpc mappings; no collective, one launch per epoch and rank; "
                         "falls back to avg where the link cannot be made); avg = periodic RCCL all-reduce of the weighted "
                         "state deltas between local runs; sync = a per-batch all-reduce of the scatter accumulator (exact "
                         "single-GPU iterates)")
    args = ap.parse_args()
    if merge_override:                                         # second pass after the peers scheme failed at run time (below)
        args.merge = merge_override

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # Called as plain `python bench.py --gpus N`: start one process per GPU ourselves.  This
        # process has not touched the GPU (no torch import yet), the ranks are CHILD processes, and
        # their stdout (rank 0's one JSON line) is relayed.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch
    import sgdnet_amd as sa
    from sgdnet_amd import data as D
    from sgdnet_amd.parallel import (HipShard, HipSyncShard, ShardedSaga, SyncShardedSaga, merge_segments,
                                     shard_bounds)

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the SAGA backend has no CPU fallback")
    sa.set_option("fused_epoch", args.fused_epoch)
    dist = None
    # SGDNET_BENCH_BACKEND=gloo (+ SGDNET_BENCH_ONE_GPU=1) rehearses the multi-rank control flow
    # on a single GPU; measured runs use nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("SGDNET_BENCH_BACKEND", "nccl")
    if os.environ.get("SGDNET_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_merge = world == 1 and os.environ.get("SGDNET_BENCH_FORCE_MERGE") == "1"
    if world > 1 or force_merge:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if dist.is_initialized():
            pass                                               # (second pass)
        elif backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    verbose = os.environ.get("SGDNET_BENCH_VERBOSE") == "1"

    def note(msg):
        if verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    n, p, density, family, K, seed = WORKLOADS[args.workload]
    if args.weak:
        n *= world
    lo, hi = shard_bounds(n, world, rank)
    n_local = hi - lo
    t_gen = time.time()
    prob = D.make_sparse_glm(n, p, density, family=family, n_classes=K, seed=seed, lo=lo, hi=hi)
    X = D.as_scipy(prob)
    t_gen = time.time() - t_gen
    note(f"generated shard [{lo}, {hi}) in {t_gen:.1f}s")

    # fit settings: alpha = 0.5, lambda = 1/n, intercept, no standardisation
    mix, lam = 0.5, 1.0 / n
    a_l2, b_l1 = (1.0 - mix) * lam, mix * lam
    row_sq = np.add.reduceat(prob["val"] ** 2, prob["ptr"][:-1])
    # class counts (multinomial) or the sum of the 0/1 response (binomial)
    ycount = (np.bincount(prob["y"].ravel().astype(np.int64), minlength=K) if family == "multinomial"
              else np.array([prob["y"].sum()])).astype(np.float64)
    mx = torch.tensor([float(row_sq.max())], dtype=torch.float64, device=red_dev)
    sm = torch.tensor(ycount, dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    max_sq = float(mx[0])
    gamma = D.step_size(max_sq, a_l2, True, family, n)          # src/utils.h:31-51
    if family == "multinomial":                                  # families.h:287-298
        lpi = np.log(np.asarray(sm.cpu()) / n)
        b0 = lpi - lpi.mean()
    else:
        ybar = min(max(float(sm[0]) / n, 1e-9), 1 - 1e-9)
        b0 = np.array([np.log(ybar / (1 - ybar))])              # families.h:190-201

    # staleness window: the library's default rule, 2 * L_max / diag(X'X/n) clamped to 65536
    col_sq = np.bincount(prob["idx"], weights=prob["val"] ** 2, minlength=p)
    cs = torch.tensor(col_sq, dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
    batch = args.batch if args.batch > 0 else sa.auto_batch(max_sq, float(cs.max()) / n)
    sync_mode = (world > 1 or force_merge) and args.merge == "sync"
    peers_mode = world > 1 and args.merge == "peers" and K == 1
    if not sync_mode:
        batch = min(batch, n_local)       # sync mode: `batch` is the GLOBAL staleness window
    note(f"gamma={gamma:.5g} batch={batch}")
    epochs_total = args.warmup + args.steps + 1                 # +1: the event-profiled epoch
    # virtual shards per GPU: the fit driver's rule (driver.cpp) -- up to 8 replicas while every
    # shard keeps 100 samples per feature; the synchronous mode does not use them
    V = 1
    if K == 1 and not sync_mode:
        while V < 8 and 2 * V * 100 * p <= n_local:
            V *= 2
        if peers_mode:                                         # the job stays (at most) an 8-way average, >= 2 shards per rank
            V = max(2, min(V, 8 // world))
        if args.vshards >= 0:
            V = max(1, args.vshards)
    S = sa.SagaSolver(X, prob["y"], family=family, n_classes=K, fit_intercept=True, n_total=n,
                      device=local_rank)
    S.set_penalty("elasticnet", gamma, a_l2, b_l1)
    S.set("intercept", b0)
    if os.environ.get("SGDNET_BENCH_ONE_GPU") == "1" and world > 1 and K == 1 and not sync_mode:
        S.set_cu_budget(256 // world - 16)                     # rehearsal: the ranks share one GPU's CUs
    peers_note = note_override
    if peers_mode:
        # link the ranks' solvers (sgdnet_solver_link_ipc: hipIpc mappings of the exchange buffers); every rank must
        # succeed, else everybody takes the RCCL scheme (--merge avg) -- the line's `merge` string says which ran
        ok = 1.0
        try:
            S.set_virtual_shards(V)
            S.set_merge_period(max(1, (n_local // V) // 4))
            infos = [None] * world
            dist.all_gather_object(infos, S.peer_info())
            S.link_ipc(rank, infos)
        except Exception as e:                                 # noqa: BLE001
            peers_note = f"{type(e).__name__}: {e}"[:200]
            note(f"peer link failed: {peers_note}")
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        dist.barrier()
        if float(flag[0]) < 0.5:
            peers_mode = False
            S.set_virtual_shards(0)                            # (also drops a half-made link)
            V = 1
            while V < 8 and 2 * V * 100 * p <= n_local:
                V *= 2
            if args.vshards >= 0:
                V = max(1, args.vshards)
    if args.batch <= 0 and V > 1 and not sync_mode:
        # the fit driver's rule for shards (sgdnet_shard_window): at most 1/8 beyond the rule's window when that saves the
        # short last round of every shard's epoch
        batch = sa.shard_window(batch, n_local // V)
        note(f"window for {V} shards of {n_local // V} draws: {batch}")
    # sample order: R's Mersenne-Twister, set.seed(config id [+ rank] [+ 100 shard]).  With virtual
    # shards every local run (an epoch, or a merge segment when N > 1) is laid out shard after shard
    # (include/sgdnet_hip.h: sgdnet_solver_set_virtual_shards)
    merged_job = (world > 1 or force_merge) and not sync_mode and not peers_mode
    # every shard (virtual or not) runs n / 32 draws between merges: a rank with V virtual shards
    # exchanges after V * n / 32 draws, when its own shards are averaged on the device anyway
    # measured with the HIP kernels (profiles/r02d_multi_gpu_emulation.txt): the averaging keeps the
    # single-process epochs-to-tolerance when every shard runs a QUARTER OF ITS OWN samples between
    # merges and holds >= 100 samples per feature -- 8 shards of C4 (n / 32, the round-1 rule), and also 16
    # and 32 shards of a 2x / 4x larger problem (--weak); 16 / 32 shards of the SAME 10M samples need 39 /
    # >150 epochs instead of 27.  SGDNET_BENCH_PERIOD_DIV forces n / div.
    div = os.environ.get("SGDNET_BENCH_PERIOD_DIV")
    shard_period = max(1, n // int(div)) if div else max(1, (n_local // max(1, V)) // 4 if V > 1 else n // 32)
    runs = merge_segments(n_local, n, V * min(batch, n_local // V), period=V * shard_period) \
        if merged_job else [n_local]
    # The sample order is R's single Mersenne-Twister stream of set.seed(config id + rank), produced ON THE
    # DEVICE INSIDE THE TIMED REGION, one epoch ahead on a side stream, by several generators kept on that
    # one stream by jump-ahead (sgdnet_solver_rng_*, as sgdnet_fit_* does) -- for every N: a sample-sharded
    # rank gets its epoch laid out merge segment by merge segment (sgdnet_solver_rng_layout), so the
    # N > 1 lines time the same configuration as the N = 1 line.  (The synchronous mode and
    # SGDNET_BENCH_RESIDENT_STREAM=1 keep a host-generated resident stream.)
    pipe = (not sync_mode and os.environ.get("SGDNET_BENCH_RESIDENT_STREAM") != "1"
            and (not merged_job or len(set(runs[:-1])) <= 1))
    gens = int(os.environ.get("SGDNET_BENCH_RNG_GENERATORS", "0")) or (
        min(32, max(8, n_local // 300000)) if n_local >= 200000 else 1)
    if V > 1:
        from sgdnet_amd.parallel import shard_bounds as sb
        if not peers_mode:                                     # (the linked solvers have theirs already)
            S.set_virtual_shards(V)
            S.set_merge_period(shard_period)
        rngs = [sa.RRng(seed + rank + 100 * v) for v in range(V)]

        def host_stream(epochs):
            parts = []
            for _ in range(epochs):
                for run in runs:
                    dps = run // V
                    for v in range(V):
                        lo_v, hi_v = sb(n_local, V, v)
                        parts.append((rngs[v].stream(hi_v - lo_v, dps).astype(np.int64) + lo_v).astype(np.uint32))
                    if run - dps * V:
                        parts.append(np.zeros(run - dps * V, dtype=np.uint32))  # positions no shard consumes
            return np.concatenate(parts)

        stream = None if pipe else host_stream(epochs_total)
    else:
        stream = None if pipe else sa.RRng(seed + rank).stream(n_local, n_local * epochs_total)
    bench_rng = sa.RRng(seed + rank)
    if pipe:
        S.rng_open(bench_rng, n_local, gens, draws_per_run=runs[0] if (merged_job and V > 1) else 0)
    else:
        S.upload_stream(stream)
    # device-ordered merge (no host sync inside an epoch) unless SGDNET_BENCH_FUSED=0
    fused = ((world > 1 or force_merge) and backend == "nccl"
             and os.environ.get("SGDNET_BENCH_FUSED", "1") == "1")
    dev = torch.device("cuda", local_rank)

    def make_job(mode):
        """mode 'sync' | 'avg' -> (epoch callable, shard, description, sync rounds)."""
        if mode == "sync":
            S.set_virtual_shards(0)
            S.set_n_total(n)
            sh = HipSyncShard(S, draws_per_epoch=n_local, device=dev, stage_on_host=(backend != "nccl"))
            sj = SyncShardedSaga(sh, n, world, batch, force_reduce=force_merge)
            desc = (f"sync: {backend} all-reduce of the scatter accumulator per global batch "
                    f"({sj.rounds} per epoch)" + (", stream-ordered" if backend == "nccl" else ""))
            return (lambda: sj.epoch(rank)), sh, desc, sj.rounds
        S.set_n_total(n_local)                    # local normalisation (sgdnet_amd/parallel.py)
        lb = min(batch, n_local)
        segs = runs if merged_job else (
            merge_segments(n_local, n, lb, period=shard_period) if (world > 1 or force_merge) else [n_local])
        sh = HipShard(S, batch=lb, draws_per_epoch=n_local, device=dev, weight=n_local / n,
                      stage_on_host=(backend != "nccl"), fused=fused)
        sj = ShardedSaga(sh, world, segs, force_merge=force_merge)
        desc = ("none" if world == 1 and not force_merge else
                (f"(peers scheme not used: {peers_note}) " if peers_note else "") +
                f"avg: locally normalised shard runs, {backend} all-reduce of the weighted state deltas every "
                f"{segs[0]} draws per rank ({len(segs)} per epoch)" + (", stream-ordered" if fused else ""))
        return sj.epoch, sh, desc, 0

    if peers_mode:
        class _NoShard:                                        # the epoch is the single-GPU loop below: nothing to exchange here
            offset = 0

            def close(self):
                pass

        S.set_n_total(n_local)
        run_epoch, shard, sync_rounds = None, _NoShard(), 0
        merge_desc = (f"peers: {world * V}-way replica average inside the ranks' epoch kernels (direct loads from the peers' "
                      f"exchange buffers, hipIpc), every {shard_period} draws per shard; one launch per epoch and rank, no collective")
    else:
        run_epoch, shard, merge_desc, sync_rounds = make_job("sync" if sync_mode else "avg")
    if pipe and merged_job:
        job_epoch = run_epoch

        def run_epoch():                          # noqa: F811 -- this epoch's draws, its local runs and merges
            shard.offset = S.rng_next()
            job_epoch()
            S.rng_done()
    elif pipe:
        lb_pipe = min(batch, n_local)

        def run_epoch():                          # noqa: F811 -- draws of this epoch, epoch, hand the slot back
            off_ = S.rng_next()
            S.enqueue_epochs(1, batch=lb_pipe, stream_offset=off_, draws_per_epoch=n_local)
            S.rng_done()

    def fence():
        S.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note("solver resident, stream uploaded")
    timed_kernel = {}

    def timed_epochs(epoch_fn):
        for _ in range(args.warmup):
            epoch_fn()
        fence()
        S.epoch_timing(True)                       # dispatch events of exactly the timed region's epoch launches
        t0 = time.perf_counter()
        for _ in range(args.steps):
            epoch_fn()
        fence()
        dt = time.perf_counter() - t0
        timed_kernel["ms"], timed_kernel["launches"] = S.epoch_timing(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        return dt

    if peers_mode:
        # The linked epoch kernels wait for each other across GPUs with bounded spins: should the peer traffic not work on
        # this node (it could only be rehearsed with the ranks sharing one GPU), every rank's launch gives up within
        # seconds and raises.  Then everybody starts over with the RCCL scheme -- a line, not a crash.
        ok = 1.0
        try:
            if os.environ.get("SGDNET_BENCH_TEST_PEERS_FAILURE") == str(rank):   # (tests: this rank never launches)
                raise RuntimeError("test: no launch on this rank")
            elapsed = timed_epochs(run_epoch)
        except Exception as e:                                 # noqa: BLE001
            peers_note = f"epochs failed: {type(e).__name__}: {e}"[:200]
            note(peers_note)
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag[0]) < 0.5:
            try:
                if pipe:
                    S.rng_close()
                S.close()
            except Exception:                                  # noqa: BLE001
                pass
            return main(merge_override="avg", note_override=peers_note or "epochs failed on another rank")
    else:
        elapsed = timed_epochs(run_epoch)

    note(f"timed region done: {elapsed:.4f}s")
    # dominant kernel, HIP events around every launch of one more epoch (same stream)
    off = S.rng_next() if pipe else shard.offset
    # sync mode: this rank's share of a global batch, same (global-atomic) gather kernel as the
    # timed region; the profiled epoch runs without the exchange, the state is reset below
    local_batch = min(batch, n_local) if not sync_mode else max(1, -(-n_local // sync_rounds))
    # a sample-sharded rank profiles ONE local run (the draws between two merges are laid out per run)
    prof_draws = runs[0] if merged_job else n_local
    prof = S.profile_epoch(batch=min(local_batch, prof_draws), stream_offset=off, draws_per_epoch=prof_draws)
    epoch_draws = S.get_stream(off, prof_draws) if pipe else stream[off:off + prof_draws]
    if pipe:
        S.rng_done()
    alg_bytes_epoch = D.algorithmic_bytes(S.row_nnz, epoch_draws, K)
    if prof["gather_kernel"] == "saga_vs_epoch_kernel" and timed_kernel.get("launches", 0) == args.steps and not merged_job:
        # the dominant kernel IS the epoch: its dispatch durations over the timed region itself (one launch per step)
        prof = dict(prof, gather_ms=timed_kernel["ms"] / args.steps, gather_launches=1, timed_region_events=True)
    gather_s = prof["gather_ms"] * 1e-3
    achieved = alg_bytes_epoch / gather_s / 1e9
    alg_bytes_epoch_job = alg_bytes_epoch * (n_local / prof_draws)     # this rank's whole epoch
    alg_all = torch.tensor([alg_bytes_epoch_job], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(alg_all, op=dist.ReduceOp.SUM)
    job_gbps = float(alg_all[0]) / (elapsed / args.steps) / 1e9

    seen = torch.ones(1, dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)   # how many ranks really took part in the collectives
    out = {
        "metric": "saga_epochs_per_sec",
        "value": args.steps / elapsed,
        "unit": "epochs/s",
        "n_gpus": world,
        "n_ranks_seen": int(seen[0]),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "achieved_hbm_gbps_job": job_gbps,
        "config": {
            "workload": f"{args.workload}: synthetic CSC {n}x{p}, {density:.4%} nnz, family={family}, "
                        f"alpha={mix}, lambda=1/n, intercept, standardize=FALSE",
            "mode": "batched", "batch": batch, "samples_per_gpu": n_local, "virtual_shards": V,
            "sample_order": (f"R MT19937 set.seed({seed}{'+rank' if world > 1 else ''}), with replacement: one stream"
                             f"{' per rank' if world > 1 else ''}, {gens} generators kept on it by "
                             "jump-ahead, generated on the device inside the timed region (one epoch ahead, side stream)"
                             if pipe else f"R MT19937 set.seed({seed}+rank [+100 shard]), with replacement, resident before the timed region"),
            "merge": merge_desc,
            "gen_s": round(t_gen, 2),
        },
        "roofline": {
            "bound": "hbm", "kernel": prof["gather_kernel"],
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "launches": prof["gather_launches"],
            "timed_over": ("the timed region's own launches (HIP events bound to every dispatch)" if prof.get("timed_region_events")
                           else "one more epoch after the timed region (HIP events bound to every dispatch)"),
            "avg_launch_us": 1e3 * prof["gather_ms"] / max(1, prof["gather_launches"]),
            "algorithmic_bytes_per_launch": alg_bytes_epoch / max(1, prof["gather_launches"]),
            "sweep_avg_launch_us": 1e3 * prof["sweep_ms"] / max(1, prof["sweep_launches"]),
            # floors of this access pattern, measured with no compute at all in the product's launch geometry on a FRESH
            # segment of the sample order per repetition (scripts/microbench/gather_patterns.hip,
            # profiles/r03j_gather_patterns_microbench.txt, r03b_*): microseconds per 2^20 draws -> GB/s at this
            # workload's algorithmic bytes per draw.  (Round 2's 39.5 % floor replayed one segment out of the Infinity
            # Cache and is gone.)
            "ceilings": {"hbm_copy_measured": 6290.0,
                         "one_random_128B_line_per_draw": (alg_bytes_epoch / max(1, prof_draws)) * 2 ** 20 / 24.1e-6 / 1e9,
                         "line_plus_exchange_into_the_line": (alg_bytes_epoch / max(1, prof_draws)) * 2 ** 20 / 51.6e-6 / 1e9,
                         "line_plus_plain_store_into_the_line": (alg_bytes_epoch / max(1, prof_draws)) * 2 ** 20 / 42.6e-6 / 1e9,
                         "source": "profiles/r03j_gather_patterns_microbench.txt"}
            if K == 1 else
                        {"hbm_copy_measured": 6290.0, "random_256B_records": 6650.0},
        },
    }

    note("profiled epoch done")

    # Epochs-to-tolerance of the same job from a cold start (outside the timed region): epochs/s
    # alone says nothing about a merge rule that trades statistical efficiency for throughput.
    def cold_start():
        S.set("w", np.zeros((K, p)))
        S.set("g_sum", np.zeros((K, p)))
        S.set("g_sum_intercept", np.zeros(K))
        S.set("g_memory", np.zeros((K, n_local)))
        S.set("intercept", b0)

    def convergence_leg(epoch_fn, sh, max_epochs):
        cold_start()
        if pipe:                                   # the epoch function draws its own sample order
            S.convergence(args.conv_thresh)
            fence()
            tconv = time.perf_counter()
            done, conv_ep = False, 0
            while not done and conv_ep < max_epochs:
                epoch_fn()
                S.sync()
                done = S.convergence(args.conv_thresh)
                if world > 1:                      # one decision for all ranks (they hold the same w)
                    flag = torch.tensor([1.0 if done else 0.0], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    done = bool(flag[0] > 0.5)
                conv_ep += 1
            tconv = time.perf_counter() - tconv
            return {"thresh": args.conv_thresh, "epochs": conv_ep, "converged": bool(done), "seconds": tconv,
                    "deviance": S.deviance() if world == 1 else None,
                    "note": "cold start, ConvergenceCheck on the merged coefficients every epoch; "
                            "includes the per-epoch sample order and one host synchronisation per epoch"}
        crng = sa.RRng(seed + 1000 + rank)
        S.convergence(args.conv_thresh)            # w_prev <- 0
        fence()
        tconv = time.perf_counter()
        done, conv_ep = False, 0
        while not done and conv_ep < max_epochs:
            if V > 1 and merged_job and S.n_shards == V:
                S.upload_stream(host_stream(1))    # laid out per merge segment: generated on the host
            else:
                S.generate_stream(crng, n_local)   # this epoch's draws, generated on the device
            sh.offset = 0
            epoch_fn()
            S.sync()
            torch.cuda.synchronize()
            done = S.convergence(args.conv_thresh)
            if world > 1:                          # one decision for all ranks (they hold the same w)
                flag = torch.tensor([1.0 if done else 0.0], dtype=torch.float64, device=red_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                done = bool(flag[0] > 0.5)
            conv_ep += 1
        tconv = time.perf_counter() - tconv
        return {"thresh": args.conv_thresh, "epochs": conv_ep, "converged": bool(done), "seconds": tconv,
                "deviance": S.deviance() if world == 1 else None,
                "note": "cold start, ConvergenceCheck on the merged coefficients every epoch; "
                        "includes per-epoch device RNG and host synchronisation"}

    if not args.no_convergence:
        try:                                        # a reported extra: never at the price of the headline
            out["convergence"] = convergence_leg(run_epoch, shard, args.conv_max_epochs)
            note(f"convergence leg: {out['convergence']['epochs']} epochs in {out['convergence']['seconds']:.3f}s")
        except Exception as e:                      # noqa: BLE001
            out["convergence"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # N > 1: the other exchange scheme on the same resident problem, reported beside the headline
    # (`sync` is exact but pays a collective per batch; `epoch` is the scheme of the
    # north star, fast per epoch, and does not reach the tolerance at this lambda)
    if (world > 1 or force_merge) and args.alt_merge and not args.no_alt_merge:
        alt = "avg" if sync_mode else "sync"
        if sync_mode:
            shard.close()                           # unbind the sync buffer
        cold_start()
        S.upload_stream(stream)
        alt_epoch, alt_shard, alt_desc, _ = make_job(alt)
        alt_dt = timed_epochs(alt_epoch)
        out["alt_merge"] = {"merge": alt_desc, "value": args.steps / alt_dt, "unit": "epochs/s",
                            "ms_per_step": 1e3 * alt_dt / args.steps}
        if not args.no_convergence:
            out["alt_merge"]["convergence"] = convergence_leg(alt_epoch, alt_shard,
                                                              min(args.conv_max_epochs, 150))
        note(f"alt merge {alt}: {out['alt_merge']['value']:.1f} epochs/s")
    # HBM traffic of the same kernel from the committed PMC passes (rocprofv3 cannot run inside
    # this process); attached only when workload, batch and kernel match that profile
    try:
        import hashlib
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        src_sha = hashlib.sha256(open(os.path.join(ROOT, "sgdnet_amd", "csrc", "saga_batched.hip"), "rb").read()).hexdigest()[:16]
        # only a profile of THIS kernel source, workload, window and shard count is quoted
        if (pmc.get("kernel_source_sha16"), pmc["workload"], pmc["batch"], pmc["n_gpus"], pmc["kernel"],
                pmc.get("virtual_shards", 1)) == (src_sha, args.workload, batch, world, prof["gather_kernel"], V):
            out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/pmc_latest.json"
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po   # the checker/baseline leg only
        st = po.new_state(K, p, n_local)
        st["intercept"][:] = b0
        ce = max(1, args.cpu_epochs)
        # bounded sample (10-30 s of one core): at most 2e7 inner iterations of the reference loop;
        # on config 5 that is a fraction of one epoch, run as the first draws of the same stream
        cpu_draws = min(ce * n_local, 20_000_000)
        if stream is None:                         # the same R stream, generated on the host for the CPU loop
            stream = sa.RRng(seed).stream(n_local, cpu_draws)
        tc = time.perf_counter()
        if cpu_draws == ce * n_local:
            po.saga(X, prob["y"], st, family=family, penalty="elasticnet", gamma=gamma, alpha=a_l2,
                    beta=b_l1, fit_intercept=True, max_iter=ce, tol=0.0, stream=stream[:cpu_draws])
        else:
            # the first cpu_draws inner iterations of the epoch, on the whole data set
            po.saga(X, prob["y"], st, family=family, penalty="elasticnet", gamma=gamma, alpha=a_l2,
                    beta=b_l1, fit_intercept=True, max_iter=1, tol=0.0, stream=stream[:cpu_draws],
                    epoch_len=cpu_draws)
        tc = time.perf_counter() - tc
        cpu_bytes = D.algorithmic_bytes(S.row_nnz, stream[:cpu_draws], K)
        out["cpu_baseline"] = {
            "value": cpu_draws / n_local / tc, "unit": "epochs/s", "cores": 1, "kind": "port",
            "sample": f"{cpu_draws / n_local:.3g} epochs ({cpu_draws} inner iterations) of the same workload and "
                      f"sample stream, exact reference iteration, gcc -O2, {tc:.1f} s",
            "algorithmic_gbps": cpu_bytes / tc / 1e9,
            "host_cpus": os.cpu_count(),
        }
    if pipe:
        S.rng_close()
    if pipe and not merged_job:
        # the same kernel WITHOUT the sample-order generators beside it (all 256 CUs, draws already resident):
        # `roofline.frac` above is what the timed epochs experience, this is the kernel's own figure
        try:
            S.sync()
            alone = S.profile_epoch(batch=min(batch, n_local), stream_offset=0, draws_per_epoch=n_local)
            bytes_alone = D.algorithmic_bytes(S.row_nnz, S.get_stream(0, n_local), K)
            out["roofline"]["kernel_alone"] = {
                "avg_launch_us": 1e3 * alone["gather_ms"] / max(1, alone["gather_launches"]),
                "launches": alone["gather_launches"],
                "achieved": bytes_alone / (alone["gather_ms"] * 1e-3) / 1e9,
                "frac": bytes_alone / (alone["gather_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "one more epoch after the timed region with the generators idle and the full grid"}
        except Exception as e:                      # noqa: BLE001 -- an extra, never at the price of the line
            out["roofline"]["kernel_alone"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    S.close()
    if world > 1 or force_merge:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
