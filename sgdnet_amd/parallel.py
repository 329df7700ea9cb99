"""Sample-sharded SAGA across the GPUs of one node (BASELINE north star).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Rank r holds a
contiguous shard of the samples and their gradient-memory rows; the model state
(w, g_sum, intercept, g_sum_intercept) is replicated.  Two exchange schemes:

* ShardedSaga -- periodic averaging with local normalisation (default).  Every rank runs the
  batched SAGA iteration on ITS shard as if the shard were the whole data set (g_sum increments
  divided by n_local), starting from the replicated (w, g_sum).  After every `period` local draws
  one all-reduce sums the packed deltas [dG | dw | dgb | db] (2*K*p + 2*K doubles, 160 KB at 10k
  features), each pre-scaled by the shard's weight n_local / n_total, and every rank continues
  from ref + sum.  At the optimum every delta is zero, so the fixed point is the reference's.
  The direction a rank follows in expectation is  grad f_r(w) - grad f_r(w_ref) + g_sum_ref :
  its own shard with full curvature plus a stale correction for the others (the DANE
  correction), which is what makes the local runs agree with each other.  It is stable when the
  local runs are short: measured (scripts/dev/merge_rule_experiment3.py and the HIP kernels,
  DESIGN.md 8) with period = n_total / 32 draws per rank the epochs-to-tolerance are those of
  one process for 2-16 ranks and lambda from 0.1/n to 10/n; n_total / 16 costs up to 1.6x the
  epochs on one of the two benchmark shapes, n_total / 2 does not converge.  The first version
  of this file summed dG with weight 1 over globally normalised local runs and averaged w once
  per epoch: that one does not converge at lambda = 1/n (every rank then sees its own shard's
  curvature diluted by 1 / world).
* SyncShardedSaga -- every GLOBAL batch of B draws is split across the ranks, one all-reduce per
  batch sums the scatter accumulator, every rank applies the same sweep: the iterates are
  exactly those of the single-GPU batched mode over the interleaved sample order; one
  collective per batch (77 per epoch at the 10M x 10k benchmark shape) and no speed-up.

The local solver is duck-typed (`snapshot`, `local_run`, `export_delta`, `apply_merged`) so
that the merge logic is testable on CPU with gloo; the product binding is HipShard below.  The
torch import is plumbing (process group, device buffer), not compute.
"""


def merge_segments(n_local, n_total, batch, period=None):
    """Draws per local run between two merges: [seg, seg, ..., remainder], summing to n_local.
    period (draws per rank) defaults to n_total / 32, rounded down to a whole number of batches
    (at least one)."""
    if period is None:
        period = max(1, n_total // 32)
    b = max(1, min(batch, n_local))
    seg = max(b, (period // b) * b)
    out = []
    left = n_local
    while left > 0:
        take = min(seg, left)
        out.append(take)
        left -= take
    return out


class HipShard:
    """Adapter: SagaSolver (libsgdnet_hip.so) + a torch device buffer for the all-reduce.

    fused=True wraps the solver's HIP stream as a torch ExternalStream and issues the
    all-reduce from it: snapshot, local epoch, delta export, RCCL all-reduce and merge are then
    ordered on the device with no host synchronisation inside an epoch (at 8 GPUs a local
    epoch is ~0.4 ms, so three host syncs per epoch would cost a quarter of it).
    """

    def __init__(self, solver, *, batch, draws_per_epoch, device, weight=1.0, stage_on_host=False,
                 fused=False):
        import torch

        from . import _lib
        _lib.require_torch_first()

        self.solver = solver
        self.batch = batch
        self.draws = draws_per_epoch
        self.weight = float(weight)           # n_local / n_total: this shard's share of the average
        self.buf = torch.zeros(solver.delta_len(), dtype=torch.float64, device=device)
        # functional rehearsals with a CPU-only backend (gloo) reduce a host copy
        self.host = torch.zeros_like(self.buf, device="cpu") if stage_on_host else None
        self.offset = 0
        self.ext = None
        if fused and not stage_on_host:
            torch.cuda.synchronize()          # buf's zero fill ran on torch's stream
            self.ext = torch.cuda.ExternalStream(solver.stream_handle(), device=device)

    def snapshot(self):
        self.solver.snapshot()

    def local_run(self, draws):
        self.solver.enqueue_epochs(1, batch=min(self.batch, draws), stream_offset=self.offset,
                                   draws_per_epoch=draws)
        self.offset += draws

    def export_delta(self):
        self.solver.export_delta_async(self.buf.data_ptr(), self.weight)
        if self.ext is not None:
            return self.buf
        self.solver.sync()
        if self.host is not None:
            self.host.copy_(self.buf)
            return self.host
        return self.buf

    def apply_merged(self, buf):
        import torch

        if self.ext is not None:
            self.solver.apply_merged_async(self.buf.data_ptr(), 1.0)
            return
        if self.host is not None:
            self.buf.copy_(buf)
        torch.cuda.synchronize()                        # all-reduce ran on torch's stream
        self.solver.apply_merged(self.buf.data_ptr(), 1.0)


class ShardedSaga:
    """Periodic averaging of locally normalised shard runs; identical on every rank.

    segments: draws of the local runs of one job epoch (merge_segments); one all-reduce after
    each.  The shard's deltas arrive pre-scaled by its weight (HipShard.export_delta), so the
    merged state is ref + sum over ranks."""

    def __init__(self, shard, world_size, segments, group=None, force_merge=False):
        self.shard = shard
        self.world = world_size
        self.segments = list(segments)
        self.group = group
        self.force_merge = force_merge        # rehearse the merge path with a single rank

    def epoch(self):
        sh = self.shard
        if self.world == 1 and not self.force_merge:
            for seg in self.segments:
                sh.local_run(seg)
            return
        import torch.distributed as dist

        ext = getattr(sh, "ext", None)
        sh.snapshot()                         # later local runs start from apply_merged's result
        for seg in self.segments:
            sh.local_run(seg)
            buf = sh.export_delta()
            if ext is not None:
                import torch
                with torch.cuda.stream(ext):  # the collective is ordered on the solver's stream
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            sh.apply_merged(buf)


class HipSyncShard:
    """Adapter for SyncShardedSaga: SagaSolver + the torch buffer its D / intercept slots live in."""

    def __init__(self, solver, *, draws_per_epoch, device, stage_on_host=False):
        import torch

        from . import _lib
        _lib.require_torch_first()

        self.solver = solver
        self.draws = draws_per_epoch
        self.buf = torch.zeros(solver.sync_buffer_len(), dtype=torch.float64, device=device)
        torch.cuda.synchronize()
        solver.sync_bind(self.buf.data_ptr())
        self.host = torch.zeros_like(self.buf, device="cpu") if stage_on_host else None
        self.ext = torch.cuda.ExternalStream(solver.stream_handle(), device=device)
        self.offset = 0

    def sync_begin(self):
        self.solver.sync_begin(self.offset, self.draws)

    def sync_gather(self, t0, m, rnd):
        self.solver.sync_gather(t0, m, rnd)

    def sync_reduce(self, group):
        import torch
        import torch.distributed as dist

        if self.host is not None:             # CPU-only backend rehearsal (gloo)
            self.solver.sync()
            self.host.copy_(self.buf)
            dist.all_reduce(self.host, op=dist.ReduceOp.SUM, group=group)
            self.buf.copy_(self.host)
            torch.cuda.synchronize()
            return
        # the caller holds reduce_context(): the collective is issued on the solver's stream,
        # ordered between this batch's gather and sweep with no host synchronisation
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group)

    def reduce_context(self):
        import contextlib

        import torch

        return contextlib.nullcontext() if self.host is not None else torch.cuda.stream(self.ext)

    def sync_sweep(self, m_global, m_local, rnd):
        self.solver.sync_sweep(m_global, m_local, rnd)

    def sync_end(self, rounds):
        self.solver.sync_end(rounds)
        self.offset += self.draws

    def close(self):
        self.solver.sync_bind(0)


def sync_rounds(n_total, batch_global):
    """Number of global batches per job epoch."""
    return max(1, -(-n_total // max(1, batch_global)))


def round_share(n_local, rounds, k):
    """Draws [lo, hi) of a rank's local epoch stream that belong to global batch k."""
    return (k * n_local) // rounds, ((k + 1) * n_local) // rounds


class SyncShardedSaga:
    """Synchronous driver: global batches split across ranks, one all-reduce per batch."""

    def __init__(self, shard, n_total, world_size, batch_global, group=None, force_reduce=False):
        self.shard = shard
        self.world = world_size
        self.group = group
        self.force_reduce = force_reduce      # rehearse the collective with a single rank
        sizes = [hi - lo for lo, hi in (shard_bounds(n_total, world_size, r) for r in range(world_size))]
        self.sizes = sizes
        # every rank contributes at least one draw to every round
        self.rounds = max(1, min(sync_rounds(n_total, batch_global), min(sizes)))
        # global draw count of every round: identical on all ranks by construction
        self.m_global = [sum(round_share(nr, self.rounds, k)[1] - round_share(nr, self.rounds, k)[0]
                             for nr in sizes) for k in range(self.rounds)]

    def epoch(self, rank):
        sh, R = self.shard, self.rounds
        n_local = self.sizes[rank]
        import contextlib

        reduce = self.world > 1 or self.force_reduce
        ctx = getattr(sh, "reduce_context", None) if reduce else None
        sh.sync_begin()
        with (ctx() if ctx else contextlib.nullcontext()):   # entered once per epoch, not per batch
            for k in range(R):
                lo, hi = round_share(n_local, R, k)
                sh.sync_gather(lo, hi - lo, k)
                if reduce:
                    sh.sync_reduce(self.group)
                sh.sync_sweep(self.m_global[k], hi - lo, k)
        sh.sync_end(R)


def shard_bounds(n_total, world, rank):
    """Contiguous sample shard of `rank`: [lo, hi)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
