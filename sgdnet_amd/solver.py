"""Python handle on the device-resident SAGA solver (sgdnet_solver_* of the C ABI).

Host-side plumbing only: every method is one call into libsgdnet_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FAMILIES, MODES, PENALTIES, check, dptr

STATE = {"w": 0, "intercept": 1, "g_memory": 2, "g_sum": 3, "g_sum_intercept": 4}


class RRng:
    """R-compatible Mersenne-Twister: RRng(seed) draws what set.seed(seed) draws."""

    def __init__(self, seed, sample_kind="Rejection"):
        """sample_kind: R's RNGkind(sample.kind = ...): "Rejection" (R >= 3.6.0, the default) or
        "Rounding" (what sample() did up to R 3.5.x, the R that printed the reference's docs)."""
        if sample_kind not in ("Rejection", "Rounding"):
            raise ValueError("sample_kind must be 'Rejection' or 'Rounding'")
        self.sample_kind = sample_kind
        self._L = _lib.load()
        self.state = _lib.Rng()
        self._L.sgdnet_rng_seed(C.byref(self.state), C.c_uint32(seed))

    def unif(self, count=1):
        return np.array([self._L.sgdnet_rng_unif(C.byref(self.state)) for _ in range(count)])

    def stream(self, n_samples, count):
        out = np.empty(count, dtype=np.uint32)
        self._L.sgdnet_rng_fill(C.byref(self.state), C.c_uint32(n_samples),
                                out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(count))
        return out

    # ---- R's sample() (R >= 3.6.0, sample.kind = "Rejection"; src/main/RNG.c R_unif_index and
    #      src/main/random.c do_sample of R itself), used by cv_sgdnet for the fold ids ----
    def _rbits(self, bits):
        v = 0
        for _ in range(0, bits + 1, 16):
            v = 65536 * v + int(np.floor(self.unif()[0] * 65536))
        return v & ((1 << bits) - 1) if bits < 64 else v

    def unif_index(self, dn):
        if self.sample_kind == "Rounding":         # R < 3.6.0: floor(dn * unif_rand())
            return int(np.floor(dn * self.unif()[0]))
        if dn <= 0:
            return 0
        bits = int(np.ceil(np.log2(dn)))
        while True:
            dv = self._rbits(bits)
            if dv < dn:
                return dv

    def sample(self, n, size=None):
        """sample(n, size) without replacement: a 1-based permutation prefix, as R returns it."""
        k = n if size is None else size
        if k < 2:
            return np.array([self.unif_index(n) + 1 for _ in range(k)], dtype=np.int64)
        x = list(range(n))
        out = np.empty(k, dtype=np.int64)
        m = n
        for i in range(k):
            j = self.unif_index(m)
            out[i] = x[j] + 1
            m -= 1
            x[j] = x[m]
        return out


def set_option(name, value):
    """sgdnet_set_option (include/sgdnet_hip.h): process-wide backend options, e.g. "virtual_shards"."""
    check(_lib.load().sgdnet_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int()
    check(_lib.load().sgdnet_get_option(name.encode(), C.byref(v)))
    return v.value


def link_peers(solvers):
    """sgdnet_solver_link_peers: the replica average of these sample-sharded solvers (one per GPU) runs over the
    virtual shards of all of them, inside their fused epoch kernels.  An empty / one-element list unlinks."""
    L = _lib.load()
    arr = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    check(L.sgdnet_solver_link_peers(arr, len(solvers)))


class option:
    """with sgdnet_amd.option("virtual_shards", 0): ...   -- the previous value comes back on exit."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def auto_batch(max_sample_sqnorm, max_feature_mean_sq):
    """Default staleness window of the batched mode (sgdnet_auto_batch of the C ABI)."""
    return int(_lib.load().sgdnet_auto_batch(float(max_sample_sqnorm), float(max_feature_mean_sq)))


def shard_window(window, draws_per_shard):
    """Window of a fit with virtual shards (sgdnet_shard_window of the C ABI): a slightly longer one when it saves
    the short last round of every shard's epoch."""
    return int(_lib.load().sgdnet_shard_window(int(window), int(draws_per_shard)))


class SagaSolver:
    """One problem resident in HBM: sample-major x, y and the five SAGA state arrays
    (reference src/sgdnet.cpp:187-198).

    x: scipy.sparse matrix of shape (p, n) in CSC form (column i = sample i), or a dense
       (p, n) array (sample i contiguous, Fortran order).
    y: (Ky, n) response, Fortran order.
    """

    def __init__(self, x, y, *, family, n_classes, fit_intercept=True, x_center_scaled=None,
                 n_total=0, device=0):
        import scipy.sparse as sp

        self._L = _lib.load()
        self.sparse = sp.issparse(x)
        p, n = x.shape
        self.n, self.p, self.K = n, p, n_classes
        self.n_shards = 0
        y = np.asfortranarray(np.asarray(y, dtype=np.float64).reshape(-1, n) if np.ndim(y) == 1
                              else np.asarray(y, dtype=np.float64))
        pb = _lib.Problem()
        pb.family = FAMILIES[family]
        pb.n_classes = n_classes
        pb.n_samples = n
        pb.n_total = n_total
        pb.n_features = p
        pb.fit_intercept = int(fit_intercept)
        keep = [y]
        if self.sparse:
            x = x.tocsc()
            x.sort_indices()
            ptr = np.ascontiguousarray(x.indptr, dtype=np.int64)
            idx = np.ascontiguousarray(x.indices, dtype=np.int32)
            val = np.ascontiguousarray(x.data, dtype=np.float64)
            pb.rowptr = ptr.ctypes.data_as(C.POINTER(C.c_int64))
            pb.colidx = idx.ctypes.data_as(C.POINTER(C.c_int32))
            pb.values = dptr(val)
            self.nnz = int(ptr[-1])
            self.row_nnz = np.diff(ptr)
            keep += [ptr, idx, val]
        else:
            xd = np.asfortranarray(x, dtype=np.float64)
            pb.x_dense = dptr(xd)
            self.nnz = n * p
            self.row_nnz = None
            keep.append(xd)
        if x_center_scaled is not None:
            c = np.ascontiguousarray(x_center_scaled, dtype=np.float64)
            pb.x_center_scaled = dptr(c)
            pb.standardize = 1
            keep.append(c)
        pb.y = dptr(y)
        pb.y_rows = y.shape[0]
        pb.device = device
        h = C.c_void_p()
        check(self._L.sgdnet_solver_create(C.byref(pb), C.byref(h)))
        self._h = h
        self.stream_len = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.sgdnet_solver_destroy(self._h)
            self._h = None

    __del__ = close

    def set_penalty(self, penalty, gamma, alpha, beta):
        check(self._L.sgdnet_solver_set_penalty(self._h, PENALTIES[penalty], gamma, alpha, beta))

    def _shape(self, name):
        return {"w": (self.K, self.p), "g_sum": (self.K, self.p), "g_memory": (self.K, self.n),
                "intercept": (self.K,), "g_sum_intercept": (self.K,)}[name]

    def get(self, name):
        out = np.zeros(self._shape(name), order="F")
        check(self._L.sgdnet_solver_get_state(self._h, STATE[name], dptr(out)))
        return out

    def set(self, name, value):
        v = np.asfortranarray(value, dtype=np.float64)
        assert v.shape == self._shape(name), (v.shape, self._shape(name))
        check(self._L.sgdnet_solver_set_state(self._h, STATE[name], dptr(v)))

    def upload_stream(self, stream):
        s = np.ascontiguousarray(stream, dtype=np.uint32)
        check(self._L.sgdnet_solver_upload_stream(
            self._h, s.ctypes.data_as(C.POINTER(C.c_uint32)), s.size))
        self.stream_len = s.size

    def generate_stream(self, rrng, count):
        """`count` draws floor(runif(0, n)) generated on the device from the RRng state
        (advanced exactly as rrng.stream(n, count) would advance it)."""
        check(self._L.sgdnet_solver_generate_stream(self._h, C.byref(rrng.state), count))
        self.stream_len = count

    # ---- the device-side sample-order pipeline (sgdnet_solver_rng_*) ----
    def rng_open(self, rrng, draws_per_epoch=None, generators=1, draws_per_run=0):
        """draws_per_run (virtual shards on a sample-sharded rank): the epoch is laid out run by run, every
        run shard after shard (sgdnet_solver_rng_layout); 0: one run = the epoch."""
        self._rrng = rrng
        check(self._L.sgdnet_solver_rng_layout(self._h, int(draws_per_run)))
        check(self._L.sgdnet_solver_rng_open(self._h, C.byref(rrng.state),
                                             self.n if draws_per_epoch is None else draws_per_epoch, generators))

    def rng_next(self):
        off = C.c_int64(0)
        check(self._L.sgdnet_solver_rng_next(self._h, C.byref(off)))
        return off.value

    def rng_done(self):
        check(self._L.sgdnet_solver_rng_done(self._h))

    def rng_close(self):
        check(self._L.sgdnet_solver_rng_close(self._h, C.byref(self._rrng.state)))

    def get_stream(self, offset=0, count=None):
        count = self.stream_len - offset if count is None else count
        out = np.empty(count, dtype=np.uint32)
        check(self._L.sgdnet_solver_get_stream(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               offset, count))
        return out

    def run(self, *, mode="exact", batch=0, stream_offset=0, draws_per_epoch=None, max_epochs=1,
            tol=0.0, losses=False):
        """Saga() for the current penalty; returns (epochs, converged[, losses])."""
        draws = self.n if draws_per_epoch is None else draws_per_epoch
        ep, conv = C.c_uint(0), C.c_int(0)
        lbuf = np.zeros(max_epochs) if losses else None
        check(self._L.sgdnet_solver_run(self._h, MODES[mode], batch, stream_offset, draws,
                                        max_epochs, tol, C.byref(ep), C.byref(conv),
                                        dptr(lbuf) if losses else None))
        if losses:
            return ep.value, bool(conv.value), lbuf[:ep.value]
        return ep.value, bool(conv.value)

    def enqueue_epochs(self, n_epochs, *, batch, stream_offset=0, draws_per_epoch=None):
        draws = self.n if draws_per_epoch is None else draws_per_epoch
        check(self._L.sgdnet_solver_enqueue_epochs(self._h, batch, stream_offset, draws, n_epochs))

    def sync(self):
        check(self._L.sgdnet_solver_sync(self._h))

    def profile_epoch(self, *, batch, stream_offset=0, draws_per_epoch=None):
        draws = self.n if draws_per_epoch is None else draws_per_epoch
        g, w = C.c_double(0), C.c_double(0)
        ng, nw = C.c_int(0), C.c_int(0)
        check(self._L.sgdnet_solver_profile_epoch(self._h, batch, stream_offset, draws, C.byref(g),
                                                  C.byref(ng), C.byref(w), C.byref(nw)))
        form = self._L.sgdnet_solver_gather_form(self._h, min(batch, draws))
        return dict(gather_ms=g.value, gather_launches=ng.value, sweep_ms=w.value,
                    sweep_launches=nw.value,
                    gather_kernel={1: "saga_batch_gather_lds_kernel", 2: "saga_binned_gather_kernel",
                                   3: "saga_vs_epoch_kernel"}.get(
                        form, "saga_batch_gather_kernel"))

    def deviance(self):
        out = C.c_double(0)
        check(self._L.sgdnet_solver_deviance(self._h, C.byref(out)))
        return out.value

    def last_change(self):
        """(max|w - w_prev|, max|w|) of the most recent convergence check."""
        ch, sz = C.c_double(0), C.c_double(0)
        check(self._L.sgdnet_solver_last_change(self._h, C.byref(ch), C.byref(sz)))
        return ch.value, sz.value

    def convergence(self, tol):
        c = C.c_int(0)
        check(self._L.sgdnet_solver_convergence(self._h, tol, C.byref(c)))
        return bool(c.value)

    # ---- multi-GPU merge hooks (device pointers; see sgdnet_amd/parallel.py) ----
    def delta_len(self):
        return int(self._L.sgdnet_solver_delta_len(self._h))

    def snapshot(self):
        check(self._L.sgdnet_solver_snapshot(self._h))

    def export_delta(self, device_ptr):
        check(self._L.sgdnet_solver_export_delta(self._h, C.c_void_p(device_ptr)))

    def apply_merged(self, device_ptr, w_weight):
        check(self._L.sgdnet_solver_apply_merged(self._h, C.c_void_p(device_ptr), w_weight))

    # ---- synchronous sample-sharded batches (sgdnet_amd/parallel.py: SyncShardedSaga) ----
    def sync_buffer_len(self):
        return int(self._L.sgdnet_solver_sync_buffer_len(self._h))

    def sync_bind(self, device_ptr):
        check(self._L.sgdnet_solver_sync_bind(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def sync_begin(self, stream_offset, draws_local):
        check(self._L.sgdnet_solver_sync_begin(self._h, stream_offset, draws_local))

    def sync_gather(self, t0_local, m_local, rnd):
        check(self._L.sgdnet_solver_sync_gather(self._h, t0_local, m_local, rnd))

    def sync_sweep(self, m_global, m_local, rnd):
        check(self._L.sgdnet_solver_sync_sweep(self._h, m_global, m_local, rnd))

    def sync_end(self, rounds):
        check(self._L.sgdnet_solver_sync_end(self._h, rounds))

    def set_virtual_shards(self, n_shards):
        check(self._L.sgdnet_solver_set_virtual_shards(self._h, n_shards))
        self.n_shards = n_shards

    def set_merge_period(self, draws_per_shard):
        check(self._L.sgdnet_solver_set_merge_period(self._h, draws_per_shard))

    def epoch_timing(self, enable):
        """(kernel milliseconds, launches) of the fused epoch launches since the previous call; enable: keep collecting."""
        ms, cnt = C.c_double(0), C.c_int(0)
        check(self._L.sgdnet_solver_epoch_timing(self._h, int(bool(enable)), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def peer_info(self):
        """This rank's block for link_ipc (bytes): hipIpc handles of its exchange buffer and counters, its shard sizes."""
        buf = C.create_string_buffer(self._L.sgdnet_solver_peer_info_bytes())
        check(self._L.sgdnet_solver_peer_info(self._h, buf))
        return buf.raw

    def link_ipc(self, rank, infos):
        """infos: every rank's peer_info() in rank order.  A barrier of the caller's must follow before the first epoch."""
        blob = b"".join(infos)
        check(self._L.sgdnet_solver_link_ipc(self._h, int(rank), len(infos), C.c_char_p(blob)))

    def set_cu_budget(self, cus):
        """CUs this solver's batched launches may fill (0: the device's): linked solvers sharing one GPU."""
        check(self._L.sgdnet_solver_set_cu_budget(self._h, int(cus)))

    def sharded_stream(self, rngs, epochs):
        """Host-side sample order for virtual shards: per epoch, shard after shard, n // V draws
        of each shard from its own generator (rngs: one RRng per shard)."""
        from .parallel import shard_bounds
        V = len(rngs)
        dps = self.n // V
        out = np.empty(epochs * V * dps, dtype=np.uint32)
        for e in range(epochs):
            for v in range(V):
                lo, hi = shard_bounds(self.n, V, v)
                seg = rngs[v].stream(hi - lo, dps).astype(np.int64) + lo
                out[(e * V + v) * dps:(e * V + v + 1) * dps] = seg
        return out

    def set_n_total(self, n_total):
        check(self._L.sgdnet_solver_set_n_total(self._h, n_total))

    def stream_handle(self):
        """The solver's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        return int(self._L.sgdnet_solver_stream(self._h) or 0)

    def export_delta_async(self, device_ptr, weight=1.0):
        check(self._L.sgdnet_solver_export_delta_weighted_async(self._h, C.c_void_p(device_ptr), weight))

    def apply_merged_async(self, device_ptr, w_weight):
        check(self._L.sgdnet_solver_apply_merged_async(self._h, C.c_void_p(device_ptr), w_weight))
