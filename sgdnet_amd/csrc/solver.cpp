// Device-resident SAGA solver behind the sgdnet_solver_* C ABI (include/sgdnet_hip.h).
//
// Owns the HBM copies of the sample-major data and of the five state arrays the
// reference keeps alive along the lambda path (src/sgdnet.cpp:187-198), and
// drives the epoch kernels of saga_exact.hip / saga_batched.hip on its own HIP
// stream.  Batched epochs are captured once per (batch, draws) shape into a
// hipGraph and replayed, because an epoch is hundreds of microsecond-scale
// launches (DESIGN.md "Launch structure").
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <chrono>
#include <atomic>
#include <vector>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "setup_device.hpp"

namespace sgdnet {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

}  // namespace sgdnet

using namespace sgdnet;

struct sgdnet_solver {
  SagaDev d{};
  bool sparse = false;
  int device = 0;
  hipStream_t st = nullptr;
  LamParams lam{};
  LamParams* lam_dev = nullptr;
  LamParams lam_dev_mirror{};   // what lam_dev holds (push_lam skips an upload that would change nothing)
  bool lam_dev_valid = false;
  // pinned staging ring for the asynchronous upload of `lam`: the host copy keeps changing
  // (stream_base, batch_seq) while earlier uploads may still be in flight
  static constexpr int kLamSlots = 8;
  LamParams* lam_stage = nullptr;
  hipEvent_t lam_ev[kLamSlots] = {};
  int lam_slot = 0;
  // owned device buffers
  std::vector<void*> owned;
  double* ref = nullptr;        // snapshot for the multi-GPU merge
  double* LS_dev = nullptr;     // lag_scaling table (exact sparse)
  int64_t LS_len = 0;
  double LS_alpha = -1.0, LS_gamma = -1.0;
  int* out_dev = nullptr;
  uint32_t* stream_dev = nullptr;
  int64_t stream_len = 0;
  int64_t stream_cap = 0;
  uint32_t* rng_dev = nullptr;  // 625 words: the device copy of a sgdnet_rng
  // sample-order pipeline of the fit driver (solver_rng_*): the next epoch's draws are
  // generated on a side stream while the current epoch runs
  struct RngPipe {
    bool open = false;
    hipStream_t st = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr};   // slot filled (side stream)
    hipEvent_t freed[2] = {nullptr, nullptr};   // slot consumed (solver stream)
    uint32_t* state[2] = {nullptr, nullptr};    // generation g reads state[g & 1], writes state[(g + 1) & 1]
    int64_t n = 0;
    int64_t gens = 0, used = 0;
    static constexpr int kMaxGen = 64;
    int G = 1;                                  // generators side by side (segments of an epoch's stream)
    // G > 1: ONE R stream.  state[.][g] is the state at the START of generator g's segment;
    // the next epoch's starts are those states jumped n draws ahead (poly_n), the ends the generation
    // kernel leaves go to `ends` and are not used
    uint32_t* poly_n = nullptr;
    uint32_t* ends = nullptr;
    int64_t run_len = 0;   // virtual shards: draws per run of the layout (0: one run = the epoch)
    // the slot holds the generators' raw words (left to the fused epoch kernel, which converts its own shares)
    bool raw[2] = {false, false};
    // generators inside the fused epoch kernel (SagaDev::rngdev): the generation that the next fused launch is to
    // produce, or -1; a generation still pending when its draws are asked for is produced on the side stream after all
    RngDev* dev = nullptr;
    int64_t pending_gen = -1;
  } pipe;
  int64_t nnz = 0;
  bool penalty_set = false;
  // cached epoch graph
  // captured epochs, one per (batch, draws) shape; gexec is the one selected by ensure_graph
  struct GraphEntry {
    int64_t batch, draws;
    hipGraph_t graph;
    hipGraphExec_t exec;
    int fused;                  // > 0: the epoch is ONE launch of saga_vs_epoch_kernel (the value of option fused_epoch)
  };
  std::vector<GraphEntry> graphs;
  hipGraphExec_t gexec = nullptr;
  bool w_prev_valid = false;
  double last_change = 0.0, last_size = 0.0;
  int64_t slab_cap = 0;         // doubles the slab buffer can hold
  double* own_D = nullptr;      // the solver's own D / d0 slots while a sync buffer is bound
  std::vector<void*> vs_owned;  // virtual-shard replicas
  int64_t vs_period = 0;        // draws per shard between device-side merges (0: n / 32)
  double* own_d0 = nullptr;
  // binned form (saga_batched.hip): ranges built once, bins sized for the current batch
  bool bin_ranges_ready = false;
  int64_t bin_batch = 0;        // the batch the bins and gcb were sized for
  void* bin_bufs[3] = {nullptr, nullptr, nullptr};   // bins, gcb, bin_off
  std::vector<double> bin_mass;  // non-zeros of every feature range
  std::vector<double> bin_sumsq; // sum over samples of (its non-zeros inside the range)^2
  double bin_slack = 8.0, bin_slack_built = 0.0;   // standard deviations of room in every bin
  bool bin_disabled = false;     // a bin kept overflowing: the solver runs the atomic form (K <= 16) from now on
  bool bin_overflowed = false;   // the last sync found an overflow (the epochs since the previous sync are void)
  // one-response sparse fits with compact records: the batched kernels keep the gradient memory inside the
  // records (saga_batched.hip "Compact records"), everything else (exact mode, the host) sees the K x n array
  bool m_in_rec = false;
  // fused epoch of the virtual shards (saga_vs_epoch_kernel): switched off for this solver once a launch could not
  // become resident (a GPU shared with another process); the separate launches take over
  bool fused_off = false;
  bool fused_in_graph = false;   // the captured epochs use it
  int fused_abort_seen = 0;      // LamParams::fused_abort as the last ConvergenceCheck read it
  FusedPeers* peers_dev = nullptr;   // sgdnet_solver_link_peers
  // sgdnet_solver_epoch_timing: dispatch start / stop events of every fused epoch launch (the benchmark's timed region)
  bool time_epochs = false;
  std::vector<hipEvent_t> epoch_ev;
  std::vector<void*> ipc_opened;   // sgdnet_solver_link_ipc: the peers' buffers as mapped here
};

namespace {
// SGDNET_TRACE: host-side split of a batched epoch (development aid)
double g_trace_launch = 0.0, g_trace_conv = 0.0, g_trace_graph = 0.0;
long g_trace_epochs = 0;


template <typename T>
int dev_alloc(sgdnet_solver* s, T** out, size_t count, bool zero) {
  void* p = nullptr;
  const size_t bytes = sizeof(T) * (count ? count : 1);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return SGDNET_ENOMEM;
  }
  s->owned.push_back(p);
  if (zero) SGD_HIP_TRY(hipMemsetAsync(p, 0, bytes, s->st));
  *out = static_cast<T*>(p);
  return SGDNET_OK;
}

template <typename T>
int dev_upload(sgdnet_solver* s, T** out, const T* host, size_t count) {
  int rc = dev_alloc(s, out, count, false);
  if (rc) return rc;
  if (count) SGD_HIP_TRY(hipMemcpyAsync(*out, host, sizeof(T) * count, hipMemcpyHostToDevice, s->st));
  return SGDNET_OK;
}

// what the device copy of `lam` holds, as far as the host can know it (the kernels advance stream_base and
// batch_seq themselves: lam_advance mirrors that; the ConvergenceCheck / loss scratch fields are the device's own)
bool lam_on_device(const sgdnet_solver* s) {
  if (!s->lam_dev_valid) return false;
  const LamParams &a = s->lam, &b = s->lam_dev_mirror;
  return a.penalty == b.penalty && a.gamma == b.gamma && a.alpha == b.alpha && a.beta == b.beta && a.r_full == b.r_full &&
         a.ls_full == b.ls_full && a.r_tail == b.r_tail && a.ls_tail == b.ls_tail && a.m_full == b.m_full &&
         a.m_tail == b.m_tail && a.stream_base == b.stream_base && a.stream_wrap == b.stream_wrap &&
         a.draws_per_epoch == b.draws_per_epoch && a.batch_seq == b.batch_seq && a.stream_raw == b.stream_raw &&
         a.rng_generate == b.rng_generate;
}

// host mirror of end_epoch (saga_batched.hip)
void lam_advance(sgdnet_solver* s, int64_t draws, int batches) {
  for (LamParams* q : {&s->lam, &s->lam_dev_mirror}) {
    int64_t sb = q->stream_base + draws;
    if (q->stream_wrap > 0 && sb >= q->stream_wrap) sb -= q->stream_wrap;
    q->stream_base = sb;
    q->batch_seq += batches;
  }
}

// Uploads `lam` unless the device already holds exactly these values: back-to-back epochs of one lambda then
// run graph after graph with no copy in between (the 120-byte upload is a blit kernel of ~20 us on the solver's
// stream: 2 % of a C4 epoch).
int push_lam(sgdnet_solver* s) {
  if (lam_on_device(s)) return SGDNET_OK;
  const int slot = s->lam_slot;
  s->lam_slot = (slot + 1) % sgdnet_solver::kLamSlots;
  SGD_HIP_TRY(hipEventSynchronize(s->lam_ev[slot]));   // the slot's previous upload has completed
  s->lam_stage[slot] = s->lam;
  SGD_HIP_TRY(hipMemcpyAsync(s->lam_dev, &s->lam_stage[slot], sizeof(LamParams), hipMemcpyHostToDevice,
                             s->st));
  SGD_HIP_TRY(hipEventRecord(s->lam_ev[slot], s->st));
  s->lam_dev_mirror = s->lam;
  s->lam_dev_valid = true;
  return SGDNET_OK;
}

// Epochs that consume the sample-order pipeline's two-epoch buffer alternate between its halves: the device
// wraps stream_base there itself, so consecutive epochs need no upload.
int64_t stream_wrap_for(const sgdnet_solver* s, int64_t stream_offset, int64_t draws) {
  const auto& P = s->pipe;
  return (P.open && draws == P.n && (stream_offset == 0 || stream_offset == P.n)) ? 2 * P.n : 0;
}

// r^m and LS_m = sum_{k<m} r^k for r = 1 - alpha*gamma: closed form of the
// reference's cumulative lag_scaling table (src/saga-sparse.h:229-240).
void batch_factors(double alpha, double gamma, int64_t m, double* r_m, double* ls_m) {
  const double a = 1.0 - (1.0 - alpha * gamma);  // 1 - r, exact for the rounded r
  if (a == 0.0) {
    *r_m = 1.0;
    *ls_m = (double)m;
  } else {
    const double e = expm1((double)m * log1p(-a));  // r^m - 1
    *r_m = 1.0 + e;
    *ls_m = -e / a;
  }
}

void drop_graph(sgdnet_solver* s) {
  for (auto& g : s->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  s->graphs.clear();
  s->gexec = nullptr;
}

// Where the gradient memory of a one-response sparse fit lives: inside the compact records while batched
// epochs run (the gather reads it with the record and stores it back in place), in the K x n array for the
// exact kernels and the host.  A move is one pass over the samples; it happens when the mode changes or the
// host reads / writes g_memory, not per epoch.  The kernels take SagaDev by value, so captured graphs go.
int m_to_record(sgdnet_solver* s) {
  if (s->m_in_rec || !s->d.cP || s->d.K != 1) return SGDNET_OK;
  int rc = launch_m_move(s->d, 1, s->st);
  if (rc) return rc;
  s->m_in_rec = true;
  s->d.m_rec = 1;
  s->d.m_base = s->d.cP + 120;
  s->d.m_stride = 128;
  drop_graph(s);
  return SGDNET_OK;
}

int m_to_array(sgdnet_solver* s) {
  if (!s->m_in_rec) return SGDNET_OK;
  int rc = launch_m_move(s->d, 0, s->st);
  if (rc) return rc;
  s->m_in_rec = false;
  s->d.m_rec = 0;
  s->d.m_base = reinterpret_cast<char*>(s->d.M);
  s->d.m_stride = 8;
  drop_graph(s);
  return SGDNET_OK;
}

// Binned form of the batched iteration for K x p tables that fit no LDS (saga_batched.hip
// "Binned form").  Ranges: contiguous features of equal non-zero mass, at most
// binned_max_range_features(K) wide; built once per solver from a device histogram of the feature
// ids.  Bins and the gradient-change buffer are sized for the batch.
int ensure_binned(sgdnet_solver* s, int64_t batch) {
  SagaDev& d = s->d;
  static const int allow = exp_env_int("SGDNET_BINNED", 1);
  // more than 16 classes: the only batched form there is (a wavefront per draw), whatever the sizes
  const bool want = allow && !s->bin_disabled && s->sparse && !d.xd && d.rec && d.idx && d.K <= 64 && !d.force_global &&
                    d.p < (1ll << 31) &&
                    (d.K > 16 || (sizeof(double) * (size_t)d.K * (size_t)d.p > 80 * 1024 && batch >= 4096));
  if (!want) {
    if (d.R > 0 && d.bins) {             // e.g. a tiny batch after a large one: fall back to the atomic form
      d.bins = nullptr;
      drop_graph(s);
    }
    return SGDNET_OK;
  }
  if (!s->bin_ranges_ready) {
    s->bin_ranges_ready = true;                       // whatever comes out: the work below is done once
    unsigned* counts = nullptr;
    SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&counts), sizeof(unsigned) * (size_t)d.p));
    struct Free { void* q; ~Free() { (void)hipFree(q); } } free_counts{counts};
    int rc = launch_col_count(d, s->nnz, counts, s->st);
    if (rc) return rc;
    std::vector<unsigned> h((size_t)d.p);
    SGD_HIP_TRY(hipMemcpyAsync(h.data(), counts, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost, s->st));
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    const int64_t fmax = (int64_t)binned_max_range_features(d.K);
    static const int target_env = exp_env_int("SGDNET_BIN_RANGES", 0);
    // range sweep workgroups: two of them fit a CU beside each other (64 KB of LDS each at the widest range), and the
    // launch carries one more workgroup for the intercept -- 2 x 256 - 12 ranges keep the whole launch resident in one
    // wave (round 4: the rule's rounding made 513 ranges + 1 of 512, the last two workgroups ran after everybody else)
    const int target = target_env > 0 ? target_env : 500;
    const double per = std::max(1.0, (double)s->nnz / target);
    std::vector<int32_t> lo{0};
    {
      double mass = 0.0;
      int64_t width = 0;
      for (int64_t j = 0; j < d.p; ++j) {
        if (width > 0 && (width >= fmax || mass + 0.5 * h[(size_t)j] >= per)) {
          lo.push_back((int32_t)j);
          mass = 0.0;
          width = 0;
        }
        mass += h[(size_t)j];
        ++width;
      }
      lo.push_back((int32_t)d.p);
    }
    const int R = (int)lo.size() - 1;
    if (R > 2048) return SGDNET_OK;      // staging counters of the gather would not fit: d.R stays 0 (no binned form)
    std::vector<uint16_t> fr((size_t)d.p);
    s->bin_mass.assign((size_t)R, 0.0);
    int wmax = 1;
    for (int r = 0; r < R; ++r) {
      wmax = std::max(wmax, (int)(lo[(size_t)r + 1] - lo[(size_t)r]));
      for (int64_t j = lo[(size_t)r]; j < lo[(size_t)r + 1]; ++j) {
        fr[(size_t)j] = (uint16_t)r;
        s->bin_mass[(size_t)r] += h[(size_t)j];
      }
    }
    d.range_max = wmax;
    int32_t* lo_dev = nullptr;
    uint16_t* fr_dev = nullptr;
    uint16_t* rc_dev = nullptr;
    // coarse cells of 2^shift features (at most 2047 of them): the range of a cell's first feature
    int shift = 7;
    while (((d.p + (1ll << shift) - 1) >> shift) + 1 > 2048) ++shift;
    const int n_coarse = (int)((d.p + (1ll << shift) - 1) >> shift);
    std::vector<uint16_t> rcv((size_t)n_coarse + 1);
    for (int c = 0; c < n_coarse; ++c) rcv[(size_t)c] = fr[(size_t)c << shift];
    rcv[(size_t)n_coarse] = (uint16_t)(R - 1);
    rc = dev_upload(s, &lo_dev, lo.data(), lo.size());
    if (!rc) rc = dev_upload(s, &fr_dev, fr.data(), fr.size());
    if (!rc) rc = dev_upload(s, &rc_dev, rcv.data(), rcv.size());
    unsigned* bc = nullptr;
    int* be = nullptr;
    if (!rc) rc = dev_alloc(s, &bc, (size_t)R, true);
    if (!rc) rc = dev_alloc(s, &be, 1, true);
    if (rc) return rc;
    // second moment of a sample's entries per range (sizes the bins below)
    s->bin_sumsq.assign((size_t)R, 0.0);
    if (d.ptr) {
      unsigned long long* sq = nullptr;
      SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sq), sizeof(unsigned long long) * (size_t)R));
      Free free_sq{sq};
      rc = launch_range_moment(d, fr_dev, sq, R, s->st);
      if (rc) return rc;
      std::vector<unsigned long long> hsq((size_t)R);
      SGD_HIP_TRY(hipMemcpyAsync(hsq.data(), sq, sizeof(unsigned long long) * (size_t)R, hipMemcpyDeviceToHost, s->st));
      SGD_HIP_TRY(hipStreamSynchronize(s->st));
      for (int r = 0; r < R; ++r) s->bin_sumsq[(size_t)r] = (double)hsq[(size_t)r];
    } else {
      s->bin_sumsq = s->bin_mass;                     // no sample-major pointers: one entry per arrival
    }
    SGD_HIP_TRY(hipStreamSynchronize(s->st));        // the uploads read host vectors that die here
    int ks = 1;
    while (ks < d.K) ks *= 2;
    d.KS = ks;
    if (ks != d.K) {
      double* wp = nullptr;
      rc = dev_alloc(s, &wp, (size_t)d.p * (size_t)ks, true);
      if (rc) return rc;
      d.wpad = wp;
    } else {
      d.wpad = d.w;
    }
    d.R = R;
    d.range_lo = lo_dev;
    d.feat_range = fr_dev;
    d.range_coarse = rc_dev;
    d.coarse_shift = shift;
    d.n_coarse = n_coarse;
    d.bin_count = bc;
    d.bin_err = be;
    if (getenv("SGDNET_TRACE")) fprintf(stderr, "[sgdnet]   binned form: %d feature ranges (<= %lld features each)\n", R, (long long)fmax);
  }
  if (d.R <= 0) return SGDNET_OK;
  if (batch != s->bin_batch || !s->bin_bufs[0] || s->bin_slack != s->bin_slack_built) {
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    for (void*& q : s->bin_bufs) {
      if (q) (void)hipFree(q);
      q = nullptr;
    }
    // Range r receives the entries of `batch` uniformly drawn samples: a sum of `batch` per-sample counts c_ir
    // with mean mass_r / n and second moment sumsq_r / n.  Capacity = mean + slack standard deviations + room
    // for the largest rows (slack starts at 8 and doubles when a bin has overflowed: solver_grow_bins).
    std::vector<int64_t> off((size_t)d.R + 1, 0);
    for (int r = 0; r < d.R; ++r) {
      const double mean = (double)batch * s->bin_mass[(size_t)r] / (double)d.n;
      const double var = (double)batch * s->bin_sumsq[(size_t)r] / (double)d.n;
      double cap = mean + s->bin_slack * std::sqrt(var) + 64.0 * s->bin_slack;
      cap = std::min(cap, (double)batch * (double)d.range_max + 512.0);   // nothing can send more than this
      off[(size_t)r + 1] = off[(size_t)r] + (int64_t)cap;
    }
    SGD_HIP_TRY(hipMalloc(&s->bin_bufs[0], (size_t)16 * (size_t)off[(size_t)d.R]));
    SGD_HIP_TRY(hipMalloc(&s->bin_bufs[1], sizeof(double) * (size_t)batch * (size_t)d.KS));
    SGD_HIP_TRY(hipMalloc(&s->bin_bufs[2], sizeof(int64_t) * off.size()));
    SGD_HIP_TRY(hipMemcpy(s->bin_bufs[2], off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice));
    s->bin_batch = batch;
    s->bin_slack_built = s->bin_slack;
    drop_graph(s);
  }
  if (!d.bins) drop_graph(s);
  d.bins = static_cast<char*>(s->bin_bufs[0]);
  d.gcb = static_cast<double*>(s->bin_bufs[1]);
  d.bin_off = static_cast<const int64_t*>(s->bin_bufs[2]);
  return SGDNET_OK;
}

// Dense x whose K x p accumulator fits no LDS (saga_batched.hip "tiled"): the gather leaves the
// batch's gradient changes in d.gcb and D is formed feature tile by feature tile.
int ensure_dense_tiled(sgdnet_solver* s, int64_t batch) {
  SagaDev& d = s->d;
  // the forms that hand the gradient changes of a batch to an accumulate pass: K x p tables beyond the LDS, and 17..64
  // classes whatever the table (saga_dense_cl_gather_kernel)
  if (!d.xd || d.K > 64 || (d.K <= 16 && sizeof(double) * (size_t)d.K * (size_t)d.p <= 80 * 1024)) return SGDNET_OK;
  if (batch > s->bin_batch || !s->bin_bufs[1]) {
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    if (s->bin_bufs[1]) (void)hipFree(s->bin_bufs[1]);
    s->bin_bufs[1] = nullptr;
    SGD_HIP_TRY(hipMalloc(&s->bin_bufs[1], sizeof(double) * (size_t)batch * (size_t)d.K));
    s->bin_batch = batch;
    drop_graph(s);
  }
  d.gcb = static_cast<double*>(s->bin_bufs[1]);
  d.KS = d.K;
  return SGDNET_OK;
}

int set_batch_shape(sgdnet_solver* s, int64_t batch, int64_t draws) {
  if (batch < 1) batch = 1;
  {
    const int rcm = m_to_record(s);          // batched kernels: the gradient memory rides in the records
    if (rcm) return rcm;
  }
  if (s->d.V > 1 && vs_eligible(s->d, (int)batch)) {
    // per-shard batches; the scratch is sized for the launch that carries V of them
    const int64_t dps = draws / s->d.V;
    if (batch > dps) batch = dps;
    s->d.v_dps = dps;
    const int64_t slab_need = (int64_t)s->d.v_bps * s->d.V * s->d.K * s->d.p;
    if (slab_need > s->slab_cap) {
      SGD_HIP_TRY(hipStreamSynchronize(s->st));
      if (s->d.slab) SGD_HIP_TRY(hipFree(s->d.slab));
      s->d.slab = nullptr;
      SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d.slab), sizeof(double) * (size_t)slab_need));
      s->slab_cap = slab_need;
      drop_graph(s);
    }
    const int64_t tail = dps - (dps / batch) * batch;
    s->lam.m_full = batch;
    s->lam.m_tail = tail;
    batch_factors(s->lam.alpha, s->lam.gamma, batch, &s->lam.r_full, &s->lam.ls_full);
    batch_factors(s->lam.alpha, s->lam.gamma, tail, &s->lam.r_tail, &s->lam.ls_tail);
    s->lam.draws_per_epoch = draws;
    return SGDNET_OK;
  }
  if (batch > draws) batch = draws;
  {
    int rcb = ensure_binned(s, batch);
    if (!rcb) rcb = ensure_dense_tiled(s, batch);
    if (rcb) return rcb;
  }
  // scratch must cover the full batches AND the tail batch, whose launch geometry (and even
  // its gather form) can differ
  const int64_t tail_m = draws - (draws / batch) * batch;
  int64_t slab_need = batch_gather_slab_doubles(s->d, (int)batch);
  if (tail_m > 0) slab_need = std::max(slab_need, batch_gather_slab_doubles(s->d, (int)tail_m));
  if (slab_need > s->slab_cap) {
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    if (s->d.slab) SGD_HIP_TRY(hipFree(s->d.slab));
    s->d.slab = nullptr;
    SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d.slab), sizeof(double) * (size_t)slab_need));
    s->slab_cap = slab_need;
    drop_graph(s);
  }
  const int64_t full = draws / batch;
  const int64_t tail = draws - full * batch;
  s->lam.m_full = batch;
  s->lam.m_tail = tail;
  batch_factors(s->lam.alpha, s->lam.gamma, batch, &s->lam.r_full, &s->lam.ls_full);
  batch_factors(s->lam.alpha, s->lam.gamma, tail, &s->lam.r_tail, &s->lam.ls_tail);
  s->lam.draws_per_epoch = draws;
  return SGDNET_OK;
}

bool vs_active(const sgdnet_solver* s, int64_t batch);
bool vs_fused_active(const sgdnet_solver* s);

// The epoch(s) about to be enqueued read the stream at `stream_offset`.  A slot of the sample-order pipeline that was
// left raw goes to the fused epoch kernel as it is (LamParams::stream_raw; the kernel converts it, so the slot counts
// as converted from here on); for any other consumer it is converted now, on the solver's stream (which has waited
// for the generators: solver_rng_acquire).
int prepare_stream_slot(sgdnet_solver* s, int64_t batch, int64_t stream_offset, int64_t draws, int n_epochs) {
  auto& P = s->pipe;
  s->lam.stream_raw = 0;
  s->lam.rng_generate = 0;
  if (!P.open) return SGDNET_OK;
  const bool one_slot = n_epochs == 1 && draws == P.n && (stream_offset == 0 || stream_offset == P.n);
  const int slot = stream_offset == 0 ? 0 : 1;
  const bool fused = one_slot && vs_active(s, batch) && vs_fused_active(s);
  // the launch that consumes generation `used` also produces the pending generation used + 1 (its spare workgroups)
  if (fused && s->d.rngdev && P.pending_gen >= 0 && P.pending_gen == P.used + 1 && slot == (int)(P.used & 1)) {
    s->lam.rng_generate = 1;
    P.pending_gen = -1;
  }
  if (!(P.raw[0] || P.raw[1])) return SGDNET_OK;
  if (fused && P.raw[slot]) {
    s->lam.stream_raw = 1;
    P.raw[slot] = false;
    return SGDNET_OK;
  }
  for (int q = 0; q < 2; ++q) {
    if (!P.raw[q] || (one_slot && q != slot)) continue;
    if (P.pending_gen >= 0 && (int)(P.pending_gen & 1) == q) continue;   // not produced yet (solver_rng_acquire will)
    if (!one_slot) SGD_HIP_TRY(hipStreamSynchronize(P.st));      // a slot that may still be generated
    int rc = launch_rng_convert(s->stream_dev + (int64_t)q * P.n, P.n, (uint32_t)s->d.n, s->st, s->d.V, s->d.v_size,
                                P.run_len);
    if (rc) return rc;
    P.raw[q] = false;
  }
  return SGDNET_OK;
}

int n_batches(int64_t batch, int64_t draws) {
  if (batch < 1) batch = 1;
  if (batch > draws) batch = draws;
  return (int)((draws + batch - 1) / batch);
}

// ---- virtual shards (DESIGN.md 8 "one GPU") ----------------------------------------------------
// `draws` is the epoch's total; every shard does draws / V of them in batches of `batch`, all
// shards' k-th batch in one gather + one sweep launch; the replicas are averaged every
// vs_merge_batches batches and at the end of the epoch.
bool vs_active(const sgdnet_solver* s, int64_t batch) { return s->d.V > 1 && vs_eligible(s->d, (int)batch); }

int vs_merge_batches(const sgdnet_solver* s, int64_t batch) {
  // draws per shard between merges: n / 32 of the job (parallel.py), settable for sharded jobs
  const int64_t period = s->vs_period > 0 ? s->vs_period : s->d.n / 32;
  const int64_t b = period / batch;
  return (int)(b < 1 ? 1 : b);
}

// The whole epoch in one launch (saga_batched.hip "Fused epoch"): option fused_epoch, the kernel's own limits, a
// device with at least as many CUs as the launch has workgroups, and no earlier launch of this solver that failed
// to become resident.
bool vs_fused_active(const sgdnet_solver* s) {
  if (s->fused_off || !option(kOptFusedEpoch) || !vs_fused_eligible(s->d)) return false;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess) return false;
  return s->d.V * s->d.v_bps + vs_fused_rng_workgroups(s->d) <= cus;
}

int enqueue_epoch_kernels_vs(sgdnet_solver* s, int64_t batch, int64_t draws, std::vector<hipEvent_t>* ev) {
  SagaDev& d = s->d;
  const int64_t dps = draws / d.V;
  if (batch > dps) batch = dps;
  const int nb = n_batches(batch, dps);
  const int every = vs_merge_batches(s, batch);
  if (vs_fused_active(s)) {
    s->d.vs_xcd_local = option(kOptFusedEpoch) == 1 ? 1 : 0;
    if (ev) {
      hipEvent_t e[2];
      for (auto& x : e) SGD_HIP_TRY(hipEventCreate(&x));
      int rcf = launch_vs_epoch(d, s->lam_dev, nb, every, s->st, e[0], e[1]);
      if (rcf) return rcf;
      hipEvent_t z[2];                          // no separate sweep launches: an empty interval
      for (auto& x : z) SGD_HIP_TRY(hipEventCreate(&x));
      SGD_HIP_TRY(hipEventRecord(z[0], s->st));
      SGD_HIP_TRY(hipEventRecord(z[1], s->st));
      ev->push_back(e[0]);
      ev->push_back(e[1]);
      ev->push_back(z[0]);
      ev->push_back(z[1]);
      return SGDNET_OK;
    }
    if (s->time_epochs) {
      hipEvent_t e[2];
      for (auto& x : e) SGD_HIP_TRY(hipEventCreate(&x));
      s->epoch_ev.push_back(e[0]);
      s->epoch_ev.push_back(e[1]);
      return launch_vs_epoch(d, s->lam_dev, nb, every, s->st, e[0], e[1]);
    }
    return launch_vs_epoch(d, s->lam_dev, nb, every, s->st);
  }
  int rc = launch_vs_broadcast(d, s->st);
  if (rc) return rc;
  rc = launch_vs_cw(d, s->st);
  if (rc) return rc;
  for (int k = 0; k < nb; ++k) {
    const int64_t t0 = (int64_t)k * batch;
    const int64_t m = (dps - t0 < batch) ? dps - t0 : batch;
    const int tail = (m != batch) ? 1 : 0;
    if (ev) {
      hipEvent_t e[4];
      for (auto& x : e) SGD_HIP_TRY(hipEventCreate(&x));
      rc = launch_vs_gather(d, s->lam_dev, t0, (int)m, s->st, e[0], e[1], k);
      if (rc) return rc;
      rc = launch_vs_sweep(d, s->lam_dev, tail, (int)m, s->st, e[2], e[3]);
      if (rc) return rc;
      for (auto x : e) ev->push_back(x);
    } else {
      rc = launch_vs_gather(d, s->lam_dev, t0, (int)m, s->st, nullptr, nullptr, k);
      if (rc) return rc;
      rc = launch_vs_sweep(d, s->lam_dev, tail, (int)m, s->st);
      if (rc) return rc;
    }
    const bool last = k + 1 == nb;
    if (last || (k + 1) % every == 0) {
      rc = launch_vs_merge(d, last ? 1 : 0, s->st, last ? s->lam_dev : nullptr, nb);   // the last one also ends the epoch
      if (rc) return rc;
    }
    if (!last) {
      rc = launch_vs_cw(d, s->st);                // c . w of the replicas the next gather reads
      if (rc) return rc;
    }
  }
  return SGDNET_OK;
}

// Enqueue the kernels of one batched epoch (eager or under stream capture).
int enqueue_epoch_kernels(sgdnet_solver* s, int64_t batch, int64_t draws, std::vector<hipEvent_t>* ev) {
  if (batch < 1) batch = 1;
  if (vs_active(s, batch)) return enqueue_epoch_kernels_vs(s, batch, draws, ev);
  if (batch > draws) batch = draws;
  const int nb = n_batches(batch, draws);
  if (binned_active(s->d, (int)batch)) {
    const int rcw = launch_wpad_refresh(s->d, s->st);
    if (rcw) return rcw;
  }
  for (int k = 0; k < nb; ++k) {
    const int64_t t0 = (int64_t)k * batch;
    const int64_t m = (draws - t0 < batch) ? draws - t0 : batch;
    const int tail = (m != batch) ? 1 : 0;
    // the launch geometry must fit the scratch sized by set_batch_shape (a mismatch would
    // write past d0_part / slab on the device)
    if (batch_gather_slab_doubles(s->d, (int)m) > s->slab_cap) {
      set_error("internal: gather geometry of a %lld-draw batch exceeds its scratch", (long long)m);
      return SGDNET_EINVAL;
    }
    if (ev) {
      // dispatch-level start/stop timestamps of each kernel (no host gaps inside the interval)
      hipEvent_t e[4];
      for (auto& x : e) SGD_HIP_TRY(hipEventCreate(&x));
      int rc = launch_batch_gather(s->d, s->lam_dev, t0, (int)m, tail, k, s->st, e[0], e[1]);
      if (rc) return rc;
      rc = launch_batch_sweep(s->d, s->lam_dev, s->lam.penalty, tail, (int)m, k, s->st, e[2], e[3]);
      if (rc) return rc;
      for (auto x : e) ev->push_back(x);
    } else {
      int rc = launch_batch_gather(s->d, s->lam_dev, t0, (int)m, tail, k, s->st);
      if (rc) return rc;
      rc = launch_batch_sweep(s->d, s->lam_dev, s->lam.penalty, tail, (int)m, k, s->st);
      if (rc) return rc;
    }
  }
  return launch_epoch_end(s->lam_dev, nb, s->st);
}

int ensure_graph(sgdnet_solver* s, int64_t batch, int64_t draws) {
  const int fused = (vs_active(s, batch) && vs_fused_active(s)) ? option(kOptFusedEpoch) : 0;
  for (auto& g : s->graphs)
    if (g.batch == batch && g.draws == draws && g.fused == fused) {
      s->gexec = g.exec;
      s->fused_in_graph = fused != 0;
      return SGDNET_OK;
    }
  if (s->graphs.size() >= 4) {   // a sharded epoch uses at most two shapes (segments + remainder)
    auto& old = s->graphs.front();
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    (void)hipGraphExecDestroy(old.exec);
    (void)hipGraphDestroy(old.graph);
    s->graphs.erase(s->graphs.begin());
  }
  s->fused_in_graph = fused != 0;
  SGD_HIP_TRY(hipStreamBeginCapture(s->st, hipStreamCaptureModeThreadLocal));
  int rc = enqueue_epoch_kernels(s, batch, draws, nullptr);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s->st, &g);
  if (rc) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) {
    set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    return SGDNET_EHIP;
  }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return SGDNET_EHIP;
  }
  s->graphs.push_back({batch, draws, g, ex, fused});
  s->gexec = ex;
  return SGDNET_OK;
}

// One epoch on the solver's stream: the captured graph -- or, where the epoch is ONE kernel anyway (fused epoch of the
// virtual shards), that launch itself: a one-node graph replay left ~14 us between two epochs, a plain launch ~2.
int launch_epoch(sgdnet_solver* s, int64_t batch, int64_t draws) {
  if (s->fused_in_graph) return enqueue_epoch_kernels(s, batch, draws, nullptr);
  SGD_HIP_TRY(hipGraphLaunch(s->gexec, s->st));
  return SGDNET_OK;
}

int check_batched_ok(const sgdnet_solver* s) {
  if (s->d.K > batched_max_classes()) {
    set_error("batched mode supports n_classes <= %d (got %d)", batched_max_classes(), s->d.K);
    return SGDNET_EUNSUPPORTED;
  }
  if (!s->sparse) {
    // dense x: one LDS copy of the K x p accumulator per workgroup (saga_batch_gather_dense_kernel), the tiled form when
    // that copy fits no LDS, the class-lane form for 17..64 classes
    return SGDNET_OK;
  }
  if (!s->d.rec) {
    set_error("batched mode: packed sample records were not built");
    return SGDNET_EUNSUPPORTED;
  }
  return SGDNET_OK;
}

int check_stream(const sgdnet_solver* s, int64_t off, int64_t need) {
  if (!s->stream_dev || off < 0 || off + need > s->stream_len) {
    set_error("sample stream too short: need [%lld, %lld) but %lld entries are resident",
              (long long)off, (long long)(off + need), (long long)s->stream_len);
    return SGDNET_ESTREAM;
  }
  return SGDNET_OK;
}

int read_convergence(sgdnet_solver* s, double tol, int* converged) {
  LamParams back;
  SGD_HIP_TRY(hipMemcpyAsync(&back, s->lam_dev, sizeof(LamParams), hipMemcpyDeviceToHost, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  double max_change, max_size;
  memcpy(&max_change, &back.max_change_bits, 8);
  memcpy(&max_size, &back.max_size_bits, 8);
  const bool all_zero = (max_size == 0.0) && (max_change == 0.0);
  const bool no_change = (max_size != 0.0) && (max_change / max_size <= tol);
  *converged = (all_zero || no_change) ? 1 : 0;
  s->last_change = max_change;
  s->last_size = max_size;
  s->fused_abort_seen = back.fused_abort;
  return SGDNET_OK;
}

// A fused epoch launch that gave up (saga_batched.hip "Fused epoch").  *rerun: the launch changed nothing, the
// caller runs the epoch again (the solver has switched to separate launches).
int fused_recover(sgdnet_solver* s, int code, int64_t draws, int batches, bool* rerun) {
  *rerun = false;
  if (!code) return SGDNET_OK;
  s->fused_off = true;
  s->fused_abort_seen = 0;
  SGD_HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(s->lam_dev) + offsetof(LamParams, fused_abort), 0, sizeof(int), s->st));
  SGD_HIP_TRY(hipMemsetAsync(s->d.vsync + vs_fused_sync_sticky_word(), 0, sizeof(unsigned), s->st));
  if (code != 1) {
    set_error("batched mode: a wait inside the fused epoch kernel timed out (internal error; the epoch is void)");
    return SGDNET_EHIP;
  }
  if (s->d.n_peers > 1) {                       // the separate launches know nothing of the other ranks
    set_error("batched mode: the epoch kernel of a linked solver could not become resident on its GPU (shared with other "
              "work?); the ranks' replicas are averaged inside that kernel, so there is no fallback");
    return SGDNET_EHIP;
  }
  if (getenv("SGDNET_TRACE"))
    fprintf(stderr, "[sgdnet]   the fused epoch launch could not become resident (GPU shared?): separate launches from now on\n");
  // the device did not advance the epoch's bookkeeping: take the host mirror back
  for (LamParams* q : {&s->lam, &s->lam_dev_mirror}) {
    int64_t sb = q->stream_base - draws;
    if (sb < 0 && q->stream_wrap > 0) sb += q->stream_wrap;
    q->stream_base = sb;
    q->batch_seq -= batches;
  }
  if (s->lam.stream_raw) {                      // the launch was to convert its slot of the sample order and did not
    const int64_t sb = s->lam.stream_base;
    int rc = launch_rng_convert(s->stream_dev + sb, s->pipe.n, (uint32_t)s->d.n, s->st, s->d.V, s->d.v_size, s->pipe.run_len);
    if (rc) return rc;
    s->lam.stream_raw = 0;
  }
  *rerun = true;
  return SGDNET_OK;
}

int device_convergence(sgdnet_solver* s, double tol, int* converged) {
  const size_t off = offsetof(LamParams, max_change_bits);
  SGD_HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(s->lam_dev) + off, 0, 16, s->st));
  int rc = launch_convergence(s->d, s->lam_dev, s->st);
  if (rc) return rc;
  return read_convergence(s, tol, converged);
}

int device_loss_sum(sgdnet_solver* s, double* out) {
  const size_t off = offsetof(LamParams, loss_acc);
  SGD_HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(s->lam_dev) + off, 0, 8, s->st));
  int rc = launch_loss(s->d, s->lam_dev, s->sparse, s->st);
  if (rc) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(out, reinterpret_cast<char*>(s->lam_dev) + off, 8, hipMemcpyDeviceToHost,
                             s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

int ensure_ls_table(sgdnet_solver* s, int64_t nit) {
  if (s->LS_dev && s->LS_len == nit + 1 && s->LS_alpha == s->lam.alpha && s->LS_gamma == s->lam.gamma)
    return SGDNET_OK;
  // saga-sparse.h:229-240, same sequential arithmetic
  std::vector<double> ls((size_t)(nit + 1 > 2 ? nit + 1 : 2));
  ls[0] = 0.0;
  ls[1] = 1.0;
  double geo = 1.0;
  const double upd = 1.0 - s->lam.alpha * s->lam.gamma;
  for (int64_t i = 2; i < nit + 1; ++i) {
    geo *= upd;
    ls[(size_t)i] = ls[(size_t)i - 1] + geo;
  }
  if (!s->LS_dev || s->LS_len != nit + 1) {
    if (s->LS_dev) {
      SGD_HIP_TRY(hipStreamSynchronize(s->st));
      SGD_HIP_TRY(hipFree(s->LS_dev));
      s->LS_dev = nullptr;
    }
    SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->LS_dev), sizeof(double) * ls.size()));
  }
  SGD_HIP_TRY(hipMemcpy(s->LS_dev, ls.data(), sizeof(double) * ls.size(), hipMemcpyHostToDevice));
  s->LS_len = nit + 1;
  s->LS_alpha = s->lam.alpha;
  s->LS_gamma = s->lam.gamma;
  return SGDNET_OK;
}


// Packs the sample-major CSR rows into fixed-stride records for the batched gather
// (layout: saga_batched.hip "Packed sample records").
constexpr int kOvfStride = 256;
constexpr int kOvfCap = 20;

int build_records(sgdnet_solver* s, const sgdnet_problem* pb) {
  const int64_t n = pb->n_samples;
  SagaDev& d = s->d;
  // capacity: 90th percentile of the row lengths (histogram; rows rarely exceed a few hundred)
  std::vector<int64_t> hist(65, 0);
  int64_t zmax = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t z = pb->rowptr[i + 1] - pb->rowptr[i];
    hist[(size_t)(z < 64 ? z : 64)]++;
    zmax = z > zmax ? z : zmax;
  }
  int cap = 1;
  {
    int64_t acc = 0;
    const int64_t want = (int64_t)(0.9 * (double)n);
    for (int z = 0; z <= 64; ++z) {
      acc += hist[(size_t)z];
      cap = z < 1 ? 1 : z;
      if (acc >= want) break;
    }
    if (cap >= 64) cap = (int)(zmax < 512 ? zmax : 512);
  }
  auto rec_bytes = [](int c) { return 16 + ((4 * c + 7) & ~7) + 8 * c; };
  // records are aligned to 128 B: random HBM requests on gfx950 are served in 128-B units
  // (measured: 64-B-aligned 192-B records cost 13 % more gather time than 256-B ones)
  static const int align = exp_env_int("SGDNET_REC_ALIGN", 128);
  const int stride = (rec_bytes(cap) + align - 1) / align * align;
  while (rec_bytes(cap + 1) <= stride) ++cap;
  const int val_off = 16 + ((4 * cap + 7) & ~7);
  if ((double)n * stride > 64e9) {
    set_error("packed records would need %.1f GB", (double)n * stride / 1e9);
    return SGDNET_ENOMEM;
  }
  std::vector<char> rec((size_t)n * (size_t)stride, 0);
  std::vector<char> ovf;
  const bool y_in_rec = pb->y_rows == 1;
  for (int64_t i = 0; i < n; ++i) {
    char* base = rec.data() + (size_t)i * (size_t)stride;
    const int64_t q0 = pb->rowptr[i];
    const int nnz = (int)(pb->rowptr[i + 1] - q0);
    const double y0 = y_in_rec ? pb->y[i] : 0.0;
    memcpy(base, &y0, 8);
    memcpy(base + 8, &nnz, 4);
    const int c0 = nnz < cap ? nnz : cap;
    memcpy(base + 16, pb->colidx + q0, sizeof(int32_t) * (size_t)c0);
    memcpy(base + val_off, pb->values + q0, sizeof(double) * (size_t)c0);
    int done = c0;
    int32_t* link = reinterpret_cast<int32_t*>(base + 12);   // where the next record's index goes
    size_t link_off = (size_t)(reinterpret_cast<char*>(link) - rec.data());
    bool link_in_rec = true;
    while (done < nnz) {
      const int c = (nnz - done) < kOvfCap ? (nnz - done) : kOvfCap;
      const int32_t id = (int32_t)(ovf.size() / kOvfStride);
      ovf.resize(ovf.size() + kOvfStride, 0);
      char* ob = ovf.data() + (size_t)id * kOvfStride;
      memcpy(ob + 4, &c, 4);
      memcpy(ob + 8, pb->colidx + q0 + done, sizeof(int32_t) * (size_t)c);
      memcpy(ob + 8 + 4 * kOvfCap, pb->values + q0 + done, sizeof(double) * (size_t)c);
      // patch the previous link (vectors may have been reallocated: use offsets)
      if (link_in_rec) memcpy(rec.data() + link_off, &id, 4);
      else memcpy(ovf.data() + link_off, &id, 4);
      link_off = (size_t)id * kOvfStride;
      link_in_rec = false;
      done += c;
    }
  }
  char* rec_dev = nullptr;
  char* ovf_dev = nullptr;
  int rc = dev_upload(s, &rec_dev, rec.data(), rec.size());
  if (rc) return rc;
  rc = dev_upload(s, &ovf_dev, ovf.data(), ovf.size());
  if (rc) return rc;
  SGD_HIP_TRY(hipStreamSynchronize(s->st));   // host vectors die at return
  d.rec = rec_dev;
  d.ovf = ovf_dev;
  d.rec_stride = stride;
  d.rec_cap = cap;
  d.rec_val_off = val_off;
  return SGDNET_OK;
}

}  // namespace

namespace {
struct OptionDef {
  const char* name;
  int dflt, lo, hi;
};
const OptionDef kOptionDefs[sgdnet::kOptCount] = {
    {"virtual_shards", -1, -1, 8}, {"rng_generators", 0, 0, 64},     {"window_eigenvalue", 1, 0, 1},
    {"host_setup", 0, 0, 1},       {"exact_epoch_blocks", 1, 0, 1}, {"exact_row_registers", 1, 0, 4},
    {"fused_epoch", 1, 0, 2},
};
std::atomic<int> g_options[sgdnet::kOptCount] = {{-1}, {0}, {1}, {0}, {1}, {1}, {1}};
int find_option(const char* name) {
  if (name)
    for (int i = 0; i < sgdnet::kOptCount; ++i)
      if (!strcmp(name, kOptionDefs[i].name)) return i;
  return -1;
}
}  // namespace

namespace sgdnet {
int option(Option o) { return g_options[o].load(std::memory_order_relaxed); }
}  // namespace sgdnet

extern "C" {

int sgdnet_abi_version(void) { return SGDNET_ABI_VERSION; }

int sgdnet_set_option(const char* name, int value) {
  const int i = find_option(name);
  if (i < 0 || value < kOptionDefs[i].lo || value > kOptionDefs[i].hi) {
    set_error("sgdnet_set_option: %s = %d is not an option (include/sgdnet_hip.h)", name ? name : "(null)", value);
    return SGDNET_EINVAL;
  }
  g_options[i].store(value, std::memory_order_relaxed);
  return SGDNET_OK;
}

int sgdnet_get_option(const char* name, int* value) {
  const int i = find_option(name);
  if (i < 0 || !value) {
    set_error("sgdnet_get_option: unknown option %s", name ? name : "(null)");
    return SGDNET_EINVAL;
  }
  *value = g_options[i].load(std::memory_order_relaxed);
  return SGDNET_OK;
}

const char* sgdnet_last_error(void) { return g_last_error.c_str(); }

int sgdnet_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" void sgdnet_solver_destroy(sgdnet_solver* s);

static int solver_create_impl(const sgdnet_problem* pb, DeviceSetup* adopt, sgdnet_solver** out) {
  if (!pb || !out) {
    set_error("sgdnet_solver_create: null argument");
    return SGDNET_EINVAL;
  }
  *out = nullptr;
  if (pb->n_samples <= 0 || pb->n_features <= 0 || pb->n_classes <= 0 || !pb->y || pb->y_rows <= 0 ||
      (!adopt && !pb->x_dense && !(pb->rowptr && pb->colidx && pb->values))) {
    set_error("sgdnet_solver_create: invalid problem description");
    return SGDNET_EINVAL;
  }
  if (pb->family < SGDNET_GAUSSIAN || pb->family > SGDNET_MGAUSSIAN) {
    set_error("sgdnet_solver_create: unknown family %d", pb->family);
    return SGDNET_EINVAL;
  }
  if (pb->n_samples > 0xFFFFFFFFll) {
    set_error("sgdnet_solver_create: n_samples exceeds the reference's unsigned range");
    return SGDNET_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: the SAGA backend has no CPU fallback");
    return SGDNET_ENODEVICE;
  }
  if (pb->device < 0 || pb->device >= ndev) {
    set_error("device %d out of range (%d devices)", pb->device, ndev);
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(pb->device));

  sgdnet_solver* s = new sgdnet_solver();
  s->device = pb->device;
  s->sparse = adopt ? adopt->xd_t == nullptr : pb->x_dense == nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete s;
    return SGDNET_EHIP;
  }
  SagaDev& d = s->d;
  d.family = pb->family;
  d.K = pb->n_classes;
  d.Ky = pb->y_rows;
  d.fit_intercept = pb->fit_intercept;
  d.standardize = (s->sparse && pb->standardize && (adopt || pb->x_center_scaled)) ? 1 : 0;
  d.n = pb->n_samples;
  d.p = pb->n_features;
  d.n_total = (double)(pb->n_total > 0 ? pb->n_total : pb->n_samples);
#ifdef SGDNET_EXPERIMENTS
  d.ablate = getenv("SGDNET_ABLATE") ? atoi(getenv("SGDNET_ABLATE")) : 0;
#else
  d.ablate = 0;
#endif
#ifdef SGDNET_PHASE_TIMING
  if (hipMalloc(&d.dbg, sizeof(unsigned long long) * 16 * 4096) != hipSuccess) d.dbg = nullptr;
  if (d.dbg) (void)hipMemset(d.dbg, 0, sizeof(unsigned long long) * 16 * 4096);
#endif

  const size_t n = (size_t)d.n, p = (size_t)d.p, K = (size_t)d.K;
  int rc = SGDNET_OK;
#define TRY(x)                     \
  do {                             \
    rc = (x);                      \
    if (rc) {                      \
      sgdnet_solver_destroy(s);    \
      return rc;                   \
    }                              \
  } while (0)
  if (adopt && adopt->xd_t) {
    // dense x standardised and transposed on the device (dense_setup_*): take ownership
    s->nnz = (int64_t)n * (int64_t)p;
    d.xd = adopt->xd_t;
    s->owned.push_back(adopt->xd_t);
    adopt->xd_t = nullptr;
  } else if (adopt) {
    // buffers produced on the device by setup_device.hip: take ownership
    s->nnz = adopt->nnz;
    d.avg_nnz = adopt->avg_nnz;
    d.ptr = adopt->sptr;
    d.idx = adopt->sidx;
    d.val = adopt->sval;
    d.rec = adopt->rec;
    d.ovf = adopt->ovf;
    d.rec_stride = adopt->rec_stride;
    d.rec_cap = adopt->rec_cap;
    d.rec_val_off = adopt->rec_val_off;
    for (void* q : {(void*)adopt->sptr, (void*)adopt->sidx, (void*)adopt->sval, (void*)adopt->rec,
                    (void*)adopt->ovf})
      s->owned.push_back(q);
    adopt->sptr = nullptr;
    adopt->sidx = nullptr;
    adopt->sval = nullptr;
    adopt->rec = adopt->ovf = nullptr;
    if (d.standardize) {
      d.c = adopt->center_scaled;
      s->owned.push_back(adopt->center_scaled);
      adopt->center_scaled = nullptr;
    }
  } else if (s->sparse) {
    s->nnz = pb->rowptr[n];
    d.avg_nnz = (float)((double)s->nnz / (double)n);
    int64_t* ptr;
    int32_t* idx;
    double* val;
    TRY(dev_upload(s, &ptr, pb->rowptr, n + 1));
    TRY(dev_upload(s, &idx, pb->colidx, (size_t)s->nnz));
    TRY(dev_upload(s, &val, pb->values, (size_t)s->nnz));
    d.ptr = ptr;
    d.idx = idx;
    d.val = val;
    TRY(build_records(s, pb));
  } else {
    double* xd;
    TRY(dev_upload(s, &xd, pb->x_dense, n * p));
    d.xd = xd;
  }
  {
    double* y;
    TRY(dev_upload(s, &y, pb->y, n * (size_t)d.Ky));
    d.y = y;
    if (d.standardize && !adopt) {
      double* c;
      TRY(dev_upload(s, &c, pb->x_center_scaled, p));
      d.c = c;
    }
  }
  d.y_binary = 0;
  if (d.family == SGDNET_BINOMIAL) {
    d.y_binary = 1;
    for (size_t i = 0; i < n * (size_t)d.Ky; ++i)
      if (pb->y[i] != 0.0 && pb->y[i] != 1.0) {
        d.y_binary = 0;
        break;
      }
  }
  if (s->sparse && compact_eligible(d)) {
    // an optimisation, not a requirement: without the memory for the planes the K = 1 gather
    // reads the 256-B records
    char *cP = nullptr, *cQ = nullptr;
    uint32_t* mt = nullptr;
    if (dev_alloc(s, &cP, (n + 1) * 128, false) == SGDNET_OK && dev_alloc(s, &cQ, (n + 1) * 128, false) == SGDNET_OK &&
        dev_alloc(s, &mt, (n + 15) / 16 + 1, false) == SGDNET_OK) {
      TRY(launch_pack_compact(d, cP, cQ, mt, s->st));
      d.cP = cP;
      d.cQ = cQ;
      d.cmeta = mt;
      d.cE = compact_entries(d);
    } else {
      (void)hipGetLastError();
    }
  }
  TRY(dev_alloc(s, &d.w, K * p, true));
  TRY(dev_alloc(s, &d.G, K * p, true));
  TRY(dev_alloc(s, &d.M, K * n, true));
  d.m_base = reinterpret_cast<char*>(d.M);
  d.m_stride = 8;
  TRY(dev_alloc(s, &d.b, K, true));
  TRY(dev_alloc(s, &d.gb, K, true));
  TRY(dev_alloc(s, &d.w_prev, K * p, true));
  TRY(dev_alloc(s, &d.lag, p, true));
  TRY(dev_alloc(s, &d.D, K * p, true));
  TRY(dev_alloc(s, &d.claim, n, false));
  TRY(dev_alloc(s, &d.d0_part, 2 * 256 * K, true));   // two parity sets of 256 slots (saga_batched.hip)
  TRY(dev_alloc(s, &d.cw, 2 * 16 * K, true));
  TRY(dev_alloc(s, &s->ref, 2 * K * p + 2 * K, true));
  TRY(dev_alloc(s, &s->out_dev, 4, true));
  TRY(dev_alloc(s, &s->lam_dev, 1, true));
#undef TRY
  e = hipHostMalloc(reinterpret_cast<void**>(&s->lam_stage), sizeof(LamParams) * sgdnet_solver::kLamSlots,
                    hipHostMallocDefault);
  for (int i = 0; e == hipSuccess && i < sgdnet_solver::kLamSlots; ++i)
    e = hipEventCreateWithFlags(&s->lam_ev[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("solver initialisation failed: %s", hipGetErrorString(e));
    sgdnet_solver_destroy(s);
    return SGDNET_EHIP;
  }
  e = hipMemsetAsync(d.claim, 0xFF, sizeof(int) * n, s->st);  // -1: never a batch id
  if (e == hipSuccess) e = hipStreamSynchronize(s->st);
  if (e != hipSuccess) {
    set_error("solver initialisation failed: %s", hipGetErrorString(e));
    sgdnet_solver_destroy(s);
    return SGDNET_EHIP;
  }
  memset(&s->lam, 0, sizeof(s->lam));
  *out = s;
  return SGDNET_OK;
}

int sgdnet_solver_create(const sgdnet_problem* pb, sgdnet_solver** out) {
  return solver_create_impl(pb, nullptr, out);
}

void sgdnet_solver_destroy(sgdnet_solver* s) {
  if (!s) return;
  if (getenv("SGDNET_TRACE") && g_trace_epochs) {
    fprintf(stderr, "[sgdnet]   batched epochs %ld: graph launch calls %.3f s, waiting for the epoch + ConvergenceCheck %.3f s, epoch graphs on the device %.3f s\n",
            g_trace_epochs, g_trace_launch, g_trace_conv, g_trace_graph);
    g_trace_graph = 0.0;
    g_trace_epochs = 0;
    g_trace_launch = g_trace_conv = 0.0;
  }
  (void)hipSetDevice(s->device);
  if (s->st) (void)hipStreamSynchronize(s->st);
  drop_graph(s);
  for (void* q : s->ipc_opened) (void)hipIpcCloseMemHandle(q);
  for (hipEvent_t e : s->epoch_ev) (void)hipEventDestroy(e);
  for (void* p : s->owned) (void)hipFree(p);
  if (s->LS_dev) (void)hipFree(s->LS_dev);
  if (s->d.slab) (void)hipFree(s->d.slab);
  for (void* q : s->bin_bufs)
    if (q) (void)hipFree(q);
  for (void* q : s->vs_owned) (void)hipFree(q);
  if (s->lam_stage) (void)hipHostFree(s->lam_stage);
  for (hipEvent_t ev : s->lam_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (s->stream_dev) (void)hipFree(s->stream_dev);
  if (s->rng_dev) (void)hipFree(s->rng_dev);
  if (s->pipe.st) {
    (void)hipStreamSynchronize(s->pipe.st);
    (void)hipStreamDestroy(s->pipe.st);
  }
  for (int i = 0; i < 2; ++i) {
    if (s->pipe.ready[i]) (void)hipEventDestroy(s->pipe.ready[i]);
    if (s->pipe.freed[i]) (void)hipEventDestroy(s->pipe.freed[i]);
    if (s->pipe.state[i]) (void)hipFree(s->pipe.state[i]);
  }
  if (s->pipe.poly_n) (void)hipFree(s->pipe.poly_n);
  if (s->pipe.dev) (void)hipFree(s->pipe.dev);
  if (s->pipe.ends) (void)hipFree(s->pipe.ends);
  if (s->st) (void)hipStreamDestroy(s->st);
  delete s;
}

int sgdnet_solver_set_penalty(sgdnet_solver* s, int penalty, double gamma, double alpha, double beta) {
  if (!s || penalty < SGDNET_RIDGE || penalty > SGDNET_GROUPLASSO || !(gamma > 0.0)) {
    set_error("sgdnet_solver_set_penalty: invalid argument");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  if (penalty != s->lam.penalty) drop_graph(s);   // the captured sweep kernel depends on it
  s->lam.penalty = penalty;
  s->lam.gamma = gamma;
  s->lam.alpha = alpha;
  s->lam.beta = beta;
  s->penalty_set = true;
  return push_lam(s);
}

static int state_ptr(sgdnet_solver* s, int which, double** p, size_t* count) {
  const size_t K = (size_t)s->d.K, n = (size_t)s->d.n, pp = (size_t)s->d.p;
  switch (which) {
    case 0: *p = s->d.w; *count = K * pp; break;
    case 1: *p = s->d.b; *count = K; break;
    case 2: *p = s->d.M; *count = K * n; break;
    case 3: *p = s->d.G; *count = K * pp; break;
    case 4: *p = s->d.gb; *count = K; break;
    default:
      set_error("unknown state array %d", which);
      return SGDNET_EINVAL;
  }
  return SGDNET_OK;
}

int sgdnet_solver_get_state(sgdnet_solver* s, int which, double* host) {
  if (!s || !host) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  double* p;
  size_t count;
  int rc = state_ptr(s, which, &p, &count);
  if (rc) return rc;
  if (which == 2) rc = m_to_array(s);
  if (rc) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(host, p, sizeof(double) * count, hipMemcpyDeviceToHost, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

int sgdnet_solver_set_state(sgdnet_solver* s, int which, const double* host) {
  if (!s || !host) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  double* p;
  size_t count;
  int rc = state_ptr(s, which, &p, &count);
  if (rc) return rc;
  if (which == 2) rc = m_to_array(s);
  if (rc) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(p, host, sizeof(double) * count, hipMemcpyHostToDevice, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  if (which == 0) s->w_prev_valid = false;
  return SGDNET_OK;
}

static int reserve_stream(sgdnet_solver* s, int64_t count) {
  if (count > s->stream_cap || !s->stream_dev) {
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    if (s->stream_dev) SGD_HIP_TRY(hipFree(s->stream_dev));
    s->stream_dev = nullptr;
    SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->stream_dev), sizeof(uint32_t) * (size_t)count));
    s->stream_cap = count;
    drop_graph(s);  // captured kernels hold the old pointer
  }
  s->stream_len = count;
  s->d.stream = s->stream_dev;
  return SGDNET_OK;
}

int sgdnet_solver_upload_stream(sgdnet_solver* s, const uint32_t* host, int64_t count) {
  if (!s || !host || count <= 0) {
    set_error("sgdnet_solver_upload_stream: invalid argument");
    return SGDNET_EINVAL;
  }
  // the kernels use the entries as addresses (ptr[s + 1], g_memory[s * K]): reject anything
  // that is not a sample index before it reaches the device
  uint32_t top = 0;
  for (int64_t i = 0; i < count; ++i) top = host[i] > top ? host[i] : top;
  if ((int64_t)top >= s->d.n) {
    set_error("sgdnet_solver_upload_stream: entry %u is not a sample index (n_samples = %lld)", top,
              (long long)s->d.n);
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = reserve_stream(s, count);
  if (rc) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(s->stream_dev, host, sizeof(uint32_t) * (size_t)count, hipMemcpyHostToDevice,
                             s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

// ---- sample-order pipeline in the C ABI (what sgdnet_fit_* uses internally) ----
int sgdnet_solver_rng_open(sgdnet_solver* s, sgdnet_rng* rng, int64_t draws_per_epoch, int generators) {
  if (!s || !rng || draws_per_epoch <= 0) {
    set_error("sgdnet_solver_rng_open: invalid argument");
    return SGDNET_EINVAL;
  }
  int rc = solver_rng_open(s, rng, draws_per_epoch, generators);
  if (rc) return rc;
  return solver_rng_prefetch(s);
}

int sgdnet_solver_rng_layout(sgdnet_solver* s, int64_t draws_per_run) {
  if (!s || draws_per_run < 0) return SGDNET_EINVAL;
  if (s->pipe.open) {
    set_error("sgdnet_solver_rng_layout: set the layout before sgdnet_solver_rng_open");
    return SGDNET_EINVAL;
  }
  s->pipe.run_len = draws_per_run;
  return SGDNET_OK;
}

int sgdnet_solver_rng_next(sgdnet_solver* s, int64_t* stream_offset) {
  if (!s || !stream_offset) return SGDNET_EINVAL;
  int rc = solver_rng_prefetch(s);               // the epoch after this one, concurrently
  if (rc) return rc;
  return solver_rng_acquire(s, stream_offset);
}

int sgdnet_solver_rng_done(sgdnet_solver* s) { return s ? solver_rng_release(s) : SGDNET_EINVAL; }

int sgdnet_solver_rng_close(sgdnet_solver* s, sgdnet_rng* rng) {
  if (!s || !rng) return SGDNET_EINVAL;
  return solver_rng_close(s, rng);
}

int sgdnet_solver_get_stream(sgdnet_solver* s, uint32_t* host, int64_t offset, int64_t count) {
  if (!s || !host || offset < 0 || count <= 0 || offset + count > s->stream_len) {
    set_error("sgdnet_solver_get_stream: range outside the resident stream");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipMemcpyAsync(host, s->stream_dev + offset, sizeof(uint32_t) * (size_t)count,
                             hipMemcpyDeviceToHost, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

int sgdnet_solver_generate_stream(sgdnet_solver* s, sgdnet_rng* rng, int64_t count) {
  if (!s || !rng || count <= 0) {
    set_error("sgdnet_solver_generate_stream: invalid argument");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = reserve_stream(s, count);
  if (rc) return rc;
  if (!s->rng_dev) SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->rng_dev), sizeof(sgdnet_rng)));
  static_assert(sizeof(sgdnet_rng) == 625 * sizeof(uint32_t), "sgdnet_rng is mti + 624 words");
  SGD_HIP_TRY(hipMemcpyAsync(s->rng_dev, rng, sizeof(sgdnet_rng), hipMemcpyHostToDevice, s->st));
  rc = launch_rng_fill(s->rng_dev, s->rng_dev, (uint32_t)s->d.n, s->stream_dev, count, s->st, s->d.V, s->d.v_size);
  if (rc) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(rng, s->rng_dev, sizeof(sgdnet_rng), hipMemcpyDeviceToHost, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

}  // extern "C"

// w = g_sum = g_memory = g_sum_intercept = 0, intercept = b0 (K values): the state a fit starts
// from (driver.cpp restarts a lambda from here when the automatic staleness window diverged)
// Binned form, recovery from a bin overflow (driver.cpp): true when the last synchronisation found one.
bool solver_fused_aborted(const sgdnet_solver* s) { return s && s->fused_abort_seen != 0; }
bool solver_bin_overflowed(const sgdnet_solver* s) { return s && s->bin_overflowed; }

// After an overflow: twice the room in every bin (rebuilt by the next run); after three doublings the
// solver gives the binned form up -- the atomic form takes over where there is one (n_classes <= 16),
// otherwise SGDNET_EUNSUPPORTED (no batched form left: the caller falls back to the exact iteration).
int solver_grow_bins(sgdnet_solver* s) {
  s->bin_overflowed = false;
  if (s->bin_slack < 64.0) {
    s->bin_slack *= 2.0;
    return SGDNET_OK;
  }
  s->bin_disabled = true;
  if (s->d.K > 16) {
    set_error("batched mode: the binned form keeps overflowing and more than 16 classes have no other batched form");
    return SGDNET_EUNSUPPORTED;
  }
  return SGDNET_OK;
}

// Is there a batched form for this solver at this window?  (More than 16 classes need the binned form,
// which needs feature ranges: at most 2048 of them, sparse x.)
bool solver_batched_available(sgdnet_solver* s, int64_t batch) {
  if (!s) return false;
  if (s->d.K <= 16) return true;
  if (s->d.K > 64) return false;
  if (!s->sparse) return true;                  // dense x: the class-lane form
  if (ensure_binned(s, batch < 1 ? 1 : batch) != SGDNET_OK) return false;
  return s->d.R > 0 && !s->bin_disabled;
}

int solver_reset_state(sgdnet_solver* s, const double* b0) {
  SGD_HIP_TRY(hipSetDevice(s->device));
  if (s->m_in_rec) {              // the array is cleared below; the records take it over again at the next batched run
    s->m_in_rec = false;
    s->d.m_rec = 0;
    s->d.m_base = reinterpret_cast<char*>(s->d.M);
    s->d.m_stride = 8;
    drop_graph(s);
  }
  const SagaDev& d = s->d;
  const size_t K = (size_t)d.K;
  SGD_HIP_TRY(hipMemsetAsync(d.w, 0, sizeof(double) * K * (size_t)d.p, s->st));
  SGD_HIP_TRY(hipMemsetAsync(d.G, 0, sizeof(double) * K * (size_t)d.p, s->st));
  SGD_HIP_TRY(hipMemsetAsync(d.M, 0, sizeof(double) * K * (size_t)d.n, s->st));
  SGD_HIP_TRY(hipMemsetAsync(d.gb, 0, sizeof(double) * K, s->st));
  SGD_HIP_TRY(hipMemcpyAsync(d.b, b0, sizeof(double) * K, hipMemcpyHostToDevice, s->st));
  SGD_HIP_TRY(hipMemsetAsync(d.lag, 0, sizeof(unsigned) * (size_t)d.p, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  s->w_prev_valid = false;
  return SGDNET_OK;
}

// ---- sample-order pipeline (driver.cpp) -------------------------------------------------------
// The stream buffer holds two epochs; epoch e reads half e & 1 while the side stream fills the
// other half with the draws of epoch e + 1.  The generator state ping-pongs between two device
// buffers, so the state after exactly `used` epochs survives one speculative generation.
// generators > 1 (batched mode, where the trajectory is not the reference's anyway): the epoch's
// stream is cut into that many consecutive segments, each filled by its own MT19937 -- the
// caller's generator for segment 0, and for segment g a generator seeded (set.seed scrambling,
// r_rng.cpp) with floor(2^32 * unif_rand()) drawn from the caller's generator at this point.
// One generator makes 10M draws in 5.3 ms, which is six epochs of the batched kernels at C4.
// jump_draws: draws of the WHOLE job per epoch when this solver holds one rank's range of a stream shared by several
// (driver.cpp, control.n_gpus); 0: n
int solver_rng_open(sgdnet_solver* s, sgdnet_rng* rng, int64_t n, int generators, int64_t jump_draws) {
  SGD_HIP_TRY(hipSetDevice(s->device));
  auto& P = s->pipe;
  if (generators < 1) generators = 1;
  if (generators > sgdnet_solver::RngPipe::kMaxGen) generators = sgdnet_solver::RngPipe::kMaxGen;
  int rc = reserve_stream(s, 2 * n);
  if (rc) return rc;
  if (!P.st) {
    SGD_HIP_TRY(hipStreamCreateWithFlags(&P.st, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      SGD_HIP_TRY(hipEventCreateWithFlags(&P.ready[i], hipEventDisableTiming));
      SGD_HIP_TRY(hipEventCreateWithFlags(&P.freed[i], hipEventDisableTiming));
      SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&P.state[i]),
                            sizeof(sgdnet_rng) * sgdnet_solver::RngPipe::kMaxGen));
    }
  }
  // Several generators work on ONE stream -- R's, as set.seed() left it: generator g starts
  // g * seg draws into the epoch (its state = the first one jumped g * seg draws ahead), and every
  // epoch all starts move n draws on (mt_jump.cpp).  If the jump polynomials cannot be had the fit
  // keeps a single generator.
  const int64_t seg = (n + generators - 1) / generators;
  std::vector<uint32_t> poly_seg(624), poly_n(624);
  if (generators > 1 && !(mt_jump_poly((uint64_t)seg, poly_seg.data()) &&
                          mt_jump_poly((uint64_t)(jump_draws > 0 ? jump_draws : n), poly_n.data())))
    generators = 1;
  P.G = generators;
  {
    // the generators' workgroups get CUs of their own: the LDS gather forms shrink their grids
    // one workgroup per generator up to 8 workgroups, then up to four generators per workgroup: the generators' work
    // per epoch (state step: seg / 624 blocks of ~0.46 us; jump: ~64 us per generator and workgroup) must stay below the
    // epoch's own duration -- at C3 (1M draws, 8 generators) two workgroups of four needed 370 us beside a 246-us epoch
    const int per_wg = rng_generators_per_workgroup();
    const int reserve = generators > 1 ? std::max(std::min(generators, 8), (generators + per_wg - 1) / per_wg) : 0;
    // (Tried: confining the side stream to exactly those CUs with hipExtStreamCreateWithCUMask -- mask bit i is a
    // CU of XCC i % 8, scripts/microbench/cu_mask.hip -- so that the conversion kernel's 2048 small workgroups
    // cannot spread over CUs a gather launch is about to need: on 8 CUs that kernel takes 0.8 ms instead of 0.05,
    // the side stream becomes the epoch's critical path (1.32 ms per epoch) and the gather beside it is slower,
    // not faster (90 us per launch).  A masked stream for the solver itself places N whole-LDS workgroups on
    // N - 1 of its N CUs, i.e. runs two rounds.  profiles/r03p_cu_mask_microbench.txt, r03q_*.)
    if (reserve != s->d.cu_reserve) {
      SGD_HIP_TRY(hipStreamSynchronize(s->st));
      s->d.cu_reserve = reserve;
      if (s->d.V > 1) s->d.v_bps = lds_target_grid(s->d) / s->d.V;
      drop_graph(s);
    }
  }
  SGD_HIP_TRY(hipMemcpy(P.state[0], rng, sizeof(sgdnet_rng), hipMemcpyHostToDevice));
  if (generators > 1) {
    if (!P.poly_n) {
      SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&P.poly_n), sizeof(uint32_t) * 624 * 2));
      SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&P.ends), sizeof(sgdnet_rng) * sgdnet_solver::RngPipe::kMaxGen));
    }
    SGD_HIP_TRY(hipMemcpy(P.poly_n, poly_n.data(), sizeof(uint32_t) * 624, hipMemcpyHostToDevice));
    SGD_HIP_TRY(hipMemcpy(P.poly_n + 624, poly_seg.data(), sizeof(uint32_t) * 624, hipMemcpyHostToDevice));
    for (int g = 1; g < generators; ++g) {      // start[g] = start[g - 1] jumped seg draws: once per fit
      int rcj = launch_rng_jump(P.state[0] + (size_t)(g - 1) * 625, P.state[0] + (size_t)g * 625, P.poly_n + 624, 1, P.st);
      if (rcj) return rcj;
    }
    SGD_HIP_TRY(hipStreamSynchronize(P.st));
  }
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  for (int i = 0; i < 2; ++i) SGD_HIP_TRY(hipEventRecord(P.freed[i], s->st));
  P.n = n;
  P.gens = P.used = 0;
  P.pending_gen = -1;
  P.raw[0] = P.raw[1] = false;
  // the same generators as the fused epoch kernel of the virtual shards runs them on its spare workgroups
  RngDev* want = nullptr;
  if (generators > 1 && s->d.V > 1 && s->d.vsync) {
    if (!P.dev) SGD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&P.dev), sizeof(RngDev)));
    RngDev h{};
    h.state[0] = P.state[0];
    h.state[1] = P.state[1];
    h.ends = P.ends;
    h.stream = s->stream_dev;
    h.poly = P.poly_n;
    h.n = n;
    h.seg = seg;
    h.gens = generators;
    h.gen = 0u;
    SGD_HIP_TRY(hipMemcpy(P.dev, &h, sizeof(RngDev), hipMemcpyHostToDevice));
    want = P.dev;
  }
  if (want != s->d.rngdev) {
    s->d.rngdev = want;
    drop_graph(s);
  }
  P.open = true;
  return SGDNET_OK;
}

// generation g of the sample order on the side stream: slot g & 1, start states state[g & 1] -> state[(g + 1) & 1]
static int rng_side_generate(sgdnet_solver* s, int64_t g, bool keep_raw) {
  auto& P = s->pipe;
  const int slot = (int)(g & 1);
  SGD_HIP_TRY(hipStreamWaitEvent(P.st, P.freed[slot], 0));
  int rc;
  P.raw[slot] = keep_raw;
  if (P.G > 1) {
    rc = launch_rng_fill(P.state[g & 1], P.ends, (uint32_t)s->d.n, s->stream_dev + (int64_t)slot * P.n, P.n,
                         P.st, s->d.V, s->d.v_size, P.G, P.run_len, keep_raw ? 0 : 1, 0, s->d.cu_reserve);
    // the jump's workgroups take their generators in turn: the side stream never holds more CUs than
    // the generators' own (a wider launch would push gather workgroups into a second round)
    if (!rc) rc = launch_rng_jump(P.state[g & 1], P.state[(g + 1) & 1], P.poly_n, P.G, P.st,
                                  std::max(1, s->d.cu_reserve));
  } else {
    rc = launch_rng_fill(P.state[g & 1], P.state[(g + 1) & 1], (uint32_t)s->d.n,
                         s->stream_dev + (int64_t)slot * P.n, P.n, P.st, s->d.V, s->d.v_size, P.G, P.run_len,
                         keep_raw ? 0 : 1);
  }
  if (rc) return rc;
  if (P.dev)                                    // the in-kernel generators continue from here
    SGD_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&P.dev->gen), (int)(g + 1), 1, P.st));
  SGD_HIP_TRY(hipEventRecord(P.ready[slot], P.st));
  return SGDNET_OK;
}

// enqueue the next generation (never more than one ahead of the epoch being consumed)
int solver_rng_prefetch(sgdnet_solver* s) {
  auto& P = s->pipe;
  if (!P.open || P.gens > P.used + 1) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  // Virtual shards with the fused epoch kernel: that kernel holds every CU for a whole epoch, so its own spare
  // workgroups produce the next generation (raw words: every workgroup of the NEXT epoch's launch converts the share
  // it reads), and nothing is launched here: the generation is pending until the launch that carries it is enqueued
  // (prepare_stream_slot), or until its draws are asked for without such a launch (solver_rng_acquire).
  // (Generators launched beside the epoch kernel raced it for CUs: dispatched together, one epoch workgroup per XCD
  //  found its CU taken and the whole epoch waited for the generators, +215 us; dispatched later, their own
  //  workgroups could stall until the epoch ended -- profiles/r04_rng_placement.txt.)
  const bool keep_raw = s->d.V > 1 && P.run_len == 0 && vs_fused_active(s);
  if (keep_raw && P.dev && s->d.rngdev && P.G > 1 && P.gens >= 1 && P.pending_gen < 0) {
    P.pending_gen = P.gens;
    P.raw[P.gens & 1] = true;
    ++P.gens;
    return SGDNET_OK;
  }
  int rc = rng_side_generate(s, P.gens, keep_raw);
  if (rc) return rc;
  ++P.gens;
  return SGDNET_OK;
}

// the solver's stream waits for the draws of the next unconsumed epoch; *offset = where they are
int solver_rng_acquire(sgdnet_solver* s, int64_t* offset) {
  auto& P = s->pipe;
  if (!P.open || P.gens <= P.used) return SGDNET_EINVAL;
  const int slot = (int)(P.used & 1);
  if (P.pending_gen == P.used) {                // no fused launch carried this generation: the side stream makes it now
    const int64_t g = P.pending_gen;
    P.pending_gen = -1;
    SGD_HIP_TRY(hipEventRecord(P.freed[slot], s->st));       // after everything enqueued so far
    int rc = rng_side_generate(s, g, P.raw[slot]);
    if (rc) return rc;
  }
  SGD_HIP_TRY(hipStreamWaitEvent(s->st, P.ready[slot], 0));
  *offset = (int64_t)slot * P.n;
  return SGDNET_OK;
}

// the epoch that consumed the acquired draws has been enqueued on the solver's stream
int solver_rng_release(sgdnet_solver* s) {
  auto& P = s->pipe;
  SGD_HIP_TRY(hipEventRecord(P.freed[P.used & 1], s->st));
  ++P.used;
  return SGDNET_OK;
}

// generator state after exactly `used` epochs (a speculative generation is discarded)
int solver_rng_close(sgdnet_solver* s, sgdnet_rng* rng) {
  auto& P = s->pipe;
  if (!P.open) return SGDNET_OK;
  SGD_HIP_TRY(hipSetDevice(s->device));
  if (P.pending_gen >= 0) {                     // a generation nobody produced: it does not exist (the state below is
    P.raw[P.pending_gen & 1] = false;           // the one after `used` epochs either way)
    --P.gens;
    P.pending_gen = -1;
  }
  SGD_HIP_TRY(hipStreamSynchronize(P.st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  SGD_HIP_TRY(hipMemcpy(rng, P.state[P.used & 1], sizeof(sgdnet_rng), hipMemcpyDeviceToHost));
  for (int q = 0; q < 2; ++q) {                 // a slot generated ahead for the fused epoch kernel and never consumed
    if (!P.raw[q]) continue;
    int rcq = launch_rng_convert(s->stream_dev + (int64_t)q * P.n, P.n, (uint32_t)s->d.n, s->st, s->d.V, s->d.v_size, P.run_len);
    if (rcq) return rcq;
    P.raw[q] = false;
  }
  P.open = false;
  if (s->d.rngdev) {
    s->d.rngdev = nullptr;
    drop_graph(s);
  }
  if (s->d.cu_reserve) {
    s->d.cu_reserve = 0;
    if (s->d.V > 1) s->d.v_bps = lds_target_grid(s->d) / s->d.V;
    drop_graph(s);
  }
  return SGDNET_OK;
}

extern "C" {

static int check_bins(sgdnet_solver* s);

int sgdnet_solver_run(sgdnet_solver* s, int mode, int64_t batch, int64_t stream_offset,
                      int64_t draws_per_epoch, unsigned max_epochs, double tol, unsigned* epochs_run,
                      int* converged_out, double* losses) {
  if (!s || draws_per_epoch <= 0 || max_epochs == 0 || !epochs_run || !converged_out) {
    set_error("sgdnet_solver_run: invalid argument");
    return SGDNET_EINVAL;
  }
  if (!s->penalty_set) {
    set_error("sgdnet_solver_run: call sgdnet_solver_set_penalty first");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = check_stream(s, stream_offset, draws_per_epoch);
  if (rc) return rc;
  const size_t wbytes = sizeof(double) * (size_t)s->d.K * (size_t)s->d.p;
  unsigned done = 0;
  int converged = 0;

  if (mode == SGDNET_MODE_EXACT) {
    rc = m_to_array(s);                       // the exact kernels read the K x n gradient memory
    if (rc) return rc;
    if (s->sparse) {
      rc = ensure_ls_table(s, draws_per_epoch);
      if (rc) return rc;
    }
    const size_t lds_cap = 160 * 1024 - 256;
    // small dense problems: the register-resident kernel (saga_exact.hip); SGDNET_EXACT_SMALL=0 keeps the general one
    static const int small_ok = exp_env_int("SGDNET_EXACT_SMALL", 1);
    const size_t lds_small = (!s->sparse && small_ok) ? dense_exact_small_lds_bytes(s->d, draws_per_epoch) : 0;
    // ... with a feeder wavefront where there is one response (SGDNET_EXACT_SMALL2=0 in experiment builds: without)
    static const int small2_ok = exp_env_int("SGDNET_EXACT_SMALL2", 1);
    const size_t lds_small2 = (lds_small && small2_ok) ? dense_exact_small2_lds_bytes(s->d, s->lam.penalty, draws_per_epoch) : 0;
    // wider dense rows (up to 16 classes): the workgroup-per-iteration kernel; SGDNET_EXACT_WIDE=0 keeps the one-wavefront one
    static const int wide_ok = exp_env_int("SGDNET_EXACT_WIDE", 1);
    const bool wide = !s->sparse && !lds_small && wide_ok && dense_exact_wide_threads(s->d) > 0;
    // sparse x, one response, no implicit centring: the register-resident iteration (option exact_row_registers:
    // 0 keeps the general kernel; 2 keeps w, g_sum and lag out of the LDS even where they fit; 4: as 1 without the
    // multi-consumer kernel)
    const int k1_opt = option(kOptExactRowRegisters);
    const bool k1 = s->sparse && k1_opt != 0 && sparse_exact_k1_eligible(s->d);
    int k1_cache = 0, k1_stage = 0;
    size_t k1_lds = k1 ? sparse_exact_k1_lds_bytes(s->d, draws_per_epoch, k1_opt == 1 || k1_opt == 4, &k1_cache, &k1_stage) : 0;
    // ... and with several consumer wavefronts where draws seldom share a feature (1: where the rule below expects it
    // to pay; 3: wherever it is legal; 2, 4: never).
    bool k1m = false;
    if (k1 && (k1_opt == 1 || k1_opt == 3)) {
      const double upd = 1.0 - s->lam.alpha * s->lam.gamma;
      const bool odd_scale = !(upd > 0.0);        // w_scale changing sign: the one-consumer kernel's plain soft threshold
      const double avg = (double)s->nnz / (double)s->d.n;
      const double share = (double)(sparse_exact_k1m_consumers() - 1) * avg * avg / (double)s->d.p;   // P(a draw in flight shares a feature)
      k1m = !odd_scale && draws_per_epoch >= 64 && (k1_opt == 3 || share < 6.0);   // (measured: 1.2 against 1.4 us at share 2.5, 1.36 against 1.40 at 4.5)
      if (k1m) {
        k1_lds = sparse_exact_k1m_lds_bytes(draws_per_epoch, &k1_cache);
        k1_stage = 0;
      }
    }
    // several classes (or one response where the kernels above do not apply): the general iteration on several
    // wavefronts at once, same options (1: where draws seldom share features, 3: wherever legal)
    bool mc = false;
    if (s->sparse && !k1m && !k1 && (k1_opt == 1 || k1_opt == 3) && sparse_exact_mc_eligible(s->d)) {
      const double upd = 1.0 - s->lam.alpha * s->lam.gamma;
      const double avg = (double)s->nnz / (double)s->d.n;
      const double share = (double)(sparse_exact_mc_wavefronts() - 1) * avg * avg / (double)s->d.p;
      mc = upd > 0.0 && draws_per_epoch >= 64 && (k1_opt == 3 || share < 1.0);   // (measured: 4.04 against 3.90 us at share 3.5)
    }
    const size_t lds_full = s->sparse ? sparse_exact_lds_bytes(s->d, true)
                                      : (wide ? dense_exact_wide_lds_bytes(s->d, true) : dense_exact_lds_bytes(s->d, true));
    const bool stage = lds_full <= lds_cap;
    const size_t lds = stage ? lds_full
                             : (s->sparse ? sparse_exact_lds_bytes(s->d, false)
                                          : (wide ? dense_exact_wide_lds_bytes(s->d, false) : dense_exact_lds_bytes(s->d, false)));
    if (lds > lds_cap) {
      set_error("exact mode: per-iteration scratch (%zu bytes) exceeds LDS", lds);
      return SGDNET_EUNSUPPORTED;
    }
    while (done < max_epochs && !converged) {
      const int64_t avail = (s->stream_len - stream_offset) / draws_per_epoch;
      if (avail <= 0) {
        set_error("sample stream exhausted after %u epochs", done);
        return SGDNET_ESTREAM;
      }
      unsigned chunk = max_epochs - done;
      if ((int64_t)chunk > avail) chunk = (unsigned)avail;
      if (losses) chunk = 1;  // per-epoch loss needs a launch boundary
      ExactCtl ctl{};
      ctl.stream_off = stream_offset;
      ctl.nit = draws_per_epoch;
      ctl.max_epochs = chunk;
      ctl.tol = tol;
      ctl.LS = s->LS_dev;
      ctl.use_lds = stage ? 1 : 0;
      ctl.out = s->out_dev;
      if (k1) {
        ctl.use_lds = k1_stage;
        ctl.ls_cache = k1_cache;
      }
      rc = mc ? launch_sparse_exact_mc(s->d, s->lam_dev, ctl, s->st)
         : k1m ? launch_sparse_exact_k1m(s->d, s->lam_dev, ctl, k1_lds, s->st)
         : k1 ? launch_sparse_exact_k1(s->d, s->lam_dev, ctl, k1_lds, s->st)
         : s->sparse ? launch_sparse_exact(s->d, s->lam_dev, ctl, lds, s->st)
                     : (lds_small2 ? launch_dense_exact_small2(s->d, s->lam.penalty, s->lam_dev, ctl, lds_small2, s->st)
                        : lds_small ? launch_dense_exact_small(s->d, s->lam.penalty, s->lam_dev, ctl, lds_small, s->st)
                        : wide    ? launch_dense_exact_wide(s->d, s->lam_dev, ctl, lds, s->st)
                                  : launch_dense_exact(s->d, s->lam_dev, ctl, lds, s->st));
      if (rc) return rc;
      int out[2] = {0, 0};
      SGD_HIP_TRY(hipMemcpyAsync(out, s->out_dev, sizeof(out), hipMemcpyDeviceToHost, s->st));
      SGD_HIP_TRY(hipStreamSynchronize(s->st));
      if (out[1] < 0) {
        set_error("exact mode: the sample-order producer of the sparse kernel stalled (internal error)");
        return SGDNET_EHIP;
      }
      if (losses) {
        double sum = 0.0;
        rc = device_loss_sum(s, &sum);
        if (rc) return rc;
        losses[done] = sum / (double)s->d.n;
      }
#ifdef SGDNET_PHASE_TIMING
      if (lds_small2 && !s->sparse && s->d.dbg && out[0] > 0) {
        unsigned long long c[5];
        SGD_HIP_TRY(hipMemcpy(c, s->d.dbg + 24, sizeof(c), hipMemcpyDeviceToHost));
        (void)hipMemset(s->d.dbg + 24, 0, sizeof(c));
        const double its = (double)out[0] * (double)draws_per_epoch;
        fprintf(stderr, "[sgdnet] small dense kernel with feeder, consumer cycles per draw: slot+history %.0f, dot %.0f, gradient+store %.0f, "
                        "intercept %.0f, step+penalty+average %.0f\n", c[0] / its, c[1] / its, c[2] / its, c[3] / its, c[4] / its);
      }
      if (k1m && s->d.dbg && out[0] > 0) {
        (void)hipDeviceSynchronize();
        unsigned long long c[5];
        SGD_HIP_TRY(hipMemcpy(c, s->d.dbg + 16, sizeof(c), hipMemcpyDeviceToHost));
        (void)hipMemset(s->d.dbg + 16, 0, sizeof(c));
        const double its = (double)out[0] * (double)draws_per_epoch;
        fprintf(stderr, "[sgdnet] multi-consumer sparse kernel, polls per draw: slot %.2f, registration %.2f, dependency %.2f, chain %.2f, barrier %.3f\n",
                c[0] / its, c[1] / its, c[2] / its, c[3] / its, c[4] / its);
      } else if (k1 && s->d.dbg && out[0] > 0) {
        (void)hipDeviceSynchronize();
        unsigned long long c[12];
        SGD_HIP_TRY(hipMemcpy(c, s->d.dbg, sizeof(c), hipMemcpyDeviceToHost));
        (void)hipMemset(s->d.dbg, 0, sizeof(c));
        fprintf(stderr, "[sgdnet] producer/consumer sparse kernel, %d x %lld draws: producer waited %llu times (%llu polls), consumer %llu times (%llu polls)\n",
                out[0], (long long)draws_per_epoch, c[9], c[8], c[11], c[10]);
        const double its = (double)out[0] * (double)draws_per_epoch;
        fprintf(stderr, "[sgdnet]   consumer cycles per draw: slot+requests %.0f, catch-up+sum %.0f, gradient %.0f, scale+intercept+early threshold %.0f, "
                        "step+stores %.0f, forward %.0f\n", c[0] / its, c[1] / its, c[2] / its, c[3] / its, c[4] / its, c[5] / its);
      }
      if (wide && s->d.dbg && out[0] > 0) {   // development aid: shader-clock cycles of thread 0 per phase of the wide kernel
        unsigned long long ph[6];
        SGD_HIP_TRY(hipMemcpy(ph, s->d.dbg, sizeof(ph), hipMemcpyDeviceToHost));
        (void)hipMemset(s->d.dbg, 0, sizeof(ph));
        const double its = (double)out[0] * (double)draws_per_epoch;
        fprintf(stderr, "[sgdnet] wide exact kernel, cycles per iteration (thread 0): loads+dot+sum %.0f, scale %.0f, barrier A %.0f, "
                        "class %.0f, barrier B %.0f, step %.0f\n",
                (double)ph[0] / its, (double)ph[1] / its, (double)ph[2] / its, (double)ph[3] / its, (double)ph[4] / its, (double)ph[5] / its);
      }
#endif
      done += (unsigned)out[0];
      converged = out[1];
      stream_offset += (int64_t)out[0] * draws_per_epoch;
    }
    s->w_prev_valid = true;
  } else if (mode == SGDNET_MODE_BATCHED) {
    rc = check_batched_ok(s);
    if (rc) return rc;
    if (batch < 1) batch = 1;
    if (batch > draws_per_epoch) batch = draws_per_epoch;
    rc = set_batch_shape(s, batch, draws_per_epoch);
    if (rc) return rc;
    s->lam.stream_base = stream_offset;
    s->lam.stream_wrap = stream_wrap_for(s, stream_offset, draws_per_epoch);
    rc = prepare_stream_slot(s, batch, stream_offset, draws_per_epoch, (int)max_epochs);
    if (rc) return rc;
    rc = push_lam(s);
    if (rc) return rc;
    if (s->d.standardize) {
      rc = launch_cw_init(s->d, s->lam_dev, s->st);
      if (rc) return rc;
    }
    // ConvergenceCheck{w, tol}: w_prev starts as the warm-start w (saga-sparse.h:251)
    SGD_HIP_TRY(hipMemcpyAsync(s->d.w_prev, s->d.w, wbytes, hipMemcpyDeviceToDevice, s->st));
    rc = ensure_graph(s, batch, draws_per_epoch);
    if (rc) return rc;
    const int nb = n_batches(batch, draws_per_epoch);
    while (done < max_epochs && !converged) {
      rc = check_stream(s, s->lam.stream_base, draws_per_epoch);
      if (rc) return rc;
      const auto tl0 = std::chrono::steady_clock::now();
      static hipEvent_t tev0 = nullptr, tev1 = nullptr;
      const bool tr = getenv("SGDNET_TRACE") != nullptr;
      if (tr && !tev0) {
        (void)hipEventCreate(&tev0);
        (void)hipEventCreate(&tev1);
      }
      if (tr) (void)hipEventRecord(tev0, s->st);
      rc = launch_epoch(s, batch, draws_per_epoch);
      if (rc) return rc;
      if (tr) (void)hipEventRecord(tev1, s->st);
      g_trace_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
      lam_advance(s, draws_per_epoch, nb);     // mirrors end_epoch on the device
      if (losses) {
        double sum = 0.0;
        rc = device_loss_sum(s, &sum);
        if (rc) return rc;
        losses[done] = sum / (double)s->d.n;
      }
      const auto tc0 = std::chrono::steady_clock::now();
      rc = device_convergence(s, tol, &converged);
      if (rc) return rc;
      if (s->fused_in_graph && s->fused_abort_seen) {
        bool rerun = false;
        rc = fused_recover(s, s->fused_abort_seen, draws_per_epoch, nb, &rerun);
        if (rc) return rc;
        if (rerun) {                            // nothing was modified: the same epoch as separate launches
          converged = 0;
          rc = ensure_graph(s, batch, draws_per_epoch);
          if (rc) return rc;
          continue;
        }
      }
      rc = check_bins(s);
      if (rc) return rc;
      g_trace_conv += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
      if (tr) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, tev0, tev1) == hipSuccess) g_trace_graph += ms * 1e-3;
      }
      ++g_trace_epochs;
      ++done;
    }
    s->w_prev_valid = true;
  } else {
    set_error("unknown mode %d", mode);
    return SGDNET_EINVAL;
  }
  *epochs_run = done;
  *converged_out = converged;
  return SGDNET_OK;
}

int sgdnet_solver_enqueue_epochs(sgdnet_solver* s, int64_t batch, int64_t stream_offset,
                                 int64_t draws_per_epoch, int n_epochs) {
  if (!s || draws_per_epoch <= 0 || n_epochs <= 0 || !s->penalty_set) {
    set_error("sgdnet_solver_enqueue_epochs: invalid argument");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = check_batched_ok(s);
  if (rc) return rc;
  rc = check_stream(s, stream_offset, draws_per_epoch * n_epochs);
  if (rc) return rc;
  if (batch < 1) batch = 1;
  if (batch > draws_per_epoch) batch = draws_per_epoch;
  rc = set_batch_shape(s, batch, draws_per_epoch);
  if (rc) return rc;
  s->lam.stream_base = stream_offset;
  s->lam.stream_wrap = stream_wrap_for(s, stream_offset, draws_per_epoch);
  rc = prepare_stream_slot(s, batch, stream_offset, draws_per_epoch, n_epochs);
  if (rc) return rc;
  rc = push_lam(s);
  if (rc) return rc;
  if (s->d.standardize) {
    rc = launch_cw_init(s->d, s->lam_dev, s->st);
    if (rc) return rc;
  }
  rc = ensure_graph(s, batch, draws_per_epoch);
  if (rc) return rc;
  const int nb = n_batches(batch, draws_per_epoch);
  for (int e = 0; e < n_epochs; ++e) {
    rc = launch_epoch(s, batch, draws_per_epoch);
    if (rc) return rc;
    lam_advance(s, draws_per_epoch, nb);
  }
  return SGDNET_OK;
}

// binned form: a bin that overflowed dropped entries -- the epoch's result is not the algorithm's
static int check_bins(sgdnet_solver* s) {
  if (s->d.R <= 0 || !s->d.bins || !s->d.bin_err) return SGDNET_OK;
  int flag = 0;
  SGD_HIP_TRY(hipMemcpyAsync(&flag, s->d.bin_err, sizeof(int), hipMemcpyDeviceToHost, s->st));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  if (flag) {
    SGD_HIP_TRY(hipMemsetAsync(s->d.bin_err, 0, sizeof(int), s->st));
    s->bin_overflowed = true;       // sgdnet_fit_* recovers (solver_grow_bins); a direct caller sees the error
    set_error("batched mode (binned form): a feature range received more entries in one batch than its bin holds "
              "(the epochs since the last synchronisation are void); pass a smaller batch");
    return SGDNET_EUNSUPPORTED;
  }
  return SGDNET_OK;
}

int sgdnet_solver_sync(sgdnet_solver* s) {
  if (!s) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  if (s->fused_in_graph && s->d.vsync) {        // epochs enqueued without a check of their own: did a fused launch give up?
    unsigned code = 0;
    SGD_HIP_TRY(hipMemcpyAsync(&code, s->d.vsync + vs_fused_sync_sticky_word(), sizeof(unsigned), hipMemcpyDeviceToHost, s->st));
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    if (code) {
      bool rerun = false;
      (void)fused_recover(s, 2, 0, 0, &rerun);
      drop_graph(s);
      set_error("batched mode: a fused epoch launch %s; the epochs enqueued since the last synchronisation are void "
                "(sgdnet_set_option(\"fused_epoch\", 0) keeps the separate launches)",
                code == 1 ? "could not become resident on the GPU (is it shared with another process?)"
                          : "timed out inside the epoch");
      return SGDNET_EHIP;
    }
  }
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return check_bins(s);
}

int sgdnet_solver_profile_epoch(sgdnet_solver* s, int64_t batch, int64_t stream_offset,
                                int64_t draws_per_epoch, double* gather_ms, int* gather_launches,
                                double* sweep_ms, int* sweep_launches) {
  if (!s || draws_per_epoch <= 0 || !s->penalty_set) {
    set_error("sgdnet_solver_profile_epoch: invalid argument");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = check_batched_ok(s);
  if (rc) return rc;
  rc = check_stream(s, stream_offset, draws_per_epoch);
  if (rc) return rc;
  if (batch < 1) batch = 1;
  if (batch > draws_per_epoch) batch = draws_per_epoch;
  rc = set_batch_shape(s, batch, draws_per_epoch);
  if (rc) return rc;
  s->lam.stream_base = stream_offset;
  s->lam.stream_wrap = stream_wrap_for(s, stream_offset, draws_per_epoch);
  rc = prepare_stream_slot(s, batch, stream_offset, draws_per_epoch, 1);
  if (rc) return rc;
  rc = push_lam(s);
  if (rc) return rc;
  if (s->d.standardize) {
    rc = launch_cw_init(s->d, s->lam_dev, s->st);
    if (rc) return rc;
  }
  std::vector<hipEvent_t> ev;
#ifdef SGDNET_PHASE_TIMING
  const bool fused_prof = vs_active(s, batch) && vs_fused_active(s);
  if (fused_prof && s->d.dbg) SGD_HIP_TRY(hipMemsetAsync(s->d.dbg, 0, sizeof(unsigned long long) * 16 * 1024, s->st));
#endif
  rc = enqueue_epoch_kernels(s, batch, draws_per_epoch, &ev);
  if (rc) return rc;
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  lam_advance(s, draws_per_epoch, n_batches(batch, draws_per_epoch));
  double g = 0.0, w = 0.0;
  int ng = 0;
  for (size_t i = 0; i + 3 < ev.size(); i += 4) {
    float ms = 0.f;
    SGD_HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    g += ms;
    SGD_HIP_TRY(hipEventElapsedTime(&ms, ev[i + 2], ev[i + 3]));
    w += ms;
    ++ng;
  }
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
#ifdef SGDNET_PHASE_TIMING
  if (s->d.dbg && fused_prof) {   // fused epoch kernel: thread 0's time per phase, summed over the rounds
    const int grid = s->d.V * s->d.v_bps;
    std::vector<unsigned long long> t(16 * 1024);
    SGD_HIP_TRY(hipMemcpy(t.data(), s->d.dbg, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost));
    static const char* nm[8] = {"stage w + ids", "draw loop", "publish slab", "wait shard (1)", "slice sweep", "merge",
                                "store + arrive", "wait shard (2)"};
    unsigned long long first = ~0ull, last = 0;
    for (int b = 0; b < grid; ++b) {
      first = std::min(first, t[b * 16 + 14]);
      last = std::max(last, t[b * 16 + 15]);
    }
    fprintf(stderr, "[phase] fused epoch kernel: span seen by the workgroups %.1f us (%d workgroups)\n", (double)(last - first) / 100.0, grid);
    double tot_mean = 0;
    for (int ph = 0; ph < 8; ++ph) {
      double sum = 0, mx = 0, mn = 1e30;
      for (int b = 0; b < grid; ++b) {
        const double dt = (double)t[b * 16 + ph] / 100.0;
        sum += dt; mx = std::max(mx, dt); mn = std::min(mn, dt);
      }
      tot_mean += sum / grid;
      fprintf(stderr, "[phase] %-16s per epoch: mean %7.1f us  min %7.1f  max %7.1f\n", nm[ph], sum / grid, mn, mx);
    }
    fprintf(stderr, "[phase] sum of the means %.1f us\n", tot_mean);
    for (int v = 0; v < s->d.V; ++v) {
      double a0 = 0, a1 = 0;
      for (int b = v * s->d.v_bps; b < (v + 1) * s->d.v_bps; ++b) {
        a0 += (double)(t[b * 16 + 14] - first) / 100.0;
        a1 += (double)(t[b * 16 + 15] - first) / 100.0;
      }
      fprintf(stderr, "[phase]   shard %d: mean start %.1f us, mean end %.1f us\n", v, a0 / s->d.v_bps, a1 / s->d.v_bps);
    }
  } else
  if (s->d.dbg && binned_active(s->d, (int)batch)) {   // binned form: slots 0-5 gather, 6-10 range sweep
    std::vector<unsigned long long> t(16 * 1024);
    SGD_HIP_TRY(hipMemcpy(t.data(), s->d.dbg, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost));
    static const char* nm[10] = {"gather: init", "gather: draw loop", "gather: barrier", "gather: reserve runs",
                                 "gather: place entries", "", "sweep: zero + first entries", "sweep: entries -> LDS",
                                 "sweep: barrier", "sweep: update features"};
    for (int ph = 0; ph < 10; ++ph) {
      if (ph == 5) continue;
      double sum = 0, mx = 0;
      int cnt = 0;
      for (int b = 0; b < 1024; ++b) {
        if (!t[b * 16 + ph] || !t[b * 16 + ph + 1] || t[b * 16 + ph + 1] < t[b * 16 + ph]) continue;
        const double dt = (double)(t[b * 16 + ph + 1] - t[b * 16 + ph]) / 100.0;
        sum += dt; mx = std::max(mx, dt); ++cnt;
      }
      if (cnt) fprintf(stderr, "[phase] %-28s mean %6.2f us max %6.2f us (%d workgroups)\n", nm[ph], sum / cnt, mx, cnt);
    }
    for (int k0 : {0, 6}) {
      unsigned long long first = ~0ull, last = 0;
      const int k1 = k0 == 0 ? 5 : 10;
      for (int b = 0; b < 1024; ++b) {
        if (t[b * 16 + k0] && t[b * 16 + k0] < first) first = t[b * 16 + k0];
        if (t[b * 16 + k1] > last) last = t[b * 16 + k1];
      }
      fprintf(stderr, "[phase] %s span seen by the workgroups %.2f us\n", k0 == 0 ? "gather" : "sweep", (double)(last - first) / 100.0);
      std::vector<double> st, en;
      for (int b = 0; b < 1024; ++b)
        if (t[b * 16 + k0] && t[b * 16 + k1] >= t[b * 16 + k0]) {
          st.push_back((double)(t[b * 16 + k0] - first) / 100.0);
          en.push_back((double)(t[b * 16 + k1] - first) / 100.0);
        }
      std::sort(st.begin(), st.end());
      std::sort(en.begin(), en.end());
      if (!st.empty())
        fprintf(stderr, "[phase]   workgroup start offsets: p10 %.1f p50 %.1f p90 %.1f max %.1f us; end offsets: p10 %.1f p50 %.1f p90 %.1f max %.1f us\n",
                st[st.size() / 10], st[st.size() / 2], st[st.size() * 9 / 10], st.back(), en[en.size() / 10], en[en.size() / 2],
                en[en.size() * 9 / 10], en.back());
    }
  } else
  if (s->d.dbg) {   // stamps of the epoch's last gather launch (the tail batch unless batch divides the epoch)
    std::vector<unsigned long long> t(16 * 256);
    SGD_HIP_TRY(hipMemcpy(t.data(), s->d.dbg, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull, last = 0;
    for (int b = 0; b < 256; ++b) {
      if (t[b * 16] && t[b * 16] < first) first = t[b * 16];
      if (t[b * 16 + 5] > last) last = t[b * 16 + 5];
    }
    if (getenv("SGDNET_PHASE_DUMP")) {
      fprintf(stderr, "[phase-dump] draw loop us by workgroup:");
      for (int b = 0; b < 256; ++b) fprintf(stderr, " %.1f", (double)(t[b * 16 + 2] - t[b * 16 + 1]) / 100.0);
      fprintf(stderr, "\n[phase-dump] draw loop + barrier us by workgroup:");
      for (int b = 0; b < 256; ++b) fprintf(stderr, " %.1f", (double)(t[b * 16 + 3] - t[b * 16 + 1]) / 100.0);
      fprintf(stderr, "\n[phase-dump] start offset us by workgroup:");
      for (int b = 0; b < 256; ++b) fprintf(stderr, " %.1f", (double)(t[b * 16 + 1] - first) / 100.0);
      fprintf(stderr, "\n");
    }
    static const char* nm[11] = {"lds zero+sync", "draw loop", "barrier", "slab flush", "d0 partial", "",
                                 "stream idx", "record loads", "w gather", "M exchange", "scatter"};
    fprintf(stderr, "[phase] kernel span %.2f us (memtime ticks at 100 MHz)\n", (double)(last - first) / 100.0);
    double start_spread = 0;
    for (int b = 0; b < 256; ++b) if (t[b * 16]) start_spread = std::max(start_spread, (double)(t[b * 16] - first));
    fprintf(stderr, "[phase] workgroup start spread %.2f us\n", start_spread / 100.0);
    for (int ph = 0; ph < 11; ++ph) {
      if (ph == 5) continue;
      const int a = ph < 5 ? ph : ph, bslot = a + 1;
      double sum = 0, mx = 0; int cnt = 0;
      for (int b = 0; b < 256; ++b) {
        if (!t[b * 16 + a] || !t[b * 16 + bslot]) continue;
        const double dt = (double)(t[b * 16 + bslot] - t[b * 16 + a]) / 100.0;
        sum += dt; mx = std::max(mx, dt); ++cnt;
      }
      if (cnt) fprintf(stderr, "[phase] %-14s mean %6.2f us max %6.2f us\n", nm[ph], sum / cnt, mx);
    }
  }
#endif
  if (gather_ms) *gather_ms = g;
  if (gather_launches) *gather_launches = ng;
  if (sweep_ms) *sweep_ms = w;
  if (sweep_launches) *sweep_launches = ng;
  return SGDNET_OK;
}

int sgdnet_solver_gather_form(const sgdnet_solver* s, int64_t batch) {
  if (!s || batch < 1) return 0;
  if (vs_active(s, batch) && vs_fused_active(s)) return 3;
  if (binned_active(s->d, (int)batch)) return 2;
  return batch_gather_slab_doubles(s->d, (int)batch) > 0 ? 1 : 0;
}

int sgdnet_solver_deviance(sgdnet_solver* s, double* out) {
  if (!s || !out) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  double sum = 0.0;
  int rc = device_loss_sum(s, &sum);
  if (rc) return rc;
  *out = 2.0 * sum;
  return SGDNET_OK;
}

int64_t sgdnet_solver_delta_len(const sgdnet_solver* s) {
  if (!s) return 0;
  return 2 * (int64_t)s->d.K * s->d.p + 2 * s->d.K;
}

int sgdnet_solver_snapshot(sgdnet_solver* s) {
  if (!s) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  const size_t KP = (size_t)s->d.K * (size_t)s->d.p, K = (size_t)s->d.K;
  SGD_HIP_TRY(hipMemcpyAsync(s->ref, s->d.G, 8 * KP, hipMemcpyDeviceToDevice, s->st));
  SGD_HIP_TRY(hipMemcpyAsync(s->ref + KP, s->d.w, 8 * KP, hipMemcpyDeviceToDevice, s->st));
  SGD_HIP_TRY(hipMemcpyAsync(s->ref + 2 * KP, s->d.gb, 8 * K, hipMemcpyDeviceToDevice, s->st));
  SGD_HIP_TRY(hipMemcpyAsync(s->ref + 2 * KP + K, s->d.b, 8 * K, hipMemcpyDeviceToDevice, s->st));
  return SGDNET_OK;
}

void* sgdnet_solver_stream(sgdnet_solver* s) { return s ? static_cast<void*>(s->st) : nullptr; }

int sgdnet_solver_export_delta_weighted_async(sgdnet_solver* s, void* device_buf, double weight) {
  if (!s || !device_buf) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  return launch_delta_export(s->d, s->ref, static_cast<double*>(device_buf), weight, s->st);
}

int sgdnet_solver_export_delta_async(sgdnet_solver* s, void* device_buf) {
  return sgdnet_solver_export_delta_weighted_async(s, device_buf, 1.0);
}

int sgdnet_solver_apply_merged_async(sgdnet_solver* s, const void* device_buf, double w_weight) {
  if (!s || !device_buf) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  return launch_delta_apply(s->d, s->ref, static_cast<const double*>(device_buf), w_weight, s->st);
}

int sgdnet_solver_export_delta(sgdnet_solver* s, void* device_buf) {
  int rc = sgdnet_solver_export_delta_async(s, device_buf);
  if (rc) return rc;
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

int sgdnet_solver_apply_merged(sgdnet_solver* s, const void* device_buf, double w_weight) {
  int rc = sgdnet_solver_apply_merged_async(s, device_buf, w_weight);
  if (rc) return rc;
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

int sgdnet_solver_last_change(const sgdnet_solver* s, double* max_change, double* max_size) {
  if (!s || !max_change || !max_size) return SGDNET_EINVAL;
  *max_change = s->last_change;
  *max_size = s->last_size;
  return SGDNET_OK;
}

int64_t sgdnet_auto_batch(double max_sample_sqnorm, double max_feature_mean_sq) {
  if (!(max_sample_sqnorm > 0.0) || !(max_feature_mean_sq > 0.0)) return 64;
  const double b = 2.0 * max_sample_sqnorm / max_feature_mean_sq;
  if (!(b < 131072.0)) return 131072;
  return b < 64.0 ? 64 : (int64_t)b;
}

int64_t sgdnet_shard_window(int64_t window, int64_t draws_per_shard) {
  if (window < 1 || draws_per_shard <= window || draws_per_shard % window == 0) return window;
  const int64_t rounds = (draws_per_shard + window - 1) / window;          // >= 2
  const int64_t longer = (draws_per_shard + rounds - 2) / (rounds - 1);
  return longer <= window + window / 8 ? longer : window;
}

int64_t sgdnet_solver_sync_buffer_len(const sgdnet_solver* s) {
  if (!s) return 0;
  return (int64_t)s->d.K * s->d.p + 2 * 256 * (int64_t)s->d.K;
}

int sgdnet_solver_sync_bind(sgdnet_solver* s, void* device_buf) {
  if (!s) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  drop_graph(s);
  if (device_buf) {
    int rc = check_batched_ok(s);
    if (rc) return rc;
    if (!s->sparse) {
      set_error("the synchronous sharded mode is implemented for sparse x");
      return SGDNET_EUNSUPPORTED;
    }
    if (!s->own_D) {
      s->own_D = s->d.D;
      s->own_d0 = s->d.d0_part;
    }
    double* buf = static_cast<double*>(device_buf);
    SGD_HIP_TRY(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)sgdnet_solver_sync_buffer_len(s), s->st));
    s->d.D = buf;
    s->d.d0_part = buf + (int64_t)s->d.K * s->d.p;
    s->d.force_global = 1;
  } else if (s->own_D) {
    s->d.D = s->own_D;
    s->d.d0_part = s->own_d0;
    s->own_D = s->own_d0 = nullptr;
    s->d.force_global = 0;
  }
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  return SGDNET_OK;
}

int sgdnet_solver_sync_begin(sgdnet_solver* s, int64_t stream_offset, int64_t draws_local_per_epoch) {
  if (!s || !s->penalty_set || !s->d.force_global || draws_local_per_epoch <= 0) {
    set_error("sgdnet_solver_sync_begin: bind a sync buffer and set the penalty first");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = check_stream(s, stream_offset, draws_local_per_epoch);
  if (rc) return rc;
  s->lam.stream_base = stream_offset;
  s->lam.stream_wrap = 0;
  s->lam.draws_per_epoch = draws_local_per_epoch;
  rc = push_lam(s);
  if (rc) return rc;
  if (s->d.standardize) rc = launch_cw_init(s->d, s->lam_dev, s->st);
  return rc;
}

int sgdnet_solver_sync_gather(sgdnet_solver* s, int64_t t0_local, int64_t m_local, int round) {
  if (!s || !s->d.force_global || t0_local < 0 || m_local < 0) return SGDNET_EINVAL;
  if (m_local == 0) return SGDNET_OK;
  SGD_HIP_TRY(hipSetDevice(s->device));
  return launch_batch_gather(s->d, s->lam_dev, t0_local, (int)m_local, 0, round, s->st);
}

int sgdnet_solver_sync_sweep(sgdnet_solver* s, int64_t m_global, int64_t m_local, int round) {
  if (!s || !s->d.force_global || m_global <= 0) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  double r_m, ls_m;
  batch_factors(s->lam.alpha, s->lam.gamma, m_global, &r_m, &ls_m);
  return launch_batch_sweep(s->d, s->lam_dev, s->lam.penalty, 0, (int)(m_local > 0 ? m_local : 1), round, s->st,
                            nullptr, nullptr, r_m, ls_m, (double)m_global);
}

int sgdnet_solver_sync_end(sgdnet_solver* s, int rounds) {
  if (!s || rounds <= 0) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  int rc = launch_epoch_end(s->lam_dev, rounds, s->st);
  if (rc) return rc;
  lam_advance(s, s->lam.draws_per_epoch, rounds);   // mirrors end_epoch on the device
  return SGDNET_OK;
}

int sgdnet_solver_set_virtual_shards(sgdnet_solver* s, int n_shards) {
  if (!s || n_shards < 0 || n_shards > 8) {
    set_error("sgdnet_solver_set_virtual_shards: 0..8 shards");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  drop_graph(s);
  for (void* q : s->vs_owned) (void)hipFree(q);
  s->vs_owned.clear();
  SagaDev& d = s->d;
  d.V = 0;
  d.vw = d.vG = d.vb = d.vgb = d.vd0 = d.vref = d.vcw = d.vx = d.vpub = nullptr;
  d.vsync = d.vcol = nullptr;
  d.peers = nullptr;                            // links name the buffers just freed: link again
  d.n_peers = 0;
  if (n_shards < 2) return SGDNET_OK;
  if (d.K > 16) {
    set_error("virtual shards: up to 16 classes");
    return SGDNET_EUNSUPPORTED;
  }
  const int64_t KP = (int64_t)d.K * d.p;
  auto alloc = [&](double** out, size_t count) -> int {
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(double) * count) != hipSuccess) return SGDNET_ENOMEM;
    (void)hipMemset(q, 0, sizeof(double) * count);
    s->vs_owned.push_back(q);
    *out = static_cast<double*>(q);
    return SGDNET_OK;
  };
  int rc = alloc(&d.vw, (size_t)n_shards * KP);
  if (!rc) rc = alloc(&d.vG, (size_t)n_shards * KP);
  if (!rc) rc = alloc(&d.vb, 8 * (size_t)d.K);
  if (!rc) rc = alloc(&d.vgb, 8 * (size_t)d.K);
  if (!rc) rc = alloc(&d.vcw, 8 * (size_t)d.K);
  if (!rc) rc = alloc(&d.vd0, 256 * (size_t)d.K);
  if (!rc) rc = alloc(&d.vref, (size_t)(2 * KP + 2 * d.K));
  if (!rc && d.K == 1) {
    // the fused epoch kernel's barrier counters and reference copies (ordinary device memory), and what its merges
    // exchange -- slice counters and published slices -- in fine-grained memory: linked solvers on other GPUs add to
    // those counters and read those slices while the kernels run (sgdnet_solver_link_peers)
    auto alloc_fg = [&](void** out, size_t bytes) -> int {
      void* q = nullptr;
      if (hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&q, bytes) != hipSuccess) return SGDNET_ENOMEM;   // (one GPU: any device memory will do)
      }
      (void)hipMemset(q, 0, bytes);
      s->vs_owned.push_back(q);
      *out = q;
      return SGDNET_OK;
    };
    double* words = nullptr;
    rc = alloc(&words, (vs_fused_sync_words() * sizeof(unsigned) + sizeof(double) - 1) / sizeof(double));
    d.vsync = reinterpret_cast<unsigned*>(words);
    if (!rc) rc = alloc(&d.vx, vs_fused_exchange_doubles(d, n_shards));
    if (!rc) rc = alloc_fg(reinterpret_cast<void**>(&d.vcol), vs_fused_col_words() * sizeof(unsigned));
    if (!rc) rc = alloc_fg(reinterpret_cast<void**>(&d.vpub), vs_fused_publish_doubles(d, n_shards) * sizeof(double));
  }
  if (rc) {
    set_error("virtual shards: out of device memory");
    return rc;
  }
  d.V = n_shards;
  d.v_bps = lds_target_grid(d) / n_shards;
  // shard v owns the samples [v * base + min(v, rem), ...): sgdnet_amd/parallel.py shard_bounds
  const int64_t base = d.n / n_shards, rem = d.n % n_shards;
  for (int v = 0; v < 8; ++v) d.v_size[v] = v < n_shards ? (double)(base + (v < rem ? 1 : 0)) : 0.0;
  if (!vs_eligible(d, 0)) {   // the gather forms that carry shards keep their tables in LDS
    for (void* q : s->vs_owned) (void)hipFree(q);
    s->vs_owned.clear();
    d.V = 0;
    d.vw = d.vG = d.vb = d.vgb = d.vd0 = d.vref = d.vcw = d.vx = d.vpub = nullptr;
    d.vsync = d.vcol = nullptr;
    set_error("virtual shards: n_features too large for the LDS-resident gather");
    return SGDNET_EUNSUPPORTED;
  }
  return SGDNET_OK;
}

int sgdnet_solver_epoch_timing(sgdnet_solver* s, int enable, double* sum_ms, int* launches) {
  if (!s) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  double tot = 0.0;
  int cnt = 0;
  for (size_t i = 0; i + 1 < s->epoch_ev.size(); i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->epoch_ev[i], s->epoch_ev[i + 1]) == hipSuccess) {
      tot += ms;
      ++cnt;
    }
  }
  for (hipEvent_t e : s->epoch_ev) (void)hipEventDestroy(e);
  s->epoch_ev.clear();
  s->time_epochs = enable != 0;
  if (sum_ms) *sum_ms = tot;
  if (launches) *launches = cnt;
  return SGDNET_OK;
}

int sgdnet_solver_set_cu_budget(sgdnet_solver* s, int cus) {
  if (!s || cus < 0) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  s->d.cu_budget = cus;
  if (s->d.V > 1) s->d.v_bps = lds_target_grid(s->d) / s->d.V;
  drop_graph(s);
  return SGDNET_OK;
}

int sgdnet_solver_link_peers(sgdnet_solver** solvers, int n) {
  if (!solvers || n < 1 || n > 8) {
    set_error("sgdnet_solver_link_peers: 1..8 solvers");
    return SGDNET_EINVAL;
  }
  for (int q = 0; q < n; ++q) {
    sgdnet_solver* s = solvers[q];
    if (!s || !s->d.vsync || !s->d.vx || s->d.V < 1 || s->d.V != solvers[0]->d.V || s->d.v_bps != solvers[0]->d.v_bps ||
        s->d.p != solvers[0]->d.p || s->d.K != 1) {
      set_error("sgdnet_solver_link_peers: every solver needs the same number of virtual shards (>= 2), workgroups per shard "
                "and features, and one response (rank %d does not)", q);
      return SGDNET_EUNSUPPORTED;
    }
    if (n > 1 && !vs_fused_eligible(s->d)) {
      set_error("sgdnet_solver_link_peers: the replica average across GPUs runs inside the fused epoch kernel, which rank %d's "
                "problem cannot use (sparse x, one response, an even number of features)", q);
      return SGDNET_EUNSUPPORTED;
    }
  }
  double tot = 0.0;
  for (int q = 0; q < n; ++q)
    for (int u = 0; u < solvers[q]->d.V; ++u) tot += solvers[q]->d.v_size[u];
  for (int q = 0; q < n; ++q) {
    sgdnet_solver* s = solvers[q];
    SGD_HIP_TRY(hipSetDevice(s->device));
    SGD_HIP_TRY(hipStreamSynchronize(s->st));
    drop_graph(s);
    if (n == 1) {
      s->d.peers = nullptr;
      s->d.n_peers = 0;
      continue;
    }
    for (int r = 0; r < n; ++r) {               // direct loads, stores and atomics on the other ranks' buffers
      if (solvers[r]->device == s->device) continue;
      int can = 0;
      SGD_HIP_TRY(hipDeviceCanAccessPeer(&can, s->device, solvers[r]->device));
      if (!can) {
        set_error("sgdnet_solver_link_peers: device %d cannot access device %d", s->device, solvers[r]->device);
        return SGDNET_EUNSUPPORTED;
      }
      const hipError_t e = hipDeviceEnablePeerAccess(solvers[r]->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
        set_error("hipDeviceEnablePeerAccess(%d -> %d) failed: %s", s->device, solvers[r]->device, hipGetErrorString(e));
        return SGDNET_EHIP;
      }
      (void)hipGetLastError();
    }
    FusedPeers h{};
    h.n = n;
    h.rank = q;
    h.tot_size = tot;
    for (int r = 0; r < n; ++r) {
      h.pub[r] = solvers[r]->d.vpub;
      h.sync[r] = solvers[r]->d.vcol;
      for (int u = 0; u < 8; ++u) h.vsize[r][u] = solvers[r]->d.v_size[u];
    }
    if (!s->peers_dev) {
      void* pd = nullptr;
      SGD_HIP_TRY(hipMalloc(&pd, sizeof(FusedPeers)));
      s->owned.push_back(pd);
      s->peers_dev = static_cast<FusedPeers*>(pd);
    }
    SGD_HIP_TRY(hipMemcpy(s->peers_dev, &h, sizeof(FusedPeers), hipMemcpyHostToDevice));
    SGD_HIP_TRY(hipMemset(s->d.vsync, 0, vs_fused_sync_words() * sizeof(unsigned)));   // slice counters and launch count start together
    SGD_HIP_TRY(hipMemset(s->d.vcol, 0, vs_fused_col_words() * sizeof(unsigned)));
    s->d.peers = s->peers_dev;
    s->d.n_peers = n;
  }
  return SGDNET_OK;
}

// ---- the same link between solvers of DIFFERENT processes (one process per GPU: bench.py under torch.distributed.run) ----
// info: 2 hipIpcMemHandle_t (exchange buffer, barrier counters) + 8 shard sizes + V + workgroups per shard + features
struct PeerInfo {
  hipIpcMemHandle_t vx, vsync;
  double vsize[8];
  int V, v_bps;
  int64_t p;
};

int sgdnet_solver_peer_info_bytes(void) { return (int)sizeof(PeerInfo); }

int sgdnet_solver_peer_info(sgdnet_solver* s, void* out) {
  if (!s || !out || !s->d.vpub || !s->d.vcol) {
    set_error("sgdnet_solver_peer_info: set the virtual shards first (one response, sparse x)");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  PeerInfo h{};
  SGD_HIP_TRY(hipIpcGetMemHandle(&h.vx, s->d.vpub));
  SGD_HIP_TRY(hipIpcGetMemHandle(&h.vsync, s->d.vcol));
  for (int u = 0; u < 8; ++u) h.vsize[u] = s->d.v_size[u];
  h.V = s->d.V;
  h.v_bps = s->d.v_bps;
  h.p = s->d.p;
  memcpy(out, &h, sizeof(h));
  return SGDNET_OK;
}

int sgdnet_solver_link_ipc(sgdnet_solver* s, int rank, int n, const void* infos) {
  if (!s || !infos || n < 2 || n > 8 || rank < 0 || rank >= n) {
    set_error("sgdnet_solver_link_ipc: 2..8 ranks");
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  if (!vs_fused_eligible(s->d)) {
    set_error("sgdnet_solver_link_ipc: the replica average across GPUs runs inside the fused epoch kernel, which this "
              "problem cannot use (sparse x, one response, an even number of features, virtual shards set)");
    return SGDNET_EUNSUPPORTED;
  }
  const PeerInfo* I = static_cast<const PeerInfo*>(infos);
  FusedPeers h{};
  h.n = n;
  h.rank = rank;
  for (int r = 0; r < n; ++r) {
    if (I[r].V != s->d.V || I[r].v_bps != s->d.v_bps || I[r].p != s->d.p) {
      set_error("sgdnet_solver_link_ipc: rank %d has %d shards of %d workgroups on %lld features, this rank %d of %d on %lld",
                r, I[r].V, I[r].v_bps, (long long)I[r].p, s->d.V, s->d.v_bps, (long long)s->d.p);
      return SGDNET_EUNSUPPORTED;
    }
    for (int u = 0; u < 8; ++u) {
      h.vsize[r][u] = I[r].vsize[u];
      if (u < I[r].V) h.tot_size += I[r].vsize[u];
    }
    if (r == rank) {
      h.pub[r] = s->d.vpub;
      h.sync[r] = s->d.vcol;
    } else {
      void *px = nullptr, *py = nullptr;
      SGD_HIP_TRY(hipIpcOpenMemHandle(&px, I[r].vx, hipIpcMemLazyEnablePeerAccess));
      SGD_HIP_TRY(hipIpcOpenMemHandle(&py, I[r].vsync, hipIpcMemLazyEnablePeerAccess));
      s->ipc_opened.push_back(px);
      s->ipc_opened.push_back(py);
      h.pub[r] = static_cast<double*>(px);
      h.sync[r] = static_cast<unsigned*>(py);
    }
  }
  drop_graph(s);
  if (!s->peers_dev) {
    void* pd = nullptr;
    SGD_HIP_TRY(hipMalloc(&pd, sizeof(FusedPeers)));
    s->owned.push_back(pd);
    s->peers_dev = static_cast<FusedPeers*>(pd);
  }
  SGD_HIP_TRY(hipMemcpy(s->peers_dev, &h, sizeof(FusedPeers), hipMemcpyHostToDevice));
  // (the counters were zeroed when the shards were set; every rank links before any of them enqueues an epoch -- the
  //  caller's barrier -- so nothing is cleared here that a peer may already have added to)
  s->d.peers = s->peers_dev;
  s->d.n_peers = n;
  return SGDNET_OK;
}

int sgdnet_solver_set_merge_period(sgdnet_solver* s, int64_t draws_per_shard) {
  if (!s || draws_per_shard < 0) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  s->vs_period = draws_per_shard;
  drop_graph(s);
  return SGDNET_OK;
}

int sgdnet_solver_set_n_total(sgdnet_solver* s, int64_t n_total) {
  if (!s || n_total <= 0) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  SGD_HIP_TRY(hipStreamSynchronize(s->st));
  s->d.n_total = (double)n_total;
  drop_graph(s);   // captured kernels carry the old value
  return SGDNET_OK;
}

int sgdnet_solver_convergence(sgdnet_solver* s, double tol, int* converged) {
  if (!s || !converged) return SGDNET_EINVAL;
  SGD_HIP_TRY(hipSetDevice(s->device));
  return device_convergence(s, tol, converged);
}

}  // extern "C"

namespace sgdnet {

int solver_create_adopting(const sgdnet_problem* pb, DeviceSetup& S, sgdnet_solver** out) {
  return solver_create_impl(pb, &S, out);
}

}  // namespace sgdnet
