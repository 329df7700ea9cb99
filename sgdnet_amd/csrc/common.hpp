// Shared host/device declarations of the gfx950 SAGA backend.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgdnet_hip.h"

// Timing-only variants of the gather (no gradient-memory exchange, no scatter, ...) exist for the
// ablation tables of DESIGN.md 5.  They compute wrong results, so the shipped library does not
// contain them: build with EXTRA_FLAGS=-DSGDNET_EXPERIMENTS to get the SGDNET_ABLATE switch.
#ifdef SGDNET_EXPERIMENTS
#define SGD_ABLATE(d, bits) (((d).ablate & (bits)) != 0)
#else
#define SGD_ABLATE(d, bits) false
#endif

#include <stdlib.h>

namespace sgdnet {

void set_error(const char* fmt, ...);

// Kernel A/B switches of the experiments behind DESIGN.md 5 (SGDNET_GATHER, SGDNET_LANES8, SGDNET_COMPACT,
// SGDNET_W_LDS, SGDNET_BINNED, SGDNET_BIN_RANGES, SGDNET_LDS_GRID, SGDNET_EXACT_SMALL / _WIDE, SGDNET_REC_ALIGN):
// environment variables in -DSGDNET_EXPERIMENTS builds, constants in the shipped library -- a user's
// environment cannot change which kernel a fit runs.  What a user may tune is sgdnet_set_option
// (include/sgdnet_hip.h).
#ifdef SGDNET_EXPERIMENTS
inline int exp_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
inline const char* exp_env_str(const char* name) { return getenv(name); }
#else
inline int exp_env_int(const char*, int dflt) { return dflt; }
inline const char* exp_env_str(const char*) { return nullptr; }
#endif

// sgdnet_set_option values (solver.cpp)
enum Option { kOptVirtualShards = 0, kOptRngGenerators, kOptWindowEigenvalue, kOptHostSetup, kOptExactEpochBlocks,
              kOptExactRowRegisters, kOptFusedEpoch, kOptCount };
int option(Option o);

#define SGD_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ::sgdnet::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                          __FILE__, __LINE__);                                         \
      return SGDNET_EHIP;                                                              \
    }                                                                                  \
  } while (0)

// src/constants.h:22 of the reference: 100 * DBL_EPSILON
constexpr double kSmall = 100.0 * 2.220446049250313e-16;

// The sample-order generators as the fused epoch kernel of the virtual shards runs them (saga_batched.hip): its
// spare workgroups produce the NEXT epoch's raw words and jump the generators' start states while the epoch runs.
// Lives in device memory (the captured launch reads it; `gen` is advanced by the kernel itself).
struct RngDev {
  uint32_t* state[2];     // start states of the generators: generation g reads state[g & 1], its jump writes state[(g + 1) & 1]
  uint32_t* ends;         // where the state step leaves the generators (not used)
  uint32_t* stream;       // two slots of n words: generation g fills slot g & 1
  const uint32_t* poly;   // x^n mod phi(x): the jump of one epoch
  int64_t n;              // draws per epoch
  int64_t seg;            // words per generator
  int gens;
  unsigned gen;           // the generation the next producing launch makes
};

// Solvers linked for the replica average across GPUs (the fused epoch kernel's merge, saga_batched.hip): what every
// rank's kernel needs to reach the others -- their exchange buffers and barrier counters (peer pointers, or
// hipIpc mappings of them) and their shard sizes.  Lives in device memory.
struct FusedPeers {
  double* pub[8];         // every rank's published slices (its SagaDev::vpub), this rank's own included
  unsigned* sync[8];      // every rank's slice counters (SagaDev::vcol)
  double vsize[8][8];     // samples of rank q's shard u
  double tot_size;        // samples of the whole job
  int n, rank;
};

// Device view of one problem + its solver state.  Passed to kernels by value.
struct SagaDev {
  int family;
  int K;         // n_classes
  int Ky;        // rows of y
  int fit_intercept;
  int standardize;
  int64_t n;     // resident samples
  int64_t p;     // features
  double n_total;  // samples of the whole job (the 1/n of the gradient average)
  float avg_nnz;   // mean non-zeros per sample (sparse)
  // virtual shards (K == 1 LDS gather, DESIGN.md 8 "one GPU"): V locally normalised replicas of
  // (w, g_sum, b, g_sum_b) over V contiguous sample ranges, averaged periodically on the device
  int V;                 // 0 / 1: off
  int v_bps;             // gather workgroups per shard
  int64_t v_dps;         // draws per shard and epoch = stride between the shards' stream regions
  double v_size[8];      // samples per shard (g_sum normalisation, merge weights = size / n)
  double* vw;            // V x K*p
  double* vG;            // V x K*p
  double* vb;            // V
  double* vgb;           // V
  double* vcw;           // V: c . w of every replica (implicit centring)
  double* vd0;           // one intercept partial per gather workgroup
  double* vref;          // snapshot [g_sum | w | g_sum_b | b] the replicas started from
  // fused epoch of the virtual shards (saga_vs_epoch_kernel): counters of its in-launch barriers, and the
  // exchange buffer [2 parities x V published slices | V reference copies | c.w partials]
  unsigned* vsync;       // shard-local barrier counters, start / go / exit words
  double* vx;            // [V reference copies | c.w partials]
  // ... and what the merges exchange, in FINE-GRAINED memory of their own (linked solvers on other GPUs add to the slice
  // counters and read the published slices while the kernels run; kept apart from the per-round counters above, which
  // are polled every round: fine-grained memory for those cost the epoch 20 %)
  unsigned* vcol;        // slice counters col[i]: one 128-B line each
  double* vpub;          // 2 parities x V published slices [g_sum | w | g_sum_b | b]
  FusedPeers* peers;     // linked solvers (one per GPU): their replicas take part in the merge; n_peers <= 1: none
  int n_peers;
  int cu_budget;         // > 0: CUs this solver may fill (several linked solvers sharing one GPU in tests); 0: the device's
  int vs_xcd_local;      // fused epoch kernel: shards whose workgroups share an XCD hand off through its L2 (plain stores)
  RngDev* rngdev;        // sample-order generators inside the fused epoch kernel (cu_reserve workgroups), or nullptr
  unsigned long long* dbg;  // SGDNET_PHASE_TIMING builds only: per-workgroup phase stamps
  int cu_reserve;    // CUs left to the sample-order generators that run beside the epoch (LDS gather grids shrink by it)
  int force_global;  // synchronous sharded mode: always the global-atomic gather (D must be one array)
  int ablate;      // -DSGDNET_EXPERIMENTS builds only: SGDNET_ABLATE bit mask, timing-only variants of the gather (results are wrong)
  // data, sample-major (SURVEY.md 8a "x")
  const int64_t* ptr;
  const int32_t* idx;
  const double* val;
  const double* xd;   // dense p x n
  const double* c;    // x_center_scaled (p) or nullptr
  const double* y;    // Ky x n
  // packed per-sample records for the batched gather (saga_batched.hip)
  const char* rec;    // n records of rec_stride bytes
  const char* ovf;    // overflow records (256 B each)
  int rec_stride;
  int rec_cap;        // entries held by the main record
  int rec_val_off;    // byte offset of val[] inside a record
  // compact two-plane records of the K == 1 LDS gather (saga_batched.hip) or nullptr
  char* cP;           // n x 128 B: first cE entries (16-bit feature ids), response, gradient memory
  const char* cQ;     // n x 128 B: entries cE .. cE + 11 of the rows that have them
  const uint32_t* cmeta;  // 2 bits per sample: row longer than cE entries, response != 0 (binomial)
  int y_binary;       // binomial response verified to be 0 / 1 (solver.cpp): it may ride as one bit of cmeta
  int cE;             // entries in plane P: 12 (binomial 0 / 1 response: a bit of cmeta) or 11
  int m_rec;          // one-response sparse fits: the gradient memory is inside the records (cP + 120), not in M
  char* m_base;       // ... its address for sample s is m_base + s * m_stride: (M, 8) or (cP + 120, 128)
  int m_stride;
  // solver state (K fastest, like the reference's ArrayXXd K x p / K x n)
  double* w;
  double* G;      // g_sum
  double* M;      // g_memory
  double* b;      // intercept
  double* gb;     // g_sum_intercept
  double* w_prev; // ConvergenceCheck::w_prev
  unsigned* lag;
  // batched-mode scratch
  double* D;      // K x p scatter accumulator
  double* d0_part;  // 2 x 256 x K  partial sums of the intercept accumulator (two parity sets)
  double* slab;     // blocks x K x p  per-workgroup copies of D (LDS-privatised gather) or nullptr
  double* cw;       // 2 x 16 x K  slots of c.w (implicit centring in batched mode)
  int* claim;     // n     first-occurrence claims (K > 1)
  const uint32_t* stream;
  // binned form of the batched iteration (saga_batched.hip "binned"): K x p tables that fit no
  // LDS are split into R feature ranges; the gather bins every non-zero of the batch by range
  // and one workgroup per range accumulates its slice of D in LDS and sweeps it.  R == 0: off.
  int R;
  int range_max;               // features of the widest range (LDS of the range sweep)
  const int64_t* bin_off;      // R + 1: first entry of every bin (capacity follows the range's non-zero mass)
  const int32_t* range_lo;     // R + 1 feature boundaries
  const uint16_t* feat_range;  // p: range of every feature
  const uint16_t* range_coarse;  // n_coarse + 1: range of feature c << coarse_shift (the last entry: R - 1) -- where the
  int coarse_shift, n_coarse;    // binned gather's bisection for a feature's range starts (<= 2048 cells, staged in LDS)
  char* bins;                  // bin_off[R] entries {u32 draw, u32 feature, f64 value}
  unsigned* bin_count;         // R entries used (reset by the range's sweep)
  double* gcb;                 // batch x KS: gradient change of every draw of the batch
  int KS;                      // K rounded up to a power of two: row stride of gcb and wpad (a K-vector never straddles a 128-B line)
  double* wpad;                // p x KS copy of w read by the binned gather (== w when KS == K)
  int* bin_err;                // set when a bin overflowed (the epoch is then invalid)
};

// Per-lambda parameters; lives in device memory so captured graphs stay valid
// across the lambda path.
struct LamParams {
  int penalty;
  double gamma;
  double alpha;   // L2 strength
  double beta;    // L1 strength
  // batched mode: r^m and LS_m for the full batch and for the tail batch
  double r_full, ls_full;
  double r_tail, ls_tail;
  int64_t m_full, m_tail;
  // epoch bookkeeping for graph replays
  int64_t stream_base;   // offset of the current epoch in the resident stream
  int64_t stream_wrap;   // > 0: the epoch's end wraps stream_base at this length (the two-epoch buffer of the sample-order pipeline)
  int64_t draws_per_epoch;
  int batch_seq;         // running batch id (claims)
  int rng_generate;      // fused epoch kernel: this launch also produces the next generation of the sample order (SagaDev::rngdev)
  int stream_raw;        // the epoch's slot of the sample-order pipeline holds the generators' raw words: the fused epoch
                         // kernel turns them into draws itself (round 4; every other consumer gets a converted slot)
  // ConvergenceCheck scratch: bit patterns of max|dw| and max|w|
  unsigned long long max_change_bits;
  unsigned long long max_size_bits;
  double loss_acc;
  // fused epoch kernel: 0 the epoch ran; 1 the launch could not become resident and changed nothing (the host
  // runs the epoch as separate launches); 2 a wait inside the epoch timed out (the epoch is void)
  int fused_abort;
};

struct ExactCtl {
  int64_t stream_off;
  int64_t nit;          // inner iterations per epoch (the reference's n_samples)
  unsigned max_epochs;
  double tol;
  const double* LS;     // lag_scaling table, nit + 1 entries (sparse)
  int use_lds;          // w, g_sum (and lag) staged in LDS
  int ls_cache;         // register-resident sparse kernel: lag_scaling entries kept in LDS
  int* out;             // [0] epochs run, [1] converged
};

// launchers implemented in the .hip files
int launch_sparse_exact(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                        hipStream_t st);
int launch_dense_exact(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                       hipStream_t st);
size_t sparse_exact_lds_bytes(const SagaDev& d, bool stage_state);
bool sparse_exact_k1_eligible(const SagaDev& d);
size_t sparse_exact_k1_lds_bytes(const SagaDev& d, int64_t nit, bool allow_stage, int* ls_cache, int* stage_state);
int launch_sparse_exact_k1(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, hipStream_t st);
bool sparse_exact_mc_eligible(const SagaDev& d);
size_t sparse_exact_mc_lds_bytes();
int sparse_exact_mc_wavefronts();
int launch_sparse_exact_mc(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, hipStream_t st);
size_t sparse_exact_k1m_lds_bytes(int64_t nit, int* ls_cache);
int sparse_exact_k1m_consumers();
int launch_sparse_exact_k1m(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, hipStream_t st);
size_t dense_exact_small_lds_bytes(const SagaDev& d, int64_t nit);
size_t dense_exact_small2_lds_bytes(const SagaDev& d, int penalty, int64_t nit);
int launch_dense_exact_small2(const SagaDev& d, int penalty, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                              hipStream_t st);
int dense_exact_wide_threads(const SagaDev& d);
size_t dense_exact_wide_lds_bytes(const SagaDev& d, bool stage_state);
int launch_dense_exact_wide(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, hipStream_t st);
int launch_dense_exact_small(const SagaDev& d, int penalty, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                             hipStream_t st);
size_t dense_exact_lds_bytes(const SagaDev& d, bool stage_state);

int launch_batch_gather(const SagaDev& d, LamParams* lam, int64_t t0_in_epoch, int m, int tail,
                        int batch_id_offset, hipStream_t st, hipEvent_t ev0 = nullptr,
                        hipEvent_t ev1 = nullptr);
int launch_batch_sweep(const SagaDev& d, LamParams* lam, int penalty, int tail, int m, int batch_id_offset,
                       hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, double ov_r = 0.0,
                       double ov_ls = 0.0, double ov_m = 0.0);
int launch_cw_init(const SagaDev& d, const LamParams* lam, hipStream_t st);
int batch_gather_blocks(const SagaDev& d, int m);
int64_t batch_gather_slab_doubles(const SagaDev& d, int m);
int launch_epoch_end(LamParams* lam, int batches, hipStream_t st);
// virtual shards: one launch covers the same batch of all V shards
int launch_vs_broadcast(const SagaDev& d, hipStream_t st);
int launch_vs_gather(const SagaDev& d, LamParams* lam, int64_t t0_in_epoch, int m, hipStream_t st,
                     hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int batch_index = 0);
int launch_vs_sweep(const SagaDev& d, LamParams* lam, int tail, int m, hipStream_t st, hipEvent_t ev0 = nullptr,
                    hipEvent_t ev1 = nullptr);
int launch_vs_merge(const SagaDev& d, int final_merge, hipStream_t st, LamParams* epoch_end = nullptr, int batches = 0);
int launch_vs_cw(const SagaDev& d, hipStream_t st);
bool vs_eligible(const SagaDev& d, int m);
// the whole epoch of the virtual shards in ONE launch (saga_vs_epoch_kernel)
bool vs_fused_eligible(const SagaDev& d);
size_t vs_fused_sync_words();
size_t vs_fused_sync_sticky_word();
size_t vs_fused_col_words();
size_t vs_fused_publish_doubles(const SagaDev& d, int n_shards);
size_t vs_fused_exchange_doubles(const SagaDev& d, int n_shards);
int launch_vs_epoch(const SagaDev& d, LamParams* lam, int nb, int every, hipStream_t st, hipEvent_t ev0 = nullptr,
                    hipEvent_t ev1 = nullptr);
int vs_fused_rng_workgroups(const SagaDev& d);
bool compact_eligible(const SagaDev& d);
int launch_pack_compact(const SagaDev& d, char* P, char* Q, uint32_t* meta, hipStream_t st);
int compact_entries(const SagaDev& d);
int launch_m_move(const SagaDev& d, int to_record, hipStream_t st);

int launch_convergence(const SagaDev& d, LamParams* lam, hipStream_t st);
int launch_loss(const SagaDev& d, LamParams* lam, bool sparse, hipStream_t st);
int launch_delta_export(const SagaDev& d, const double* ref, double* out, double weight, hipStream_t st);
int launch_delta_apply(const SagaDev& d, double* ref, const double* merged, double w_weight, hipStream_t st);
int batched_max_classes();
int lds_target_grid(const SagaDev& d);   // workgroups of the LDS-privatised gather forms (one per CU)
// binned form: is it the form launch_batch_gather / launch_batch_sweep would use for m draws?
bool binned_active(const SagaDev& d, int m);
int launch_col_count(const SagaDev& d, int64_t nnz, unsigned* counts, hipStream_t st);
int launch_wpad_refresh(const SagaDev& d, hipStream_t st);
int launch_range_moment(const SagaDev& d, const uint16_t* feat_range, unsigned long long* sumsq, int R, hipStream_t st);
size_t binned_max_range_features(int K);
int launch_rng_fill(const uint32_t* state_in, uint32_t* state_out, uint32_t n_samples, uint32_t* out,
                    int64_t count, hipStream_t st, int n_shards = 0, const double* shard_size = nullptr,
                    int gens = 1, int64_t run_len = 0, int convert = 1, int narrow_cus = 0, int wgs = 0);
int launch_rng_convert(uint32_t* out, int64_t count, uint32_t n_samples, hipStream_t st, int n_shards,
                       const double* shard_size, int64_t run_len, int narrow_cus = 0);

// jump-ahead of R's Mersenne-Twister (mt_jump.cpp, r_rng_device.hip)
bool mt_jump_poly(uint64_t J, uint32_t* out624);
void mt_jump_host(const sgdnet_rng* in, const uint32_t* poly624, sgdnet_rng* out);
int launch_rng_unif(const uint32_t* state_in, uint32_t* state_out, uint32_t* raw, double* out, int64_t count, hipStream_t st);
int launch_rng_jump(const uint32_t* state_in, uint32_t* state_out, const uint32_t* poly_dev, int gens,
                    hipStream_t st, int max_wgs = 0);
int rng_generators_per_workgroup();

}  // namespace sgdnet
