/*
 * sgdnet_hip.h -- C ABI of the MI355X (gfx950) SAGA elastic-net backend.
 *
 * Drop-in boundary for the hot path of jolars/sgdnet (paths below are relative
 * to the reference tree):
 *
 *   src/RcppExports.cpp:11-21,24-34  _sgdnet_SgdnetDense / _sgdnet_SgdnetSparse
 *   src/sgdnet.cpp:358-375           SgdnetDense / SgdnetSparse (x, y, control)
 *   R/sgdnet.R:346-366               the `control` list and the two call sites
 *
 * sgdnet_fit_dense / sgdnet_fit_sparse take exactly what those entry points
 * take (R's column-major matrix or dgCMatrix slots, the response matrix, the 14
 * control fields) and return exactly what src/sgdnet.cpp:275-284 returns.  The
 * R-side binding (a .Call shim over these two functions) is shown in
 * INTEGRATION.md and shim/sgdnet_shim.c.
 *
 * The sgdnet_solver_* functions expose the device-resident per-lambda SAGA
 * loop (reference src/saga-sparse.h:194-383, src/saga-dense.h:99-224) for the
 * benchmark harness, the parity tests and the multi-GPU driver.
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return 0
 * on success or a negative SGDNET_E* code; sgdnet_last_error() describes the
 * most recent failure on the calling thread.  There is no CPU fallback: every
 * compute entry point fails with SGDNET_ENODEVICE when no HIP device exists.
 */
#ifndef SGDNET_HIP_H_
#define SGDNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: sgdnet_control carries losses_sink / losses_ctx, sgdnet_set_option exists.  3: sgdnet_auc_*_rng, the option
 * exact_row_registers, sgdnet_solver_rng_layout (additions only).  4: sgdnet_control ends with n_gpus / devices
 * (a fit sharded over the GPUs of a node), sgdnet_solver_link_peers, the option fused_epoch.  A caller compiled against
 * another version must not pass its structs: the shim and the Python binding compare sgdnet_abi_version()
 * with this constant when they load the library. */
#define SGDNET_ABI_VERSION 4

/* error codes */
#define SGDNET_OK          0
#define SGDNET_EINVAL     -1   /* bad argument */
#define SGDNET_ENODEVICE  -2   /* no HIP device / HIP runtime failure at init */
#define SGDNET_EHIP       -3   /* HIP call failed */
#define SGDNET_ENOMEM     -4
#define SGDNET_EUNSUPPORTED -5 /* valid request outside what a mode implements */
#define SGDNET_ESTREAM    -6   /* explicit sample stream exhausted */

/* control$family (R/sgdnet.R:185-188; src/sgdnet.cpp:305-331) */
#define SGDNET_GAUSSIAN    0
#define SGDNET_BINOMIAL    1
#define SGDNET_MULTINOMIAL 2
#define SGDNET_MGAUSSIAN   3

/* penalty functor chosen by RunSaga (src/sgdnet.cpp:80-98; src/penalties.h) */
#define SGDNET_RIDGE       0
#define SGDNET_ELASTICNET  1
#define SGDNET_GROUPLASSO  2

/* execution mode of the SAGA loop */
#define SGDNET_MODE_EXACT   0  /* the reference iteration, one draw at a time, in stream order */
#define SGDNET_MODE_BATCHED 1  /* `batch` consecutive draws against one snapshot (sparse x: up to 64 classes;
                                  dense x: up to 16 classes, 17..64 through the sparse form with every entry stored); sgdnet_fit_*
                                  fall back to the exact iteration outside that */
#define SGDNET_MODE_AUTO    2  /* sgdnet_fit_* only: batched wherever it is implemented (see above), exact otherwise.  Same optimum, not the reference's
                                  iteration order; SGDNET_MODE_EXACT stays the default. */

/* x as R passes it to SgdnetSparse: the slots of a dgCMatrix (R/sgdnet.R:226). */
typedef struct sgdnet_csc {
  int64_t        n_rows;   /* Dim[0] = samples  */
  int64_t        n_cols;   /* Dim[1] = features */
  const int32_t* colptr;   /* @p, n_cols + 1    */
  const int32_t* rowidx;   /* @i, 0-based, ascending within a column */
  const double*  values;   /* @x                */
} sgdnet_csc;

/* One uniform(0,1) draw; the R shim passes unif_rand() so that R's own RNG is
 * advanced exactly as Rcpp::RNGScope + R::runif would (src/RcppExports.cpp:14,
 * src/saga-sparse.h:261). */
typedef double (*sgdnet_unif_fn)(void* ctx);

/* Receiver of the per-epoch debug losses of one lambda (control.debug, options(sgdnet.debug = TRUE),
 * src/saga-sparse.h:350-365 / src/utils.h:199-227): called once per lambda, on the calling thread,
 * when that lambda's SAGA loop has ended, with the EpochLoss value of every epoch it ran.  The
 * reference grows a std::vector (src/saga-sparse.h:364); a sink lets the caller do the same instead
 * of reserving n_lambda x max_iter doubles (80 GB per path at the reference's benchmark setting
 * maxit = 1e8, data-raw/benchmarks.R). */
typedef void (*sgdnet_losses_fn)(void* ctx, int lambda_index, const double* losses, int count);

/* State of R's default generator (Mersenne-Twister): 625 words = mti + mt[624]. */
typedef struct sgdnet_rng { uint32_t mti; uint32_t mt[624]; } sgdnet_rng;

typedef struct sgdnet_control {
  /* ---- the control list of R/sgdnet.R:346-359, field for field ---- */
  int           debug;
  double        elasticnet_mix;
  int           family;               /* SGDNET_GAUSSIAN ... */
  int           intercept;
  int           is_sparse;            /* unused natively, as in the reference */
  const double* lambda;               /* NULL / n_lambda_user == 0 => automatic path */
  int           n_lambda_user;        /* length(control$lambda) */
  double        lambda_min_ratio;
  unsigned      max_iter;
  int           n_lambda;
  int           n_classes;
  int           standardize;
  int           standardize_response;
  double        tol;
  int           type_multinomial;     /* 0 = "ungrouped" (the only value R sends), 1 = "grouped" */

  /* ---- sample order (exactly one draw per inner iteration) ---- */
  const uint32_t* sample_stream;      /* explicit indices in [0, n), or NULL */
  int64_t         sample_stream_len;
  sgdnet_unif_fn  unif;               /* used when sample_stream == NULL and unif != NULL */
  void*           unif_ctx;
  uint32_t        seed;               /* else: built-in R-compatible Mersenne-Twister, set.seed(seed) */
  struct sgdnet_rng* rng_state;       /* else-branch only: if non-NULL the generator starts from this
                                         state (R's .Random.seed: mti + mt[624]) instead of `seed`
                                         and the advanced state is written back */

  /* ---- backend extensions; zero-initialised == reference behaviour ---- */
  int           mode;                 /* SGDNET_MODE_* */
  int64_t       batch;                /* batched mode: staleness window; 0 = automatic */
  int           device;               /* HIP device ordinal */
  sgdnet_losses_fn losses_sink;       /* debug: receives each lambda's losses (result.losses may then be NULL) */
  void*         losses_ctx;
  /* ---- ABI 4: the fit sharded over several GPUs of one node (R: options(sgdnet.gpus = N)) ----
   * n_gpus <= 1: one GPU, `device`.  n_gpus = 2..8: the samples are cut into n_gpus contiguous ranges, one solver per
   * GPU (`devices[0 .. n_gpus)`, or device, device + 1, ... when devices is NULL), each with its own virtual shards;
   * all replicas are averaged periodically inside the GPUs' epoch kernels by direct peer loads
   * (sgdnet_solver_link_peers).  Batched iteration (mode = batched / auto) of sparse x with one response, the
   * built-in generator, an even number of features; anything else returns SGDNET_EUNSUPPORTED.  The same device may
   * be named more than once (the ranks then share its CUs: rehearsals on one GPU). */
  int           n_gpus;
  const int*    devices;
} sgdnet_control;

/* Caller-allocated mirror of the list returned by src/sgdnet.cpp:275-284. */
typedef struct sgdnet_result {
  double*  a0;            /* n_classes * n_lambda            (list of numeric(K))        */
  double*  beta;          /* n_classes * n_features * n_lambda (K fastest, as unlist())  */
  double*  lambda;        /* n_lambda                                                    */
  double*  dev_ratio;     /* n_lambda                                                    */
  double*  return_codes;  /* n_lambda, 0 converged / 1 max_iter reached                  */
  double*  losses;        /* n_lambda * max_iter or NULL; filled when control.debug (or use
                             control.losses_sink, which needs no max_iter-sized block)   */
  int32_t* losses_len;    /* n_lambda or NULL                                            */
  double   nulldev;
  double   npasses;
  int64_t  draws_used;    /* inner iterations executed == uniform draws consumed         */
} sgdnet_result;

int sgdnet_abi_version(void);
const char* sgdnet_last_error(void);
int sgdnet_device_count(void);

/* ------------------------------------------------------------------------ */
/* Process-wide backend options.  These are the ONLY switches that change    */
/* how sgdnet_fit_* runs a fit; the library reads no environment variable    */
/* for that (SGDNET_TRACE prints progress to stderr and changes nothing;     */
/* kernel A/B switches exist only in -DSGDNET_EXPERIMENTS builds).           */
/*   "virtual_shards"     -1 (default): the driver's rule -- batched fits of */
/*                        up to 16 classes or responses with                 */
/*                        >= 200 000 samples run as up to 8 averaged         */
/*                        replicas; 0 or 1: never;             */
/*                        2..8: that many wherever the kernels allow it      */
/*   "rng_generators"     0 (default): 8..32 generators side by side on R's  */
/*                        one stream for epochs of >= 200 000 draws, else 1; */
/*                        1..64: that many                                   */
/*   "window_eigenvalue"  1 (default): the automatic window of sparse x uses */
/*                        the largest eigenvalue of X'X/n (power iteration); */
/*                        0: its diagonal bound only                         */
/*   "host_setup"         0 (default): standardisation, lambda_max, transpose*/
/*                        and packing run on the device for sparse x and for */
/*                        dense x of >= 4e6 elements; 1: host loops          */
/*   "exact_epoch_blocks" 1 (default): exact mode with the built-in generator*/
/*                        runs several epochs per launch; 0: one per launch  */
/*   "exact_row_registers" 1 (default): exact mode on sparse x with one      */
/*                        response and explicit x keeps the drawn row's      */
/*                        coefficients in registers for the whole draw and   */
/*                        requests the next draw's a draw ahead (same bits,  */
/*                        DESIGN.md 4.1), with several consumer wavefronts   */
/*                        where draws seldom share a feature (several classes:*/
/*                        the general iteration on eight wavefronts); 0: the general */
/*                        kernel; 2: one consumer, state kept in memory; 3:  */
/*                        several consumers wherever that is legal; 4: one   */
/*                        consumer always                                    */
/*   "fused_epoch"        1 (default): a batched epoch on virtual shards      */
/*                        (sparse x, one response) is ONE launch whose        */
/*                        workgroups synchronise shard by shard (a shard whose */
/*                        workgroups share an XCD hands off through its L2);  */
/*                        2: the same with write-through hand-offs always; 0: one */
/*                        gather and one sweep launch per batch (what the     */
/*                        library falls back to by itself when the GPU is     */
/*                        shared and the launch cannot become resident)       */
/* Unknown names and out-of-range values return SGDNET_EINVAL.  Options are  */
/* read when a fit starts (exact_row_registers and fused_epoch: whenever     */
/* sgdnet_solver_run / _enqueue_epochs is called, i.e. once per epoch block  */
/* of a fit; the results do not depend on either); change them between fits. */
/* ------------------------------------------------------------------------ */
int sgdnet_set_option(const char* name, int value);
int sgdnet_get_option(const char* name, int* value);

/* replaces _sgdnet_SgdnetSparse (src/RcppExports.cpp:24-34 -> src/sgdnet.cpp:369-375) */
int sgdnet_fit_sparse(const sgdnet_csc* x, const double* y, int y_cols,
                      const sgdnet_control* control, sgdnet_result* out);

/* replaces _sgdnet_SgdnetDense (src/RcppExports.cpp:11-21 -> src/sgdnet.cpp:359-365);
 * x is n_samples x n_features column-major (an R numeric matrix). */
int sgdnet_fit_dense(const double* x, int64_t n_samples, int64_t n_features,
                     const double* y, int y_cols,
                     const sgdnet_control* control, sgdnet_result* out);

/* ------------------------------------------------------------------------ */
/* Built-in R-compatible Mersenne-Twister (the generator behind R::runif at  */
/* src/saga-sparse.h:261 under R's default RNGkind): 625 words = mti + mt[]. */
/* ------------------------------------------------------------------------ */
void     sgdnet_rng_seed(sgdnet_rng* r, uint32_t seed);           /* set.seed(seed)        */
double   sgdnet_rng_unif(sgdnet_rng* r);                          /* unif_rand()           */
/* Jump-ahead on the same stream (sgdnet_amd/csrc/mt_jump.cpp): poly624 <- x^draws mod phi(x), the
 * GF(2) polynomial that moves a generator state `draws` unif_rand() calls down R's stream; and its
 * application on the host.  The fit driver uses the device form to run several generators on ONE
 * set.seed() stream.  sgdnet_rng_jump_poly returns SGDNET_OK or SGDNET_EUNSUPPORTED. */
int      sgdnet_rng_jump_poly(uint64_t draws, uint32_t* poly624);
void     sgdnet_rng_jump(const sgdnet_rng* in, const uint32_t* poly624, sgdnet_rng* out);
void     sgdnet_rng_fill(sgdnet_rng* r, uint32_t n_samples,       /* floor(runif(0, n))    */
                         uint32_t* out, int64_t count);

/* ------------------------------------------------------------------------ */
/* Device-resident SAGA solver: state (w, intercept, g_memory, g_sum,        */
/* g_sum_intercept) persists across calls, i.e. across the lambda path       */
/* (src/sgdnet.cpp:187-198, 217-244).                                        */
/* ------------------------------------------------------------------------ */
typedef struct sgdnet_solver sgdnet_solver;

typedef struct sgdnet_problem {
  int      family;
  int      n_classes;        /* K */
  int64_t  n_samples;        /* samples resident on this device */
  int64_t  n_total;          /* samples of the whole job (1/n of the gradient average); 0 => n_samples */
  int64_t  n_features;
  int      fit_intercept;
  int      standardize;      /* sparse: implicit centring with x_center_scaled */
  /* sample-major data, i.e. the matrix after AdaptiveTranspose (src/sgdnet.cpp:177):
   * sparse: column i = sample i (rowptr n+1, feature ids ascending, values);
   * dense:  x_dense is n_features x n_samples column-major. */
  const int64_t* rowptr;
  const int32_t* colidx;
  const double*  values;
  const double*  x_dense;
  const double*  x_center_scaled;  /* n_features or NULL */
  const double*  y;                /* y_rows x n_samples column-major (src/sgdnet.cpp:178) */
  int            y_rows;
  int            device;
} sgdnet_problem;

int  sgdnet_solver_create(const sgdnet_problem* prob, sgdnet_solver** out);
void sgdnet_solver_destroy(sgdnet_solver* s);

/* Saga(...) arguments that change per lambda: penalty functor, step size gamma,
 * L2 strength alpha, L1 strength beta (src/saga-sparse.h:196,206-208). */
int sgdnet_solver_set_penalty(sgdnet_solver* s, int penalty, double gamma, double alpha, double beta);

/* Copy a state array between host and device.  which: 0 w (K x p), 1 intercept (K),
 * 2 g_memory (K x n), 3 g_sum (K x p), 4 g_sum_intercept (K). */
int sgdnet_solver_get_state(sgdnet_solver* s, int which, double* host);
int sgdnet_solver_set_state(sgdnet_solver* s, int which, const double* host);

/* Generate `count` sample indices floor(runif(0, n_samples)) ON THE DEVICE from the R-compatible
 * Mersenne-Twister state `rng` (replaces any previous stream); `rng` is advanced exactly as
 * sgdnet_rng_fill(rng, n_samples, out, count) would advance it. */
int sgdnet_solver_generate_stream(sgdnet_solver* s, sgdnet_rng* rng, int64_t count);

/* Copy `count` entries of the resident stream, starting at `offset`, back to the host. */
int sgdnet_solver_get_stream(sgdnet_solver* s, uint32_t* host, int64_t offset, int64_t count);

/* Make `count` sample indices resident on the device (replaces any previous stream). */
int sgdnet_solver_upload_stream(sgdnet_solver* s, const uint32_t* host, int64_t count);

/* Run the SAGA loop of one lambda on the resident stream starting at `stream_offset`:
 * epochs of `draws_per_epoch` inner iterations until ConvergenceCheck
 * (src/utils.h:240-262) passes or max_epochs is reached.  Blocks until done.
 * losses (max_epochs doubles) may be NULL. */
int sgdnet_solver_run(sgdnet_solver* s, int mode, int64_t batch,
                      int64_t stream_offset, int64_t draws_per_epoch,
                      unsigned max_epochs, double tol,
                      unsigned* epochs_run, int* converged, double* losses);

/* Enqueue `n_epochs` batched epochs without convergence checks or host
 * synchronisation (benchmark stepping).  sgdnet_solver_sync waits for them. */
int sgdnet_solver_enqueue_epochs(sgdnet_solver* s, int64_t batch, int64_t stream_offset,
                                 int64_t draws_per_epoch, int n_epochs);
int sgdnet_solver_sync(sgdnet_solver* s);

/* Sample order produced on the device, one epoch ahead of the epoch that consumes it, on a side
 * stream (what sgdnet_fit_* does for the built-in generator).  generators > 1 runs that many
 * Mersenne-Twisters side by side ON THE ONE STREAM `rng` defines (jump-ahead, see
 * sgdnet_rng_jump_poly): the draws are those of a single generator whatever the count.
 *   rng_open(s, &rng, n, G); for every epoch { rng_next(s, &off); run / enqueue an epoch with
 *   stream_offset = off; rng_done(s); }  rng_close(s, &rng) -> rng = the state after exactly the
 *   epochs that were consumed (a speculative generation is discarded).
 * With virtual shards the draws come in the layout sgdnet_solver_set_virtual_shards describes;
 * sgdnet_solver_rng_layout (before rng_open) cuts an epoch into runs of draws_per_run draws -- what a
 * sample-sharded rank does between two merges with the other ranks -- each laid out shard after
 * shard, the last one shorter if the epoch does not divide (0: one run = the epoch). */
int sgdnet_solver_rng_layout(sgdnet_solver* s, int64_t draws_per_run);
int sgdnet_solver_rng_open(sgdnet_solver* s, sgdnet_rng* rng, int64_t draws_per_epoch, int generators);
int sgdnet_solver_rng_next(sgdnet_solver* s, int64_t* stream_offset);
int sgdnet_solver_rng_done(sgdnet_solver* s);
int sgdnet_solver_rng_close(sgdnet_solver* s, sgdnet_rng* rng);

/* One batched epoch launched eagerly with HIP events around every gather
 * kernel launch; returns summed kernel time and launch count. */
int sgdnet_solver_profile_epoch(sgdnet_solver* s, int64_t batch, int64_t stream_offset,
                                int64_t draws_per_epoch, double* gather_ms, int* gather_launches,
                                double* sweep_ms, int* sweep_launches);

/* Dispatch timing of the fused epoch launches (gather form 3) enqueued from now on: enable != 0 starts collecting
 * start / stop events bound to every such launch on the solver's stream; every call first synchronises and returns
 * what was collected since the previous call (summed kernel milliseconds, launches) -- the benchmark brackets
 * exactly its timed region with two calls.  Either pointer may be NULL. */
int sgdnet_solver_epoch_timing(sgdnet_solver* s, int enable, double* sum_ms, int* launches);

/* Which gather kernel a batch of `batch` draws uses: 0 = saga_batch_gather_kernel (global
 * atomics), 1 = saga_batch_gather_lds_kernel (LDS-privatised scatter), 2 = the binned form
 * (saga_binned_gather_kernel + saga_binned_sweep_kernel: K x p tables that fit no LDS; valid after a
 * run / enqueue with that batch has sized its scratch), 3 = the fused epoch of the virtual shards
 * (saga_vs_epoch_kernel: gathers, sweeps and merges of a whole epoch in one launch). */
int sgdnet_solver_gather_form(const sgdnet_solver* s, int64_t batch);

/* Sample-sharded solvers of ONE process, one per GPU of a node: after this call the periodic
 * replica average of every solver's batched epochs runs over the virtual shards of ALL of them -- inside the
 * fused epoch kernel, by direct loads from the other GPUs' exchange buffers (hipDeviceEnablePeerAccess) behind
 * counters the ranks add to remotely; no collective library, no launch per merge.  Rank q holds a contiguous
 * range of the samples; every solver needs the same number of virtual shards, and the epochs of all ranks must be
 * enqueued (sgdnet_solver_enqueue_epochs) before any of them is waited for.  n == 1 unlinks.
 * sgdnet_fit_sparse does all of this itself when control.n_gpus > 1. */
int sgdnet_solver_link_peers(sgdnet_solver** solvers, int n);
/* The same link between solvers of DIFFERENT processes (one process per GPU): every rank fills
 * sgdnet_solver_peer_info_bytes() bytes with sgdnet_solver_peer_info (hipIpc handles of its exchange buffer and counters,
 * its shard sizes), the caller gathers the ranks' blocks in rank order (any transport: MPI, torch.distributed, a pipe) and
 * every rank calls sgdnet_solver_link_ipc with all of them; a barrier of the caller's must separate the last link from the
 * first epoch.  The peers' buffers are mapped with hipIpcOpenMemHandle and unmapped when the solver is destroyed. */
int sgdnet_solver_peer_info_bytes(void);
int sgdnet_solver_peer_info(sgdnet_solver* s, void* out);
int sgdnet_solver_link_ipc(sgdnet_solver* s, int rank, int n_ranks, const void* infos);
/* CUs a solver's batched launches may fill (0: the device's): linked solvers that share ONE GPU -- tests, or two
 * fits side by side -- must fit their persistent launches side by side. */
int sgdnet_solver_set_cu_budget(sgdnet_solver* s, int cus);

/* 2 * sum_i Loss_i (src/utils.h:304-329) over the resident samples. */
int sgdnet_solver_deviance(sgdnet_solver* s, double* out);

/* Multi-GPU merge: buffers hold 2*K*p + 2*K doubles in DEVICE memory:
 * [g_sum - g_sum_ref | w - w_ref | g_sum_intercept - ref | intercept - ref].
 * snapshot() records the reference point before a local epoch; export_delta() writes
 * the deltas; apply_merged() sets state = ref + {1, w_weight, 1, w_weight} * merged. */
int sgdnet_solver_snapshot(sgdnet_solver* s);
int sgdnet_solver_export_delta(sgdnet_solver* s, void* device_buf);
int sgdnet_solver_apply_merged(sgdnet_solver* s, const void* device_buf, double w_weight);
int64_t sgdnet_solver_delta_len(const sgdnet_solver* s);

/* The solver's HIP stream (a hipStream_t), so that a caller can order its own device work
 * -- e.g. the RCCL all-reduce of the merge buffer -- after the solver's kernels without host
 * synchronisation.  export_delta_async / apply_merged_async only enqueue. */
void* sgdnet_solver_stream(sgdnet_solver* s);
int sgdnet_solver_export_delta_async(sgdnet_solver* s, void* device_buf);
/* same, every delta multiplied by `weight` (the shard's share n_local / n_total) */
int sgdnet_solver_export_delta_weighted_async(sgdnet_solver* s, void* device_buf, double weight);
/* state <- snapshot + merged (coefficients: + w_weight * merged); the result also becomes the
 * snapshot of the next local run */
int sgdnet_solver_apply_merged_async(sgdnet_solver* s, const void* device_buf, double w_weight);

/* ------------------------------------------------------------------------ */
/* Synchronous sample-sharded batches: a global batch is split  */
/* across the ranks; every rank gathers its share into the bound buffer       */
/* [D (K*p) | intercept-accumulator slots (2*256*K)], the caller sums the     */
/* buffer across ranks on sgdnet_solver_stream() (RCCL all-reduce), then      */
/* every rank sweeps with the GLOBAL draw count.  The iterates are those of   */
/* the single-GPU batched mode with batch = that global count.                */
/* ------------------------------------------------------------------------ */
int64_t sgdnet_solver_sync_buffer_len(const sgdnet_solver* s);            /* doubles */
int sgdnet_solver_sync_bind(sgdnet_solver* s, void* device_buf);         /* NULL: unbind */
int sgdnet_solver_sync_begin(sgdnet_solver* s, int64_t stream_offset, int64_t draws_local_per_epoch);
int sgdnet_solver_sync_gather(sgdnet_solver* s, int64_t t0_local, int64_t m_local, int round);
int sgdnet_solver_sync_sweep(sgdnet_solver* s, int64_t m_global, int64_t m_local, int round);
int sgdnet_solver_sync_end(sgdnet_solver* s, int rounds);

/* Virtual shards (batched mode; one response: sparse x with or without implicit centring, or dense x;
 * 2..16 classes or responses: sparse or dense x whose n_classes x n_features table fits the LDS): the samples are
 * split into n_shards contiguous ranges (sizes as shard_bounds of sgdnet_amd/parallel.py), every
 * range runs the batched iteration on its own replica of (w, g_sum, intercept) with local
 * normalisation, one launch carries the same batch of all shards, and the replicas are averaged
 * on the device every n / 32 draws per shard.  An epoch of `draws` consumes draws / n_shards
 * stream entries per shard, laid out shard after shard: entry t of shard v of the epoch at
 * stream offset o is stream[o + v * (draws / n_shards) + t] and must be a sample of shard v.
 * sgdnet_solver_generate_stream produces that layout.  0 or 1 switches it off. */
int sgdnet_solver_set_virtual_shards(sgdnet_solver* s, int n_shards);
/* draws per shard between two device-side averages (0: n_samples / 32).  A multi-GPU job sets
 * n_total / 32 of the whole job and merges across GPUs at the same cadence. */
int sgdnet_solver_set_merge_period(sgdnet_solver* s, int64_t draws_per_shard);

/* The n that g_sum increments are divided by (default: n_total of the problem description).
 * A sample-sharded job sets it to the shard size (local normalisation) or to the
 * job's sample count (synchronous mode). */
int sgdnet_solver_set_n_total(sgdnet_solver* s, int64_t n_total);

/* ConvergenceCheck on the current w against the previous call's w (device-side). */
int sgdnet_solver_convergence(sgdnet_solver* s, double tol, int* converged);

/* max|w - w_prev| and max|w| of the most recent ConvergenceCheck (src/utils.h:248-249). */
int sgdnet_solver_last_change(const sgdnet_solver* s, double* max_change, double* max_size);

/* ---- prediction and prediction error along the lambda path ----
 * Device versions of what the reference does on the host in R: predict.sgdnet
 * (R/predict.sgdnet.R:347-402, type = "link": cbind2(1, newx) %*% beta for every lambda) and
 * score.sgdnet_<family> (R/score.R:55-186), which cv_sgdnet (R/cv_sgdnet.R:161-199) calls once per
 * (alpha, fold).  x is SAMPLE-major: CSR (rowptr int64[n+1], colidx int32, values) or dense
 * row-major n x p.  a0 is K x n_lambda and beta K x p x n_lambda, K fastest, exactly as
 * sgdnet_result holds them.  y as in sgdnet_fit_*: y_rows x n (class index for binomial /
 * multinomial).  out: one value per lambda; link: n x n_lambda x K (K fastest). */
#define SGDNET_MEASURE_DEVIANCE 0
#define SGDNET_MEASURE_MSE      1
#define SGDNET_MEASURE_MAE      2
#define SGDNET_MEASURE_CLASS    3   /* binomial / multinomial */
int sgdnet_score_sparse(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                        const double* y, int y_rows, int family, int n_classes, const double* a0,
                        const double* beta, int n_lambda, int measure, int device, double* out);
int sgdnet_score_dense(const double* x, int64_t n, int64_t p, const double* y, int y_rows, int family,
                       int n_classes, const double* a0, const double* beta, int n_lambda, int measure, int device,
                       double* out);
/* type.measure = "auc" of score.sgdnet_binomial (R/score.R:98-99; auc() :203-233, the weighted branch the
 * two-column indicator matrix selects).  y: class codes 0 / 1.  tie: the reference orders equal probabilities
 * by stats::runif(2n) drawn per lambda -- pass those draws, 2n per lambda (entry i for a class-0 sample i,
 * entry n + i for a class-1 sample), or NULL to break ties by position in the stacked vector (class-0 entries
 * first, then sample order).  out: one area per lambda. */
#define SGDNET_MEASURE_AUC      4   /* binomial; through sgdnet_score_*: no tie vector */
int sgdnet_auc_sparse(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                      const double* y, const double* a0, const double* beta, int n_lambda, const double* tie,
                      int device, double* out);
int sgdnet_auc_dense(const double* x, int64_t n, int64_t p, const double* y, const double* a0, const double* beta,
                     int n_lambda, const double* tie, int device, double* out);
/* The same with the tie breakers drawn on the device: stats::runif(2n) per lambda, in lambda order, from *rng, which
 * is left where 2 n n_lambda calls of unif_rand() leave R's generator (round 3; before, the caller drew them). */
int sgdnet_auc_sparse_rng(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                          const double* y, const double* a0, const double* beta, int n_lambda, sgdnet_rng* rng,
                          int device, double* out);
int sgdnet_auc_dense_rng(const double* x, int64_t n, int64_t p, const double* y, const double* a0, const double* beta,
                         int n_lambda, sgdnet_rng* rng, int device, double* out);
int sgdnet_predict_sparse(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx,
                          const double* values, int n_classes, const double* a0, const double* beta,
                          int n_lambda, int device, double* link);
int sgdnet_predict_dense(const double* x, int64_t n, int64_t p, int n_classes, const double* a0, const double* beta,
                         int n_lambda, int device, double* link);

/* Default staleness window of the batched mode: about 2 * L_max / L_F, where L_max is the
 * largest squared sample norm and L_F is bounded below by the largest mean squared feature
 * value (the diagonal of X'X/n); clamped to [64, 131072].  DESIGN.md "Choosing the batch".
 * sgdnet_fit_* apply the same rule with the largest eigenvalue of X'X/n where it is cheap (dense x: power
 * iteration, plus the constant feature), let it go down to 8 draws, and under SGDNET_MODE_AUTO take the exact
 * iteration when the rule asks for less than that or an epoch would be more than 16384 batches. */
int64_t sgdnet_auto_batch(double max_sample_sqnorm, double max_feature_mean_sq);

/* The window of a fit with virtual shards: a shard's epoch of `draws_per_shard` draws is ceil(draws_per_shard / window)
 * rounds, each with a fixed hand-off cost inside the epoch kernel.  When `window` (sgdnet_auto_batch's, capped) leaves a
 * short last round and a window at most 1/8 longer saves that round, the longer one is returned (config 4: 1 250 000 draws
 * per shard, 131 072 -> 138 889: nine rounds instead of ten); `window` otherwise.  sgdnet_fit_* apply it to their automatic
 * window (the rule of sgdnet_auto_batch keeps a factor 3 to the unstable regime; 1/8 of it is spent here). */
int64_t sgdnet_shard_window(int64_t window, int64_t draws_per_shard);

#ifdef __cplusplus
}
#endif
#endif /* SGDNET_HIP_H_ */
