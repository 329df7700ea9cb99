/*
 * sgdnet_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded plain-C restatement of the SAGA elastic-net path of
 * jolars/sgdnet (reference tree: /root/reference, read as text only).  Used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker / timed CPU baseline.  Nothing under sgdnet_amd/ may link, import or
 * call this.
 *
 * Parity status: PINNED to outputs of a real build of the reference.  The
 * reference cannot be compiled here (needs R, Rcpp and Eigen, none present),
 * but its tree ships the console output of its documented examples
 * (docs/reference/(any).html, pkgdown 1.1.0 / R 3.5).  tests/test_refdocs_oracle.py
 * replays those examples through this restatement -- R's Mersenne-Twister,
 * rnorm and sample() included, whole cross-validation loops whose later
 * numbers depend on every earlier fit having run exactly the reference's
 * number of epochs -- and reproduces what the reference printed at print
 * precision: 300 gaussian lasso coefficients (8 decimals), 54 binomial ridge
 * link predictions after 16 fits (10 decimals), 100 multinomial deviances (9
 * significant digits), a multinomial CV score (7 digits), CV summary
 * statistics of a gaussian fit (7 digits), 104 predicted classes, and the
 * active sets of a 100-lambda mgaussian group-lasso path (all 100 once the
 * first lambda is raised by 1e-13: at lambda_max itself the group-lasso
 * threshold test is decided by the last bits of an Eigen GEMM).  Fixtures:
 * tests/golden/refdocs.npz (tests/golden/make_refdocs_fixtures.py).
 * Beyond that it is checked against the reference's own known-answer tests
 * (closed-form ridge, OLS, logistic MLE, lambda_max formulas, null deviances,
 * sparse==dense, ...; tests/test_oracle_properties.py).
 * A second build, liboracle_det.so (-DORC_DET_MATH), replaces libm's exp/log in the family
 * gradients by the plain-IEEE functions of include/sgdnet_detmath.h, which the HIP exact
 * kernels use as well: that build and those kernels agree bit for bit
 * (tests/test_gpu_bitwise.py), and it reproduces the printed numbers like the libm build.
 * Not pinned: bit-level equality of intermediate state (Eigen's packetised
 * reductions, the libm of the machine that built the docs).
 */
#ifndef SGDNET_ORACLE_H_
#define SGDNET_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_GAUSSIAN = 0, ORC_BINOMIAL = 1, ORC_MULTINOMIAL = 2, ORC_MGAUSSIAN = 3 };
enum { ORC_RIDGE = 0, ORC_ELASTICNET = 1, ORC_GROUPLASSO = 2 };

/* R-compatible Mersenne-Twister. */
typedef struct {
  uint32_t mt[624];
  int      mti;
} orc_rng;

void     orc_rng_seed(orc_rng* r, uint32_t seed);       /* == R set.seed(seed) */
double   orc_unif_rand(orc_rng* r);                     /* == R unif_rand()    */
uint32_t orc_draw(orc_rng* r, uint32_t n_samples);      /* floor(R::runif(0,n)) */
void     orc_fill_stream(orc_rng* r, uint32_t n_samples, uint32_t* out, int64_t count);

/* Source of sample indices: explicit stream (if stream != NULL) else rng. */
typedef struct {
  const uint32_t* stream;
  int64_t         pos;
  int64_t         len;
  orc_rng*        rng;
} orc_draws;

typedef struct {
  int      family;        /* ORC_* family */
  int      penalty;       /* ORC_* penalty */
  int      n_classes;     /* K */
  int64_t  n_samples;
  int64_t  n_features;
  int      fit_intercept;
  int      standardize;   /* sparse only: implicit centring with x_center_scaled */
  double   gamma;         /* step size */
  double   alpha;         /* L2 strength  (reference's `alpha` arg of Saga) */
  double   beta;          /* L1 strength  (reference's `beta` arg of Saga)  */
  unsigned max_iter;
  double   tol;
  int      debug;
  int64_t  n_total;       /* batched oracle only: samples of the whole (sharded) job that the
                             gradient average divides by; 0 => n_samples */
  int      dense_intercept; /* batched oracle only: x is a dense matrix stored with every entry -- the intercept
                               step of saga-dense.h:170-173 (no 0.01 decay) instead of saga-sparse.h:300-304 */
} orc_saga_params;

/* Sparse SAGA, sample-major CSC (column i = sample i): reference
 * src/saga-sparse.h:194-383.  y is Ky x n column-major.  State arrays are
 * updated in place (warm start).  losses (if debug) must hold max_iter doubles.
 * Returns epochs run; *return_code per saga-sparse.h:376-382. */
unsigned orc_saga_sparse(const orc_saga_params* P,
                         const int64_t* ptr, const int32_t* idx, const double* val,
                         const double* x_center_scaled,
                         const double* y, int Ky,
                         double* intercept, double* w,
                         double* g_memory, double* g_sum, double* g_sum_intercept,
                         orc_draws* draws, unsigned* return_code,
                         double* losses);

/* Dense SAGA, x is p x n column-major (sample i contiguous): reference
 * src/saga-dense.h:99-224. */
unsigned orc_saga_dense(const orc_saga_params* P,
                        const double* x,
                        const double* y, int Ky,
                        double* intercept, double* w,
                        double* g_memory, double* g_sum, double* g_sum_intercept,
                        orc_draws* draws, unsigned* return_code,
                        double* losses);

/* B-stale ("batched") sparse SAGA: the product's throughput mode restated on
 * the CPU (DESIGN.md "Batched mode").  Not a reference function: it is the
 * reference iteration (saga-sparse.h:258-337) applied to `batch` consecutive
 * draws against one snapshot of (w, intercept); batch == 1 is mathematically
 * the reference iteration in unscaled coordinates. */
unsigned orc_saga_sparse_batched(const orc_saga_params* P, int64_t batch,
                                 const int64_t* ptr, const int32_t* idx, const double* val,
                                 const double* x_center_scaled,
                                 const double* y, int Ky,
                                 double* intercept, double* w,
                                 double* g_memory, double* g_sum, double* g_sum_intercept,
                                 orc_draws* draws, unsigned* return_code,
                                 double* losses);

/* The two halves of one batch of the batched mode (gather: draws -> D, d0, g_memory;
 * sweep: D, d0 -> w, g_sum, intercept), for the sample-sharded synchronous restatement. */
void orc_batch_gather(const orc_saga_params* P, const int64_t* ptr, const int32_t* idx,
                      const double* val, const double* x_center_scaled, const double* y, int Ky,
                      const double* intercept, const double* w, double* g_memory,
                      const uint32_t* draws, int64_t m, double* D, double* d0);
void orc_batch_sweep(const orc_saga_params* P, int64_t m_global, const double* x_center_scaled,
                     double* D, double* d0, double* intercept, double* w, double* g_sum,
                     double* g_sum_intercept);

/* r^m and LS_m = sum_{k<m} r^k for r = 1 - alpha*gamma (closed form of the
 * reference's lag_scaling table, saga-sparse.h:229-240). */
void orc_batch_factors(double alpha, double gamma, int64_t m, double* r_m, double* ls_m);

/* ---- path driver: reference src/sgdnet.cpp:119-285 ---- */
typedef struct {
  int      family;
  double   elasticnet_mix;
  int      fit_intercept;
  int      standardize;
  int      standardize_response;
  int      n_classes;
  int      n_lambda;
  const double* lambda;      /* NULL or n_lambda user values */
  double   lambda_min_ratio;
  unsigned max_iter;
  double   tol;
  int      debug;
  int64_t  batch;            /* 0/1 = exact reference iteration; >1 = batched (sparse only) */
} orc_control;

typedef struct {
  double*  a0;           /* K * n_lambda */
  double*  beta;         /* K * p * n_lambda (K fastest, then feature, then lambda) */
  double*  lambda;       /* n_lambda */
  double*  dev_ratio;    /* n_lambda */
  double*  return_codes; /* n_lambda */
  double*  losses;       /* n_lambda * max_iter or NULL */
  int*     losses_len;   /* n_lambda or NULL */
  double   nulldev;
  double   npasses;
  /* diagnostics (not part of the reference's return list) */
  double*  step_size;    /* n_lambda or NULL */
  double*  alpha_l2;     /* n_lambda or NULL */
  double*  beta_l1;      /* n_lambda or NULL */
} orc_result;

/* x: feature-major CSC as R's dgCMatrix (n x p): colptr[p+1], rowidx, val. */
int orc_fit_sparse(int64_t n, int64_t p, const int32_t* colptr, const int32_t* rowidx,
                   const double* val, const double* y, int y_cols,
                   const orc_control* ctl, orc_draws* draws, orc_result* out);

/* x: dense n x p column-major (R matrix). */
int orc_fit_dense(int64_t n, int64_t p, const double* x, const double* y, int y_cols,
                  const orc_control* ctl, orc_draws* draws, orc_result* out);

#ifdef __cplusplus
}
#endif
#endif
