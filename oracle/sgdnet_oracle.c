/*
 * sgdnet_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see sgdnet_oracle.h).
 *
 * Plain-C, single-threaded restatement of the reference's SAGA path.  Every
 * function cites the reference file:line it follows (paths relative to
 * /root/reference).  Arithmetic order follows SURVEY.md Appendix A; build with
 * -O2 -ffp-contract=off (R's default flags emit no FMA on x86-64).
 *
 * Parity: pinned by the reference's known-answer tests only; bitwise parity
 * with a real build of the reference is UNPINNED (header of sgdnet_oracle.h).
 */
#include "sgdnet_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/constants.h:22 */
static const double ORC_SMALL = 100.0 * DBL_EPSILON;

/* ------------------------------------------------------------------------ */
/* R's default RNG (Mersenne-Twister) -- R's RNG.c, restated from the        */
/* published MT19937 algorithm + R's documented set.seed scrambling          */
/* (SURVEY.md Appendix B).  Call sites: src/saga-sparse.h:261,               */
/* src/saga-dense.h:152.                                                     */
/* ------------------------------------------------------------------------ */
void orc_rng_seed(orc_rng* r, uint32_t seed) {
  int j;
  for (j = 0; j < 50; ++j) seed = 69069u * seed + 1u;
  /* i_seed[0] is mti (forced to 624 by FixupSeeds), i_seed[1..624] is mt */
  seed = 69069u * seed + 1u;
  for (j = 0; j < 624; ++j) {
    seed = 69069u * seed + 1u;
    r->mt[j] = seed;
  }
  r->mti = 624;
}

static uint32_t mt_next(orc_rng* r) {
  static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
  uint32_t y;
  if (r->mti >= 624) {
    int kk;
    for (kk = 0; kk < 624 - 397; ++kk) {
      y = (r->mt[kk] & 0x80000000u) | (r->mt[kk + 1] & 0x7fffffffu);
      r->mt[kk] = r->mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
    }
    for (; kk < 623; ++kk) {
      y = (r->mt[kk] & 0x80000000u) | (r->mt[kk + 1] & 0x7fffffffu);
      r->mt[kk] = r->mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
    }
    y = (r->mt[623] & 0x80000000u) | (r->mt[0] & 0x7fffffffu);
    r->mt[623] = r->mt[396] ^ (y >> 1) ^ mag01[y & 1u];
    r->mti = 0;
  }
  y = r->mt[r->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

double orc_unif_rand(orc_rng* r) {
  const double i2_32m1 = 2.328306437080797e-10;
  double x = (double)mt_next(r) * 2.3283064365386963e-10;
  if (x <= 0.0) return 0.5 * i2_32m1;
  if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * i2_32m1;
  return x;
}

/* unsigned s_ind = floor(R::runif(0.0, n_samples))  (saga-sparse.h:261) */
uint32_t orc_draw(orc_rng* r, uint32_t n_samples) {
  double u;
  do { u = orc_unif_rand(r); } while (u <= 0.0 || u >= 1.0);
  return (uint32_t)floor(0.0 + ((double)n_samples - 0.0) * u);
}

void orc_fill_stream(orc_rng* r, uint32_t n_samples, uint32_t* out, int64_t count) {
  int64_t i;
  for (i = 0; i < count; ++i) out[i] = orc_draw(r, n_samples);
}

static uint32_t next_draw(orc_draws* d, uint32_t n_samples) {
  if (d->stream) {
    uint32_t s = d->stream[d->pos % (d->len > 0 ? d->len : 1)];
    d->pos++;
    return s;
  }
  d->pos++;
  return orc_draw(d->rng, n_samples);
}

/* ------------------------------------------------------------------------ */
/* src/prox.h:32-39                                                          */
/* ------------------------------------------------------------------------ */
static inline double soft_threshold(double x, double shrinkage) {
  return fmax(x - shrinkage, 0.0) - fmax(-x - shrinkage, 0.0);
}

/* ------------------------------------------------------------------------ */
/* src/penalties.h:27-79  (Ridge, ElasticNet, GroupLasso functors)           */
/* ------------------------------------------------------------------------ */
static inline void penalty_apply(int penalty, int K, double* w, int64_t j,
                                 double w_scale, double scaling, const double* g_sum,
                                 double gamma, double beta) {
  double* wj = w + j * K;
  const double* gj = g_sum + j * K;
  int k;
  switch (penalty) {
  case ORC_RIDGE: {                                      /* penalties.h:37 */
    double f = gamma / w_scale * scaling;
    for (k = 0; k < K; ++k) wj[k] -= f * gj[k];
    break;
  }
  case ORC_ELASTICNET: {                                 /* penalties.h:49-54 */
    for (k = 0; k < K; ++k) {
      wj[k] -= gamma / w_scale * scaling * gj[k];
      wj[k] = soft_threshold(wj[k], beta * gamma * scaling / w_scale);
    }
    break;
  }
  default: {                                             /* penalties.h:70-77 */
    double f = gamma / w_scale * scaling;
    double nrm = 0.0, factor;
    for (k = 0; k < K; ++k) wj[k] -= f * gj[k];
    for (k = 0; k < K; ++k) nrm += wj[k] * wj[k];
    nrm = sqrt(nrm);
    factor = beta * gamma * scaling / nrm;
    if (factor < 1.0) {
      double m = 1.0 - factor / w_scale;
      for (k = 0; k < K; ++k) wj[k] *= m;
    } else {
      for (k = 0; k < K; ++k) wj[k] = 0.0;
    }
  }
  }
}

/* ------------------------------------------------------------------------ */
/* src/math.h:25-33 LogSumExp                                                */
/* ------------------------------------------------------------------------ */
/* exp/log of the family gradients (src/families.h:161-168,244-260 via src/math.h:25-33): libm, like the
 * reference -- or, in the liboracle_det.so build (-DORC_DET_MATH), the plain-IEEE functions of
 * include/sgdnet_detmath.h that the HIP exact kernels use too, so that GPU and CPU agree bit for bit. */
#ifdef ORC_DET_MATH
#include "sgdnet_detmath.h"
#define ORC_EXP sgd_exp
#define ORC_LOG sgd_log
double orc_det_exp(double x) { return sgd_exp(x); }   /* test hooks (tests/test_detmath.py) */
double orc_det_log(double x) { return sgd_log(x); }
#else
#define ORC_EXP exp
#define ORC_LOG log
#endif

static double log_sum_exp(const double* x, int K) {
  double x_max = x[0], exp_sum = 0.0;
  int k;
  for (k = 1; k < K; ++k) if (x[k] > x_max) x_max = x[k];
  for (k = 0; k < K; ++k) exp_sum += ORC_EXP(x[k] - x_max);
  return ORC_LOG(exp_sum) + x_max;
}

/* ------------------------------------------------------------------------ */
/* Family::Gradient -- src/families.h:89-96 (gaussian), :161-168 (binomial), */
/* :244-260 (multinomial), :358-365 (mgaussian).  y is Ky x n col-major.     */
/* ------------------------------------------------------------------------ */
static void family_gradient(int family, int K, const double* lp, const double* y, int Ky,
                            int64_t i, double* g) {
  int k;
  switch (family) {
  case ORC_GAUSSIAN:
    g[0] = lp[0] - y[i];
    break;
  case ORC_BINOMIAL:
    g[0] = 1.0 - y[i] - 1.0 / (1.0 + ORC_EXP(lp[0]));
    break;
  case ORC_MULTINOMIAL: {
    double lse = log_sum_exp(lp, K);
    unsigned c = (unsigned)(y[i] + 0.5);
    for (k = 0; k < K; ++k) {
      g[k] = ORC_EXP(lp[k] - lse);
      if ((unsigned)k == c) g[k] -= 1.0;
    }
    break;
  }
  default:
    for (k = 0; k < K; ++k) g[k] = lp[k] - y[k + i * (int64_t)Ky];
  }
}

/* Family::Loss -- src/families.h:81-87, :152-159, :235-242, :350-356 */
static double family_loss(int family, int K, const double* lp, const double* y, int Ky,
                          int64_t i) {
  int k;
  switch (family) {
  case ORC_GAUSSIAN:
    return 0.5 * (lp[0] - y[i]) * (lp[0] - y[i]);
  case ORC_BINOMIAL:
    return log(1.0 + exp(lp[0])) - y[i] * lp[0];
  case ORC_MULTINOMIAL: {
    unsigned c = (unsigned)(y[i] + 0.5);
    return log_sum_exp(lp, K) - lp[c];
  }
  default: {
    double s = 0.0;
    for (k = 0; k < K; ++k) {
      double d = lp[k] - y[k + i * (int64_t)Ky];
      s += d * d;
    }
    return 0.5 * s;
  }
  }
}

/* ------------------------------------------------------------------------ */
/* ConvergenceCheck -- src/utils.h:240-262                                   */
/* ------------------------------------------------------------------------ */
static int convergence_check(const double* w, double* w_prev, int64_t len, double tol) {
  double max_change = 0.0, max_size = 0.0;
  int64_t i;
  for (i = 0; i < len; ++i) {
    double c = fabs(w[i] - w_prev[i]);
    double s = fabs(w[i]);
    if (c > max_change) max_change = c;
    if (s > max_size) max_size = s;
  }
  memcpy(w_prev, w, (size_t)len * sizeof(double));
  {
    int all_zero = (max_size == 0.0) && (max_change == 0.0);
    int no_change = (max_size != 0.0) && (max_change / max_size <= tol);
    return all_zero || no_change;
  }
}

/* linear predictor without wscale: used by EpochLoss / Deviance
 * (src/utils.h:219-223, :321-325) */
static void lp_sparse_plain(int K, const int64_t* ptr, const int32_t* idx, const double* val,
                            int64_t s, const double* w, const double* intercept, double* lp) {
  int k;
  int64_t q;
  for (k = 0; k < K; ++k) lp[k] = 0.0;
  for (q = ptr[s]; q < ptr[s + 1]; ++q) {
    const double* wj = w + (int64_t)idx[q] * K;
    for (k = 0; k < K; ++k) lp[k] += val[q] * wj[k];
  }
  for (k = 0; k < K; ++k) lp[k] += intercept[k];
}

static void w_dot_center(int K, int64_t p, const double* w, const double* c, double* out) {
  int k;
  int64_t j;
  for (k = 0; k < K; ++k) out[k] = 0.0;
  for (j = 0; j < p; ++j)
    for (k = 0; k < K; ++k) out[k] += w[k + j * K] * c[j];
}

/* EpochLoss (src/utils.h:199-227) when mean != 0, Deviance (:304-329) else */
static double total_loss_sparse(const orc_saga_params* P, const int64_t* ptr, const int32_t* idx,
                                const double* val, const double* c, const double* y, int Ky,
                                const double* w, const double* intercept, int mean) {
  int K = P->n_classes, k;
  int64_t n = P->n_samples, s;
  double lp[64], wc[64];
  double* lpd = K <= 64 ? lp : (double*)malloc(sizeof(double) * (size_t)K);
  double* wcd = K <= 64 ? wc : (double*)malloc(sizeof(double) * (size_t)K);
  double loss = 0.0;
  if (P->standardize) w_dot_center(K, P->n_features, w, c, wcd);
  for (s = 0; s < n; ++s) {
    lp_sparse_plain(K, ptr, idx, val, s, w, intercept, lpd);
    if (P->standardize)
      for (k = 0; k < K; ++k) lpd[k] -= wcd[k];
    if (mean)
      loss += family_loss(P->family, K, lpd, y, Ky, s) / (double)n;
    else
      loss += family_loss(P->family, K, lpd, y, Ky, s);
  }
  if (K > 64) { free(lpd); free(wcd); }
  return loss;
}

static double total_loss_dense(const orc_saga_params* P, const double* x, const double* y, int Ky,
                               const double* w, const double* intercept, int mean) {
  int K = P->n_classes, k;
  int64_t n = P->n_samples, p = P->n_features, s, j;
  double lp[64];
  double* lpd = K <= 64 ? lp : (double*)malloc(sizeof(double) * (size_t)K);
  double loss = 0.0;
  for (s = 0; s < n; ++s) {
    const double* xs = x + s * p;
    for (k = 0; k < K; ++k) lpd[k] = 0.0;
    for (j = 0; j < p; ++j)
      for (k = 0; k < K; ++k) lpd[k] += w[k + j * K] * xs[j];
    for (k = 0; k < K; ++k) lpd[k] += intercept[k];
    if (mean)
      loss += family_loss(P->family, K, lpd, y, Ky, s) / (double)n;
    else
      loss += family_loss(P->family, K, lpd, y, Ky, s);
  }
  if (K > 64) free(lpd);
  return loss;
}

/* ------------------------------------------------------------------------ */
/* Sparse SAGA -- src/saga-sparse.h:194-383                                  */
/*   LaggedUpdate :76-100, AddWeighted :114-130, Reset :132-155              */
/* ------------------------------------------------------------------------ */
unsigned orc_saga_sparse(const orc_saga_params* P,
                         const int64_t* ptr, const int32_t* idx, const double* val,
                         const double* c,
                         const double* y, int Ky,
                         double* intercept, double* w,
                         double* M, double* G, double* gb,
                         orc_draws* draws, unsigned* return_code,
                         double* losses) {
  const int K = P->n_classes;
  const int64_t n = P->n_samples, p = P->n_features;
  const double gamma = P->gamma, alpha = P->alpha, beta = P->beta;
  const int pen = P->penalty;
  int k;
  int64_t j, q;

  unsigned* lag = (unsigned*)calloc((size_t)p, sizeof(unsigned));            /* :225 */
  double wscale = 1.0;                                                        /* :227 */
  double* LS = (double*)malloc(sizeof(double) * (size_t)(n + 1 > 2 ? n + 1 : 2)); /* :229-240 */
  double geo_sum = 1.0;
  double wscale_update = 1.0 - alpha * gamma;                                 /* :234 */
  double* g = (double*)calloc((size_t)K, sizeof(double));
  double* gc = (double*)calloc((size_t)K, sizeof(double));
  double* lp = (double*)calloc((size_t)K, sizeof(double));
  double* wc = (double*)calloc((size_t)K, sizeof(double));
  double* w_prev = (double*)malloc(sizeof(double) * (size_t)(K * p));
  unsigned it_outer = 0;
  int converged = 0;
  int64_t i;

  LS[0] = 0.0;
  LS[1] = 1.0;
  for (i = 2; i < n + 1; ++i) {
    geo_sum *= wscale_update;
    LS[i] = LS[i - 1] + geo_sum;
  }
  memcpy(w_prev, w, sizeof(double) * (size_t)(K * p));                        /* :251 */

  do {
    unsigned it_inner;
    for (it_inner = 0; it_inner < (unsigned)n; ++it_inner) {
      uint32_t s = next_draw(draws, (uint32_t)n);                             /* :261 */
      const int64_t q0 = ptr[s], q1 = ptr[s + 1];

      /* LaggedUpdate(it_inner, ...) :263-272 */
      for (q = q0; q < q1; ++q) {
        j = idx[q];
        unsigned lagged = it_inner - lag[j];
        if (lagged != 0) {
          penalty_apply(pen, K, w, j, wscale, LS[lagged], G, gamma, beta);
          lag[j] = it_inner;
        }
      }

      /* :274 */
      for (k = 0; k < K; ++k) lp[k] = 0.0;
      for (q = q0; q < q1; ++q) {
        const double* wj = w + (int64_t)idx[q] * K;
        for (k = 0; k < K; ++k) lp[k] += val[q] * wj[k];
      }
      for (k = 0; k < K; ++k) lp[k] = lp[k] * wscale + intercept[k];

      if (P->standardize) {                                                   /* :276-277 */
        w_dot_center(K, p, w, c, wc);
        for (k = 0; k < K; ++k) lp[k] -= wc[k] * wscale;
      }

      family_gradient(P->family, K, lp, y, Ky, s, g);                         /* :279 */

      for (k = 0; k < K; ++k) {                                               /* :281-282 */
        gc[k] = g[k] - M[k + (int64_t)s * K];
        M[k + (int64_t)s * K] = g[k];
      }

      if (wscale < ORC_SMALL) {                                               /* :285-295 */
        for (j = 0; j < p; ++j) {
          unsigned lagged = it_inner - lag[j];
          if (lagged != 0) penalty_apply(pen, K, w, j, wscale, LS[lagged], G, gamma, beta);
        }
        for (i = 0; i < K * p; ++i) w[i] *= wscale;
        wscale = 1.0;
        for (j = 0; j < p; ++j) lag[j] = it_inner;
      }

      wscale *= wscale_update;                                                /* :297 */

      if (P->fit_intercept) {                                                 /* :300-304 */
        for (k = 0; k < K; ++k) {
          gb[k] += gc[k] / (double)n;
          intercept[k] -= gamma * (gb[k] * 0.01 + gc[k] / (double)n);
        }
      }

      /* AddWeighted(w, ..., -gamma/wscale) :306-313 */
      {
        double scaling = -gamma / wscale;
        for (k = 0; k < K; ++k) {
          for (q = q0; q < q1; ++q) w[k + (int64_t)idx[q] * K] += val[q] * gc[k] * scaling;
          if (P->standardize)
            for (j = 0; j < p; ++j) w[k + j * K] -= c[j] * gc[k] * scaling;
        }
      }

      /* LaggedUpdate(it_inner + 1, ...) :316-325 */
      for (q = q0; q < q1; ++q) {
        j = idx[q];
        unsigned lagged = (it_inner + 1) - lag[j];
        if (lagged != 0) {
          penalty_apply(pen, K, w, j, wscale, LS[lagged], G, gamma, beta);
          lag[j] = it_inner + 1;
        }
      }

      /* AddWeighted(g_sum, ..., 1.0/n_samples) :328-335 */
      {
        double scaling = 1.0 / (double)n;
        for (k = 0; k < K; ++k) {
          for (q = q0; q < q1; ++q) G[k + (int64_t)idx[q] * K] += val[q] * gc[k] * scaling;
          if (P->standardize)
            for (j = 0; j < p; ++j) G[k + j * K] -= c[j] * gc[k] * scaling;
        }
      }
    }

    /* Reset(n_samples, ...) :340-348 */
    for (j = 0; j < p; ++j) {
      unsigned lagged = (unsigned)n - lag[j];
      if (lagged != 0) penalty_apply(pen, K, w, j, wscale, LS[lagged], G, gamma, beta);
    }
    for (i = 0; i < K * p; ++i) w[i] *= wscale;
    wscale = 1.0;
    memset(lag, 0, sizeof(unsigned) * (size_t)p);

    if (P->debug && losses)                                                   /* :350-365 */
      losses[it_outer] = total_loss_sparse(P, ptr, idx, val, c, y, Ky, w, intercept, 1);

    converged = convergence_check(w, w_prev, K * p, P->tol);                  /* :367 */
    ++it_outer;
  } while (!converged && it_outer < P->max_iter);                             /* :371 */

  *return_code = (it_outer == P->max_iter) ? 1u : 0u;                         /* :376-382 */

  free(lag); free(LS); free(g); free(gc); free(lp); free(wc); free(w_prev);
  return it_outer;
}

/* ------------------------------------------------------------------------ */
/* Dense SAGA -- src/saga-dense.h:99-224                                     */
/* ------------------------------------------------------------------------ */
unsigned orc_saga_dense(const orc_saga_params* P,
                        const double* x,
                        const double* y, int Ky,
                        double* intercept, double* w,
                        double* M, double* G, double* gb,
                        orc_draws* draws, unsigned* return_code,
                        double* losses) {
  const int K = P->n_classes;
  const int64_t n = P->n_samples, p = P->n_features;
  const double gamma = P->gamma, alpha = P->alpha, beta = P->beta;
  const int pen = P->penalty;
  int k;
  int64_t j, i;
  double wscale = 1.0;                                                        /* :129 */
  double wscale_update = 1.0 - alpha * gamma;                                 /* :131 */
  double* g = (double*)calloc((size_t)K, sizeof(double));
  double* gc = (double*)calloc((size_t)K, sizeof(double));
  double* lp = (double*)calloc((size_t)K, sizeof(double));
  double* w_prev = (double*)malloc(sizeof(double) * (size_t)(K * p));
  unsigned it_outer = 0;
  int converged = 0;

  memcpy(w_prev, w, sizeof(double) * (size_t)(K * p));                        /* :142 */

  do {
    unsigned it_inner;
    for (it_inner = 0; it_inner < (unsigned)n; ++it_inner) {
      uint32_t s = next_draw(draws, (uint32_t)n);                             /* :152 */
      const double* xs = x + (int64_t)s * p;

      for (k = 0; k < K; ++k) lp[k] = 0.0;                                    /* :154 */
      for (j = 0; j < p; ++j)
        for (k = 0; k < K; ++k) lp[k] += w[k + j * K] * xs[j];
      for (k = 0; k < K; ++k) lp[k] = lp[k] * wscale + intercept[k];

      family_gradient(P->family, K, lp, y, Ky, s, g);                         /* :156 */

      for (k = 0; k < K; ++k) {                                               /* :158-159 */
        gc[k] = g[k] - M[k + (int64_t)s * K];
        M[k + (int64_t)s * K] = g[k];
      }

      if (wscale < ORC_SMALL) {                                               /* :162-166 */
        for (i = 0; i < K * p; ++i) w[i] *= wscale;
        wscale = 1.0;
      }

      wscale *= wscale_update;                                                /* :168 */

      if (P->fit_intercept) {                                                 /* :170-173 */
        for (k = 0; k < K; ++k) {
          gb[k] += gc[k] / (double)n;
          intercept[k] -= gamma * (gb[k] + gc[k] / (double)n);
        }
      }

      {                                                                       /* :176 */
        double f = gamma / wscale;
        for (j = 0; j < p; ++j)
          for (k = 0; k < K; ++k) w[k + j * K] -= gc[k] * xs[j] * f;
      }

      for (j = 0; j < p; ++j)                                                 /* :179-180 */
        penalty_apply(pen, K, w, j, wscale, 1.0, G, gamma, beta);

      for (j = 0; j < p; ++j)                                                 /* :183 */
        for (k = 0; k < K; ++k) G[k + j * K] += gc[k] * xs[j] / (double)n;
    }

    for (i = 0; i < K * p; ++i) w[i] *= wscale;                               /* :188-189 */
    wscale = 1.0;

    if (P->debug && losses)                                                   /* :191-206 */
      losses[it_outer] = total_loss_dense(P, x, y, Ky, w, intercept, 1);

    converged = convergence_check(w, w_prev, K * p, P->tol);                  /* :208 */
    ++it_outer;
  } while (!converged && it_outer < P->max_iter);

  *return_code = (it_outer == P->max_iter) ? 1u : 0u;                         /* :217-223 */
  free(g); free(gc); free(lp); free(w_prev);
  return it_outer;
}

/* ------------------------------------------------------------------------ */
/* B-stale sparse SAGA (DESIGN.md "Batched mode"): the reference iteration   */
/* saga-sparse.h:258-337 written in unscaled coordinates (true w = wscale*w) */
/* and applied to `batch` consecutive draws against one snapshot of          */
/* (w, intercept).  Per batch of m draws, for every feature j:               */
/*   D_j   = sum_i x_ij * gc_i           (first occurrence of a sample only) */
/*   w_j  <- r^m w_j - gamma*LS_m*G_j - gamma*D_j ; prox(beta*gamma*LS_m)    */
/*   G_j  <- G_j + D_j/n                                                     */
/* With implicit centring (standardize): lp -= c.w and D_j -= c_j sum_i gc_i. */
/* with r = 1 - alpha*gamma, LS_m = sum_{k<m} r^k  (the reference's          */
/* lag_scaling[m], saga-sparse.h:229-240, in closed form).                   */
/* ------------------------------------------------------------------------ */
void orc_batch_factors(double alpha, double gamma, int64_t m, double* r_m, double* ls_m) {
  double a = 1.0 - (1.0 - alpha * gamma);   /* 1 - r, exact for the rounded r */
  if (a == 0.0) {
    *r_m = 1.0;
    *ls_m = (double)m;
  } else {
    double e = expm1((double)m * log1p(-a)); /* r^m - 1 */
    *r_m = 1.0 + e;
    *ls_m = -e / a;
  }
}

/* The two halves of one batch, exposed separately so that the sample-sharded synchronous
 * mode (every rank gathers its share of a global batch, D and d0 are summed across ranks, every
 * rank sweeps) can be restated with exactly the code of the single-process loop below. */
void orc_batch_gather(const orc_saga_params* P, const int64_t* ptr, const int32_t* idx,
                      const double* val, const double* c, const double* y, int Ky,
                      const double* intercept, const double* w, double* M,
                      const uint32_t* draws, int64_t m, double* D, double* d0) {
  const int K = P->n_classes;
  const int64_t n = P->n_samples, p = P->n_features;
  const int centred = P->standardize && c != NULL;
  double* g = (double*)calloc((size_t)K, sizeof(double));
  double* lp = (double*)calloc((size_t)K, sizeof(double));
  double* wc = (double*)calloc((size_t)K, sizeof(double));
  unsigned char* seen = (unsigned char*)calloc((size_t)n, 1);
  int64_t i, q;
  int k;
  /* implicit centring (saga-sparse.h:276-277): (x_s - c).w = x_s.w - c.w, and c.w is one
   * scalar per class for the whole batch because w is a snapshot */
  if (centred) w_dot_center(K, p, w, c, wc);
  for (i = 0; i < m; ++i) {
    const uint32_t s = draws[i];
    if (seen[s]) continue;                  /* repeat within the batch: gc == 0 */
    seen[s] = 1;
    lp_sparse_plain(K, ptr, idx, val, s, w, intercept, lp);
    if (centred)
      for (k = 0; k < K; ++k) lp[k] -= wc[k];
    family_gradient(P->family, K, lp, y, Ky, s, g);
    for (k = 0; k < K; ++k) {
      const double gck = g[k] - M[k + (int64_t)s * K];
      M[k + (int64_t)s * K] = g[k];
      d0[k] += gck;
      for (q = ptr[s]; q < ptr[s + 1]; ++q) D[k + (int64_t)idx[q] * K] += val[q] * gck;
    }
  }
  for (i = 0; i < m; ++i) seen[draws[i]] = 0;
  free(g); free(lp); free(wc); free(seen);
}

/* m is the number of draws of the WHOLE batch (all ranks); D and d0 are zeroed. */
void orc_batch_sweep(const orc_saga_params* P, int64_t m, const double* c, double* D, double* d0,
                     double* intercept, double* w, double* G, double* gb) {
  const int K = P->n_classes;
  const int64_t p = P->n_features;
  const double gamma = P->gamma, beta = P->beta;
  const double nt = (double)(P->n_total > 0 ? P->n_total : P->n_samples);
  const int centred = P->standardize && c != NULL;
  double r_m, ls_m;
  int64_t j;
  int k;
  orc_batch_factors(P->alpha, gamma, m, &r_m, &ls_m);
  for (j = 0; j < p; ++j) {
    double* wj = w + j * K;
    double* gj = G + j * K;
    double* dj = D + j * K;
    /* AddWeighted's dense term (saga-sparse.h:127-128): sum_i (x_ij - c_j) gc_i */
    if (centred)
      for (k = 0; k < K; ++k) dj[k] -= c[j] * d0[k];
    for (k = 0; k < K; ++k) wj[k] = r_m * wj[k] - gamma * ls_m * gj[k] - gamma * dj[k];
    if (P->penalty == ORC_ELASTICNET) {
      for (k = 0; k < K; ++k) wj[k] = soft_threshold(wj[k], beta * gamma * ls_m);
    } else if (P->penalty == ORC_GROUPLASSO) {
      double nrm = 0.0, factor;
      for (k = 0; k < K; ++k) nrm += wj[k] * wj[k];
      nrm = sqrt(nrm);
      factor = beta * gamma * ls_m / nrm;
      if (factor < 1.0) {
        for (k = 0; k < K; ++k) wj[k] *= 1.0 - factor;
      } else {
        for (k = 0; k < K; ++k) wj[k] = 0.0;
      }
    }
    for (k = 0; k < K; ++k) {
      gj[k] += dj[k] / nt;
      dj[k] = 0.0;
    }
  }
  if (P->fit_intercept) {
    for (k = 0; k < K; ++k) {
      gb[k] += d0[k] / nt;
      intercept[k] -= gamma * (gb[k] * (P->dense_intercept ? 1.0 : 0.01) * (double)m + d0[k] / nt);
    }
  }
  for (k = 0; k < K; ++k) d0[k] = 0.0;
}

unsigned orc_saga_sparse_batched(const orc_saga_params* P, int64_t batch,
                                 const int64_t* ptr, const int32_t* idx, const double* val,
                                 const double* c,
                                 const double* y, int Ky,
                                 double* intercept, double* w,
                                 double* M, double* G, double* gb,
                                 orc_draws* draws, unsigned* return_code,
                                 double* losses) {
  const int K = P->n_classes;
  const int64_t n = P->n_samples, p = P->n_features;
  double* D = (double*)calloc((size_t)(K * p), sizeof(double));
  double* d0 = (double*)calloc((size_t)K, sizeof(double));
  double* w_prev = (double*)malloc(sizeof(double) * (size_t)(K * p));
  uint32_t* buf;
  unsigned it_outer = 0;
  int converged = 0;
  int64_t i;
  if (batch < 1) batch = 1;
  if (batch > n) batch = n;
  buf = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)batch);

  memcpy(w_prev, w, sizeof(double) * (size_t)(K * p));

  do {
    int64_t t0;
    for (t0 = 0; t0 < n; t0 += batch) {
      const int64_t m = (n - t0 < batch) ? n - t0 : batch;
      for (i = 0; i < m; ++i) buf[i] = next_draw(draws, (uint32_t)n);
      orc_batch_gather(P, ptr, idx, val, c, y, Ky, intercept, w, M, buf, m, D, d0);
      orc_batch_sweep(P, m, c, D, d0, intercept, w, G, gb);
    }

    if (P->debug && losses)
      losses[it_outer] = total_loss_sparse(P, ptr, idx, val, c, y, Ky, w, intercept, 1);

    converged = convergence_check(w, w_prev, K * p, P->tol);
    ++it_outer;
  } while (!converged && it_outer < P->max_iter);

  *return_code = (it_outer == P->max_iter) ? 1u : 0u;
  free(D); free(d0); free(w_prev); free(buf);
  return it_outer;
}

/* ======================================================================== */
/* Path driver -- src/sgdnet.cpp:119-285 and its helpers                     */
/* ======================================================================== */

/* math.h:66-79 Mean over an n x m column-major dense matrix */
static void dense_col_mean(const double* x, int64_t n, int64_t m, double* out) {
  int64_t i, j;
  for (j = 0; j < m; ++j) {
    double s = 0.0;
    for (i = 0; i < n; ++i) s += x[i + j * n];
    out[j] = s / (double)n;
  }
}

/* math.h:114-130 StandardDeviation (dense; population sd; 0 -> 1) */
static void dense_col_sd(const double* x, int64_t n, int64_t m, const double* mean, double* out) {
  int64_t i, j;
  for (j = 0; j < m; ++j) {
    double s = 0.0, var;
    for (i = 0; i < n; ++i) {
      double d = x[i + j * n] - mean[j];
      s += d * d;
    }
    var = s / (double)n;
    out[j] = (var == 0.0) ? 1.0 : sqrt(var);
  }
}

/* math.h:139-150 Standardize */
static void dense_standardize(double* x, int64_t n, int64_t m, const double* mean, const double* sd) {
  int64_t i, j;
  for (j = 0; j < m; ++j)
    for (i = 0; i < n; ++i) x[i + j * n] = (x[i + j * n] - mean[j]) / sd[j];
}

/* math.h:184-199 Proportions (y is the class-id vector) */
static void proportions(const double* y, int64_t n, int K, double* out) {
  int64_t i;
  int k;
  for (k = 0; k < K; ++k) out[k] = 0.0;
  for (i = 0; i < n; ++i) {
    int64_t c = (int64_t)(y[i] + 0.5);
    out[c] += 1.0 / (double)n;
  }
}

/* math.h:167-172 */
static double clamp(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

/* families.h:141-150 Binomial::Link */
static double binomial_link(double ybar) {
  double pmin = 1e-9, pmax = 1.0 - pmin;
  double z = clamp(ybar, pmin, pmax);
  return log(z / (1.0 - z));
}

/* Family::NullDeviance: families.h:98-110, :170-188, :262-285, :367-378.
 * yt is Ky x n column-major (samples in columns). */
static double null_deviance(int family, int K, const double* yt, int Ky, int64_t n,
                            int fit_intercept) {
  double* lp = (double*)calloc((size_t)(K > Ky ? K : Ky), sizeof(double));
  double loss = 0.0;
  int64_t i;
  int k;
  switch (family) {
  case ORC_GAUSSIAN:
  case ORC_MGAUSSIAN:
    for (k = 0; k < Ky; ++k) {
      double s = 0.0;
      for (i = 0; i < n; ++i) s += yt[k + i * Ky];
      lp[k] = s / (double)n;
    }
    for (i = 0; i < n; ++i) loss += family_loss(family, K, lp, yt, Ky, i);
    break;
  case ORC_BINOMIAL:
    if (fit_intercept) {
      double s = 0.0;
      for (i = 0; i < n; ++i) s += yt[i];
      lp[0] = binomial_link(s / (double)n);
    } else {
      lp[0] = 0.0;
    }
    for (i = 0; i < n; ++i) loss += family_loss(family, K, lp, yt, Ky, i);
    break;
  default: { /* multinomial */
    double lsum = 0.0, lse;
    if (fit_intercept)
      proportions(yt, n, K, lp);
    else
      for (k = 0; k < K; ++k) lp[k] = 1.0 / (double)K;
    for (k = 0; k < K; ++k) lsum += log(lp[k]);
    for (k = 0; k < K; ++k) lp[k] = log(lp[k]) - lsum / (double)K;
    lse = log_sum_exp(lp, K);
    for (i = 0; i < n; ++i) {
      unsigned c = (unsigned)(yt[i] + 0.5);
      loss += lse - lp[c];
    }
  }
  }
  free(lp);
  return 2.0 * loss;
}

/* Family::FitNullModel: families.h:112-117, :190-201, :287-298, :380-385 */
static void fit_null_model(int family, int K, const double* yt, int Ky, int64_t n,
                           int fit_intercept, double* intercept) {
  int64_t i;
  int k;
  switch (family) {
  case ORC_GAUSSIAN:
  case ORC_MGAUSSIAN:
    for (k = 0; k < Ky; ++k) {
      double s = 0.0;
      for (i = 0; i < n; ++i) s += yt[k + i * Ky];
      intercept[k] = s / (double)n;
    }
    break;
  case ORC_BINOMIAL:
    if (fit_intercept) {
      double s = 0.0;
      for (i = 0; i < n; ++i) s += yt[i];
      intercept[0] = binomial_link(s / (double)n);
    } else {
      intercept[0] = 0.0;
    }
    break;
  default: {
    double lsum = 0.0;
    if (fit_intercept)
      proportions(yt, n, K, intercept);
    else
      for (k = 0; k < K; ++k) intercept[k] = 1.0 / (double)K;
    for (k = 0; k < K; ++k) lsum += log(intercept[k]);
    for (k = 0; k < K; ++k) intercept[k] = log(intercept[k]) - lsum / (double)K;
  }
  }
}

/* x^T ymap for feature-major data; xty is p x Km column-major (Km = columns of ymap) */
typedef struct {
  int sparse;
  int64_t n, p;
  const int32_t* colptr; const int32_t* rowidx; const double* val; /* sparse, feature-major */
  const double* xd;                                                 /* dense n x p col-major */
} orc_xmat;

static void xt_times(const orc_xmat* X, const double* ymap, int Km, double* xty) {
  int64_t j, i;
  int k;
  for (j = 0; j < X->p; ++j) {
    for (k = 0; k < Km; ++k) {
      double s = 0.0;
      if (X->sparse) {
        int64_t q;
        for (q = X->colptr[j]; q < X->colptr[j + 1]; ++q)
          s += X->val[q] * ymap[X->rowidx[q] + (int64_t)k * X->n];
      } else {
        for (i = 0; i < X->n; ++i) s += X->xd[i + j * X->n] * ymap[i + (int64_t)k * X->n];
      }
      xty[j + (int64_t)k * X->p] = s;
    }
  }
}

/* Family::LambdaMax: families.h:119-126, :203-220, :300-325, :387-406.
 * y is n x Ky column-major (samples in rows), already preprocessed. */
static double lambda_max(int family, int K, const orc_xmat* X, const double* y, int Ky,
                         const double* y_scale) {
  int64_t n = X->n, p = X->p, i, j;
  int k;
  double out = 0.0;
  if (family == ORC_GAUSSIAN) {
    double* xty = (double*)malloc(sizeof(double) * (size_t)p);
    xt_times(X, y, 1, xty);
    for (j = 0; j < p; ++j) if (fabs(xty[j]) > out) out = fabs(xty[j]);
    free(xty);
    return y_scale[0] * out / (double)n;
  }
  if (family == ORC_BINOMIAL) {
    double* ymap = (double*)malloc(sizeof(double) * (size_t)n);
    double* xty = (double*)malloc(sizeof(double) * (size_t)p);
    double ybar, ystd;
    dense_col_mean(y, n, 1, &ybar);
    dense_col_sd(y, n, 1, &ybar, &ystd);
    for (i = 0; i < n; ++i) ymap[i] = (y[i] - ybar) / ystd;
    xt_times(X, ymap, 1, xty);
    for (j = 0; j < p; ++j) if (fabs(xty[j]) > out) out = fabs(xty[j]);
    free(ymap); free(xty);
    return ystd * out / (double)n;
  }
  if (family == ORC_MULTINOMIAL) {
    double* ymap = (double*)calloc((size_t)(n * K), sizeof(double));
    double* xty = (double*)malloc(sizeof(double) * (size_t)(p * K));
    double* ybar = (double*)malloc(sizeof(double) * (size_t)K);
    double* ystd = (double*)malloc(sizeof(double) * (size_t)K);
    for (i = 0; i < n; ++i) {
      unsigned c = (unsigned)(y[i] + 0.5);
      ymap[i + (int64_t)c * n] = 1.0;
    }
    dense_col_mean(ymap, n, K, ybar);
    dense_col_sd(ymap, n, K, ybar, ystd);
    dense_standardize(ymap, n, K, ybar, ystd);
    xt_times(X, ymap, K, xty);
    for (k = 0; k < K; ++k)
      for (j = 0; j < p; ++j) {
        double v = fabs(xty[j + (int64_t)k * p] * ystd[k]);
        if (v > out) out = v;
      }
    free(ymap); free(xty); free(ybar); free(ystd);
    return out / (double)n;
  }
  { /* mgaussian */
    double* ymap = (double*)malloc(sizeof(double) * (size_t)(n * Ky));
    double* xty = (double*)malloc(sizeof(double) * (size_t)(p * Ky));
    double* ybar = (double*)malloc(sizeof(double) * (size_t)Ky);
    double* ystd = (double*)malloc(sizeof(double) * (size_t)Ky);
    memcpy(ymap, y, sizeof(double) * (size_t)(n * Ky));
    dense_col_mean(y, n, Ky, ybar);
    dense_col_sd(y, n, Ky, ybar, ystd);
    dense_standardize(ymap, n, Ky, ybar, ystd);
    xt_times(X, ymap, Ky, xty);
    for (j = 0; j < p; ++j) {
      double s = 0.0;
      for (k = 0; k < Ky; ++k) {
        double v = xty[j + (int64_t)k * p] * (y_scale[k] * ystd[k]);
        s += v * v;
      }
      s = sqrt(s);
      if (s > out) out = s;
    }
    free(ymap); free(xty); free(ybar); free(ystd);
    return out / (double)n;
  }
}

/* utils.h:31-51 StepSize */
static double step_size(double max_squared_sum, double alpha_i, int fit_intercept,
                        double L_scaling, int64_t n) {
  double L = (max_squared_sum + (double)(fit_intercept ? 1 : 0)) * L_scaling + alpha_i;
  double mu_n = 2.0 * (double)n * alpha_i;
  return 1.0 / (2.0 * L + fmin(L, mu_n));
}

static int penalty_for(int family, double mix) {                 /* sgdnet.cpp:80-98 */
  if (mix == 0.0) return ORC_RIDGE;
  if (family == ORC_MGAUSSIAN) return ORC_GROUPLASSO;
  return ORC_ELASTICNET;
}

static double family_L_scaling(int family) {                     /* families.h:66,131,225,334 */
  return (family == ORC_GAUSSIAN || family == ORC_MGAUSSIAN) ? 1.0 : 0.25;
}

/* Response preprocessing: families.h:68-79 (gaussian), :337-348 (mgaussian) */
static void preprocess_response(int family, int standardize_response, double* y, int64_t n, int Ky,
                                double* y_center, double* y_scale) {
  if (family == ORC_GAUSSIAN) {
    int64_t i;
    dense_col_mean(y, n, 1, y_center);
    dense_col_sd(y, n, 1, y_center, y_scale);
    for (i = 0; i < n; ++i) y[i] = (y[i] - y_center[0]) / y_scale[0];
  } else if (family == ORC_MGAUSSIAN && standardize_response) {
    double* m = (double*)malloc(sizeof(double) * (size_t)Ky);
    double* s = (double*)malloc(sizeof(double) * (size_t)Ky);
    dense_col_mean(y, n, Ky, m);
    dense_col_sd(y, n, Ky, m, s);
    dense_standardize(y, n, Ky, m, s);
    free(m); free(s);
  }
}

/* Shared tail of SetupSgdnet: sgdnet.cpp:160-285 */
static int fit_common(const orc_xmat* X,            /* feature-major, preprocessed */
                      const int64_t* sptr, const int32_t* sidx, const double* sval, /* sample-major sparse */
                      const double* xt_dense,       /* p x n dense sample-major */
                      const double* x_center, const double* x_scale,
                      const double* x_center_scaled,
                      double* y, int Ky,            /* n x Ky, preprocessed in place here */
                      const orc_control* ctl, orc_draws* draws, orc_result* out) {
  const int family = ctl->family, K = ctl->n_classes;
  const int64_t n = X->n, p = X->p;
  const int n_lambda = ctl->n_lambda;
  int64_t i, j;
  int k, li;
  double* y_center = (double*)calloc((size_t)K, sizeof(double));
  double* y_scale = (double*)malloc(sizeof(double) * (size_t)K);
  double* yt = (double*)calloc((size_t)(n * Ky), sizeof(double));
  double* lambda = (double*)malloc(sizeof(double) * (size_t)n_lambda);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)n_lambda);
  double* beta = (double*)malloc(sizeof(double) * (size_t)n_lambda);
  double* intercept = (double*)calloc((size_t)K, sizeof(double));
  double* w = (double*)calloc((size_t)(K * p), sizeof(double));
  double* gb = (double*)calloc((size_t)K, sizeof(double));
  double* M = (double*)calloc((size_t)(K * n), sizeof(double));
  double* G = (double*)calloc((size_t)(K * p), sizeof(double));
  double* losses = ctl->debug ? (double*)malloc(sizeof(double) * (size_t)ctl->max_iter) : NULL;
  double max_scale, norm_max = 0.0, null_dev_scaled;
  unsigned n_iter = 0;
  orc_saga_params P;

  for (k = 0; k < K; ++k) y_scale[k] = 1.0;

  /* null deviance on the original response: sgdnet.cpp:154 */
  for (i = 0; i < n; ++i)
    for (k = 0; k < Ky; ++k) yt[k + i * Ky] = y[i + (int64_t)k * n];
  out->nulldev = null_deviance(family, K, yt, Ky, n, ctl->fit_intercept);

  preprocess_response(family, ctl->standardize_response, y, n, Ky, y_center, y_scale); /* :156-158 */

  /* RegularizationPath: utils.h:142-181 */
  if (ctl->lambda == NULL) {
    double mix = ctl->elasticnet_mix;
    double lmax = lambda_max(family, K, X, y, Ky, y_scale) / fmax(mix, 0.001);
    if (lmax != 0.0) {                                   /* math.h:42-56 LogSpace */
      double log_from = log(lmax);
      double step = (log(lmax * ctl->lambda_min_ratio) - log_from) / (double)(n_lambda - 1);
      for (li = 0; li < n_lambda; ++li) lambda[li] = exp(log_from + li * step);
    } else {
      for (li = 0; li < n_lambda; ++li) lambda[li] = 0.0;
    }
  } else {
    memcpy(lambda, ctl->lambda, sizeof(double) * (size_t)n_lambda);
  }
  max_scale = y_scale[0];
  for (k = 1; k < K; ++k) if (y_scale[k] > max_scale) max_scale = y_scale[k];
  for (li = 0; li < n_lambda; ++li) {
    alpha[li] = (1.0 - ctl->elasticnet_mix) * lambda[li] / max_scale;
    beta[li] = ctl->elasticnet_mix * lambda[li] / max_scale;
  }

  /* transposed response: sgdnet.cpp:178 */
  for (i = 0; i < n; ++i)
    for (k = 0; k < Ky; ++k) yt[k + i * Ky] = y[i + (int64_t)k * n];

  /* ColNormsMax: utils.h:60-85 */
  if (X->sparse) {
    double csq = 0.0;
    if (ctl->standardize)
      for (j = 0; j < p; ++j) csq += x_center_scaled[j] * x_center_scaled[j];
    for (i = 0; i < n; ++i) {
      double nrm = 0.0;
      int64_t q;
      if (ctl->standardize) {
        /* ||x_i - c||^2 = sum_nz (x-c)^2 + sum_{not nz} c^2 */
        double cnz = 0.0;
        for (q = sptr[i]; q < sptr[i + 1]; ++q) {
          double cj = x_center_scaled[sidx[q]];
          double d = sval[q] - cj;
          nrm += d * d;
          cnz += cj * cj;
        }
        nrm += csq - cnz;
      } else {
        for (q = sptr[i]; q < sptr[i + 1]; ++q) nrm += sval[q] * sval[q];
      }
      if (nrm > norm_max) norm_max = nrm;
    }
  } else {
    for (i = 0; i < n; ++i) {
      double nrm = 0.0;
      for (j = 0; j < p; ++j) nrm += xt_dense[j + i * p] * xt_dense[j + i * p];
      if (nrm > norm_max) norm_max = nrm;
    }
  }

  fit_null_model(family, K, yt, Ky, n, ctl->fit_intercept, intercept);       /* :210 */
  null_dev_scaled = null_deviance(family, K, yt, Ky, n, ctl->fit_intercept); /* :211 */

  memset(&P, 0, sizeof(P));
  P.family = family;
  P.penalty = penalty_for(family, ctl->elasticnet_mix);
  P.n_classes = K;
  P.n_samples = n;
  P.n_features = p;
  P.fit_intercept = ctl->fit_intercept;
  P.standardize = X->sparse ? ctl->standardize : 0;
  P.max_iter = ctl->max_iter;
  P.tol = ctl->tol;
  P.debug = ctl->debug;

  for (li = 0; li < n_lambda; ++li) {                                        /* :217-273 */
    unsigned rc = 0, epochs;
    double dev;
    P.alpha = alpha[li];
    P.beta = beta[li];
    P.gamma = step_size(norm_max, alpha[li], ctl->fit_intercept, family_L_scaling(family), n);
    if (X->sparse) {
      if (ctl->batch >= 1)
        epochs = orc_saga_sparse_batched(&P, ctl->batch, sptr, sidx, sval, x_center_scaled, yt, Ky,
                                         intercept, w,
                                         M, G, gb, draws, &rc, losses);
      else
        epochs = orc_saga_sparse(&P, sptr, sidx, sval, x_center_scaled, yt, Ky, intercept, w,
                                 M, G, gb, draws, &rc, losses);
      dev = 2.0 * total_loss_sparse(&P, sptr, sidx, sval, x_center_scaled, yt, Ky, w, intercept, 0);
    } else {
      epochs = orc_saga_dense(&P, xt_dense, yt, Ky, intercept, w, M, G, gb, draws, &rc, losses);
      dev = 2.0 * total_loss_dense(&P, xt_dense, yt, Ky, w, intercept, 0);
    }
    n_iter += epochs;
    out->return_codes[li] = (double)rc;
    out->dev_ratio[li] = 1.0 - dev / null_dev_scaled;                         /* :258 */
    out->lambda[li] = lambda[li];
    if (out->step_size) out->step_size[li] = P.gamma;
    if (out->alpha_l2) out->alpha_l2[li] = alpha[li];
    if (out->beta_l1) out->beta_l1[li] = beta[li];
    if (ctl->debug && out->losses) {
      memcpy(out->losses + (int64_t)li * ctl->max_iter, losses, sizeof(double) * epochs);
      out->losses_len[li] = (int)epochs;
    }

    /* Rescale: utils.h:352-378 */
    {
      double* bo = out->beta + (int64_t)li * K * p;
      double* ao = out->a0 + (int64_t)li * K;
      double* xbb = (double*)calloc((size_t)K, sizeof(double));
      for (j = 0; j < p; ++j)
        for (k = 0; k < K; ++k) {
          double v = w[k + j * K] * (y_scale[k] / x_scale[j]);
          bo[k + j * K] = v;
          xbb[k] += x_center[j] * v;
        }
      for (k = 0; k < K; ++k)
        ao[k] = ctl->fit_intercept ? intercept[k] * y_scale[k] + y_center[k] - xbb[k]
                                   : intercept[k];
      free(xbb);
    }
  }
  out->npasses = (double)n_iter;

  free(y_center); free(y_scale); free(yt); free(lambda); free(alpha); free(beta);
  free(intercept); free(w); free(gb); free(M); free(G); free(losses);
  return 0;
}

int orc_fit_sparse(int64_t n, int64_t p, const int32_t* colptr, const int32_t* rowidx,
                   const double* val_in, const double* y_in, int y_cols,
                   const orc_control* ctl, orc_draws* draws, orc_result* out) {
  int64_t nnz = colptr[p], j, q;
  double* val = (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  double* y = (double*)malloc(sizeof(double) * (size_t)(n * y_cols));
  double* x_center = (double*)calloc((size_t)p, sizeof(double));
  double* x_scale = (double*)malloc(sizeof(double) * (size_t)p);
  double* xcs = (double*)calloc((size_t)p, sizeof(double));
  int64_t* sptr = (int64_t*)calloc((size_t)(n + 1), sizeof(int64_t));
  int32_t* sidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  double* sval = (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  int64_t* fill;
  orc_xmat X;
  int rc;

  memcpy(val, val_in, sizeof(double) * (size_t)nnz);
  memcpy(y, y_in, sizeof(double) * (size_t)(n * y_cols));
  for (j = 0; j < p; ++j) x_scale[j] = 1.0;

  if (ctl->standardize) {                     /* utils.h:110-121, math.h:66-79,:89-112 */
    for (j = 0; j < p; ++j) {
      double s = 0.0, var = 0.0;
      int64_t nz = colptr[j + 1] - colptr[j];
      for (q = colptr[j]; q < colptr[j + 1]; ++q) s += val[q];
      x_center[j] = s / (double)n;
      for (q = colptr[j]; q < colptr[j + 1]; ++q)
        var += pow(val[q] - x_center[j], 2) / (double)n;
      var += (double)(n - nz) * x_center[j] * x_center[j] / (double)n;
      x_scale[j] = (var == 0.0) ? 1.0 : sqrt(var);
      for (q = colptr[j]; q < colptr[j + 1]; ++q) val[q] /= x_scale[j];
    }
  }
  for (j = 0; j < p; ++j) xcs[j] = x_center[j] / x_scale[j];               /* sgdnet.cpp:150 */

  /* AdaptiveTranspose (utils.h:276-281): feature-major -> sample-major */
  for (q = 0; q < nnz; ++q) sptr[rowidx[q] + 1]++;
  for (j = 0; j < n; ++j) sptr[j + 1] += sptr[j];
  fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  memcpy(fill, sptr, sizeof(int64_t) * (size_t)n);
  for (j = 0; j < p; ++j)
    for (q = colptr[j]; q < colptr[j + 1]; ++q) {
      int64_t d = fill[rowidx[q]]++;
      sidx[d] = (int32_t)j;
      sval[d] = val[q];
    }
  free(fill);

  X.sparse = 1; X.n = n; X.p = p; X.colptr = colptr; X.rowidx = rowidx; X.val = val; X.xd = NULL;
  rc = fit_common(&X, sptr, sidx, sval, NULL, x_center, x_scale, xcs, y, y_cols, ctl, draws, out);

  free(val); free(y); free(x_center); free(x_scale); free(xcs); free(sptr); free(sidx); free(sval);
  return rc;
}

int orc_fit_dense(int64_t n, int64_t p, const double* x_in, const double* y_in, int y_cols,
                  const orc_control* ctl, orc_draws* draws, orc_result* out) {
  double* x = (double*)malloc(sizeof(double) * (size_t)(n * p));
  double* xt = (double*)malloc(sizeof(double) * (size_t)(n * p));
  double* y = (double*)malloc(sizeof(double) * (size_t)(n * y_cols));
  double* x_center = (double*)calloc((size_t)p, sizeof(double));
  double* x_scale = (double*)malloc(sizeof(double) * (size_t)p);
  double* xcs = (double*)calloc((size_t)p, sizeof(double));      /* zeros: sgdnet.cpp:151 */
  int64_t i, j;
  orc_xmat X;
  int rc;

  memcpy(x, x_in, sizeof(double) * (size_t)(n * p));
  memcpy(y, y_in, sizeof(double) * (size_t)(n * y_cols));
  for (j = 0; j < p; ++j) x_scale[j] = 1.0;
  if (ctl->standardize) {                                        /* utils.h:99-108 */
    dense_col_mean(x, n, p, x_center);
    dense_col_sd(x, n, p, x_center, x_scale);
    dense_standardize(x, n, p, x_center, x_scale);
  }
  for (i = 0; i < n; ++i)                                        /* utils.h:283-288 */
    for (j = 0; j < p; ++j) xt[j + i * p] = x[i + j * n];

  X.sparse = 0; X.n = n; X.p = p; X.colptr = NULL; X.rowidx = NULL; X.val = NULL; X.xd = x;
  rc = fit_common(&X, NULL, NULL, NULL, xt, x_center, x_scale, xcs, y, y_cols, ctl, draws, out);

  free(x); free(xt); free(y); free(x_center); free(x_scale); free(xcs);
  return rc;
}
