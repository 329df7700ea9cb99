// Per-sample operators of the SAGA loop as gfx950 device functions.
// Reference behaviour (paths relative to the reference tree):
//   src/prox.h:32-39        SoftThreshold
//   src/penalties.h:27-79   Ridge / ElasticNet / GroupLasso functors
//   src/families.h          Gradient / Loss of the four families
//   src/math.h:25-33        LogSumExp
// Arithmetic order follows the reference expressions;
// this translation unit is built with -ffp-contract=off.
#pragma once

#include "common.hpp"

// exp/log of the family gradients: the exact kernels (saga_exact.hip defines SGDNET_DET_MATH) use the
// plain-IEEE functions of include/sgdnet_detmath.h, bit-identical to the CPU restatement built with
// -DORC_DET_MATH; the throughput kernels keep the device math library.
#ifdef SGDNET_DET_MATH
#include "sgdnet_detmath.h"
#define SGD_EXP sgd_exp
#define SGD_LOG sgd_log
#else
#define SGD_EXP exp
#define SGD_LOG log
#endif

namespace sgdnet {

__device__ __forceinline__ double soft_threshold(double x, double s) {
  return fmax(x - s, 0.0) - fmax(-x - s, 0.0);
}

// penalty(w, j, w_scale, scaling, g_sum) for one feature column (K entries).
// (WP / GP: pointers to the K coefficients / gradient averages of the feature, in whatever address space the
// kernel keeps them -- a pointer that may be LDS or global becomes a FLAT access, which drains both counters)
template <typename WP, typename GP>
__device__ __forceinline__ void penalty_apply(int penalty, int K, WP wj, GP gj, double w_scale, double scaling,
                                              double gamma, double beta) {
  if (penalty == SGDNET_RIDGE) {
    const double f = gamma / w_scale * scaling;
    for (int k = 0; k < K; ++k) wj[k] -= f * gj[k];
  } else if (penalty == SGDNET_ELASTICNET) {
    const double f = gamma / w_scale * scaling;
    const double tau = beta * gamma * scaling / w_scale;
    for (int k = 0; k < K; ++k) {
      double v = wj[k] - f * gj[k];
      wj[k] = soft_threshold(v, tau);
    }
  } else {
    const double f = gamma / w_scale * scaling;
    double nrm = 0.0;
    for (int k = 0; k < K; ++k) {
      double v = wj[k] - f * gj[k];
      wj[k] = v;
      nrm += v * v;
    }
    nrm = sqrt(nrm);
    const double factor = beta * gamma * scaling / nrm;
    if (factor < 1.0) {
      const double m = 1.0 - factor / w_scale;
      for (int k = 0; k < K; ++k) wj[k] *= m;
    } else {
      for (int k = 0; k < K; ++k) wj[k] = 0.0;
    }
  }
}

// The same with q = gamma / w_scale formed once by the caller (gamma / w_scale * scaling is that quotient
// times scaling: the same doubles), for kernels that apply the penalty several times per iteration.
template <typename WP, typename GP>
__device__ __forceinline__ void penalty_apply_q(int penalty, int K, WP wj, GP gj, double w_scale, double scaling, double q,
                                                double gamma, double beta) {
  const double f = q * scaling;
  if (penalty == SGDNET_RIDGE) {
    for (int k = 0; k < K; ++k) wj[k] -= f * gj[k];
  } else if (penalty == SGDNET_ELASTICNET) {
    const double tau = beta * gamma * scaling / w_scale;
    for (int k = 0; k < K; ++k) {
      double v = wj[k] - f * gj[k];
      wj[k] = soft_threshold(v, tau);
    }
  } else {
    double nrm = 0.0;
    for (int k = 0; k < K; ++k) {
      double v = wj[k] - f * gj[k];
      wj[k] = v;
      nrm += v * v;
    }
    nrm = sqrt(nrm);
    const double factor = beta * gamma * scaling / nrm;
    if (factor < 1.0) {
      const double m = 1.0 - factor / w_scale;
      for (int k = 0; k < K; ++k) wj[k] *= m;
    } else {
      for (int k = 0; k < K; ++k) wj[k] = 0.0;
    }
  }
}

// LogSumExp over K linear predictors, ascending order.
template <typename LP>
__device__ __forceinline__ double log_sum_exp(LP lp, int K) {
  double mx = lp[0];
  for (int k = 1; k < K; ++k) mx = lp[k] > mx ? lp[k] : mx;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += SGD_EXP(lp[k] - mx);
  return SGD_LOG(s) + mx;
}

// g_k for class k given all K linear predictors; y points at column s of y (Ky rows).
template <typename LP>
__device__ __forceinline__ double family_gradient_k(int family, int K, int k, LP lp, const double* ys) {
  switch (family) {
    case SGDNET_GAUSSIAN:
      return lp[0] - ys[0];
    case SGDNET_BINOMIAL:
      return 1.0 - ys[0] - 1.0 / (1.0 + SGD_EXP(lp[0]));
    case SGDNET_MULTINOMIAL: {
      const double lse = log_sum_exp(lp, K);
      const unsigned c = (unsigned)(ys[0] + 0.5);
      double g = SGD_EXP(lp[k] - lse);
      if ((unsigned)k == c) g -= 1.0;
      return g;
    }
    default:
      return lp[k] - ys[k];
  }
}

__device__ __forceinline__ double family_loss(int family, int K, const double* lp, const double* ys) {
  switch (family) {
    case SGDNET_GAUSSIAN:
      return 0.5 * (lp[0] - ys[0]) * (lp[0] - ys[0]);
    case SGDNET_BINOMIAL:
      return log(1.0 + exp(lp[0])) - ys[0] * lp[0];
    case SGDNET_MULTINOMIAL: {
      const unsigned c = (unsigned)(ys[0] + 0.5);
      return log_sum_exp(lp, K) - lp[c];
    }
    default: {
      double s = 0.0;
      for (int k = 0; k < K; ++k) {
        const double dlt = lp[k] - ys[k];
        s += dlt * dlt;
      }
      return 0.5 * s;
    }
  }
}

__device__ __forceinline__ double wave_max(double v) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace sgdnet
