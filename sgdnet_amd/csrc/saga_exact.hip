// Exact-order SAGA epoch kernels for gfx950 (the parity anchor).
//
// One 64-lane wavefront executes the reference iteration one draw at a time, in
// the order of the resident sample stream, so results agree with the CPU
// restatement to rounding of libm calls only:
//   saga_sparse_exact_kernel  <-  Saga(), src/saga-sparse.h:194-383
//                                 (LaggedUpdate :76-100, AddWeighted :114-130, Reset :132-155)
//   saga_dense_exact_kernel   <-  Saga(), src/saga-dense.h:99-224
// The chain through `intercept` makes every pair of consecutive iterations
// dependent, so this kernel is latency-bound by construction;
// the throughput path is saga_batched.hip.
//
// Lane roles: lanes stride the nonzeros of the drawn sample for the per-feature
// steps; lane k owns class k for the linear predictor / gradient / intercept.
// w, g_sum and lag are staged in LDS when they fit (160 KiB per CU).
#include "device_math.hpp"

namespace sgdnet {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// ConvergenceCheck (src/utils.h:240-262), executed by the whole wave.
__device__ __forceinline__ int convergence_check(const double* w, double* w_prev, int64_t len,
                                                 double tol, int lane) {
  double max_change = 0.0, max_size = 0.0;
  for (int64_t i = lane; i < len; i += kWave) {
    const double v = w[i];
    max_change = fmax(max_change, fabs(v - w_prev[i]));
    max_size = fmax(max_size, fabs(v));
    w_prev[i] = v;
  }
  max_change = wave_max(max_change);
  max_size = wave_max(max_size);
  const bool all_zero = (max_size == 0.0) && (max_change == 0.0);
  const bool no_change = (max_size != 0.0) && (max_change / max_size <= tol);
  return (all_zero || no_change) ? 1 : 0;
}

}  // namespace

__global__ __launch_bounds__(kWave) void saga_sparse_exact_kernel(SagaDev d, const LamParams* lamp,
                                                                  ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int K = d.K;
  const int64_t p = d.p;
  const int64_t KP = (int64_t)K * p;

  // LDS carve: [slp K][sgc K][w KP][G KP][lag p]
  double* slp = reinterpret_cast<double*>(smem);
  double* sgc = slp + K;
  double* w = d.w;
  double* G = d.G;
  unsigned* lag = d.lag;
  if (ctl.use_lds) {
    w = sgc + K;
    G = w + KP;
    lag = reinterpret_cast<unsigned*>(G + KP);
    for (int64_t i = lane; i < KP; i += kWave) {
      w[i] = d.w[i];
      G[i] = d.G[i];
    }
  }
  for (int64_t j = lane; j < p; j += kWave) lag[j] = 0u;            // saga-sparse.h:225
  for (int64_t i = lane; i < KP; i += kWave) d.w_prev[i] = w[i];     // :251
  wave_sync();

  const int penalty = lamp->penalty;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :234
  const double n_d = d.n_total;
  const double* LS = ctl.LS;
  const unsigned nit = (unsigned)ctl.nit;
  double wscale = 1.0;                                               // :227

  unsigned it_outer = 0;
  int converged = 0;
  int64_t t = ctl.stream_off;
  do {
    for (unsigned it = 0; it < nit; ++it, ++t) {
      const uint32_t s = d.stream[t];                                // :261
      const int64_t q0 = d.ptr[s], q1 = d.ptr[s + 1];

      // LaggedUpdate(it_inner): catch-up of the sample's features  :263-272
      for (int64_t q = q0 + lane; q < q1; q += kWave) {
        const int64_t j = d.idx[q];
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply(penalty, K, w + j * K, G + j * K, wscale, LS[lagged], gamma, beta);
          lag[j] = it;
        }
      }
      wave_sync();

      // linear predictor: lane k accumulates in ascending feature order  :274
      for (int k = lane; k < K; k += kWave) {
        double acc = 0.0;
        for (int64_t q = q0; q < q1; ++q) acc += d.val[q] * w[k + (int64_t)d.idx[q] * K];
        slp[k] = acc * wscale + d.b[k];
      }
      if (d.standardize) {                                           // :276-277
        wave_sync();
        for (int k = 0; k < K; ++k) {
          double part = 0.0;
          for (int64_t j = lane; j < p; j += kWave) part += w[k + j * K] * d.c[j];
          part = wave_sum(part);
          if (lane == 0) slp[k] -= part * wscale;
        }
      }
      wave_sync();

      // gradient, gradient memory  :279-282
      for (int k = lane; k < K; k += kWave) {
        const double g = family_gradient_k(d.family, K, k, slp, d.y + (int64_t)s * d.Ky);
        const int64_t mi = k + (int64_t)s * K;
        sgc[k] = g - d.M[mi];
        d.M[mi] = g;
      }

      // rescale + unlag whenever wscale becomes too small  :285-295
      if (wscale < kSmall) {
        wave_sync();
        for (int64_t j = lane; j < p; j += kWave) {
          const unsigned lagged = it - lag[j];
          if (lagged != 0)
            penalty_apply(penalty, K, w + j * K, G + j * K, wscale, LS[lagged], gamma, beta);
          for (int k = 0; k < K; ++k) w[k + j * K] *= wscale;
          lag[j] = it;
        }
        wscale = 1.0;
      }

      wscale *= wscale_update;                                       // :297
      wave_sync();

      if (d.fit_intercept) {                                         // :300-304
        for (int k = lane; k < K; k += kWave) {
          const double gck = sgc[k] / n_d;
          const double gbk = d.gb[k] + gck;
          d.gb[k] = gbk;
          d.b[k] -= gamma * (gbk * 0.01 + gck);
        }
      }

      // AddWeighted(w, ..., -gamma/wscale)  :306-313
      {
        const double scaling = -gamma / wscale;
        for (int64_t q = q0 + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const double v = d.val[q];
          for (int k = 0; k < K; ++k) w[k + j * K] += v * sgc[k] * scaling;
        }
        if (d.standardize) {
          wave_sync();
          for (int64_t j = lane; j < p; j += kWave) {
            const double cj = d.c[j];
            for (int k = 0; k < K; ++k) w[k + j * K] -= cj * sgc[k] * scaling;
          }
          wave_sync();
        }
      }

      // LaggedUpdate(it_inner + 1): the SAGA step on the sample's features  :316-325,
      // then AddWeighted(g_sum, ..., 1/n)  :328-335 (same lane, same feature)
      {
        const double scaling = 1.0 / n_d;
        for (int64_t q = q0 + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const unsigned lagged = (it + 1) - lag[j];
          if (lagged != 0) {
            penalty_apply(penalty, K, w + j * K, G + j * K, wscale, LS[lagged], gamma, beta);
            lag[j] = it + 1;
          }
          const double v = d.val[q];
          for (int k = 0; k < K; ++k) G[k + j * K] += v * sgc[k] * scaling;
        }
        if (d.standardize) {
          wave_sync();
          for (int64_t j = lane; j < p; j += kWave) {
            const double cj = d.c[j];
            for (int k = 0; k < K; ++k) G[k + j * K] -= cj * sgc[k] * scaling;
          }
        }
      }
      wave_sync();
    }

    // Reset(n_samples): unlag and rescale  :340-348
    for (int64_t j = lane; j < p; j += kWave) {
      const unsigned lagged = nit - lag[j];
      if (lagged != 0)
        penalty_apply(penalty, K, w + j * K, G + j * K, wscale, LS[lagged], gamma, beta);
      for (int k = 0; k < K; ++k) w[k + j * K] *= wscale;
      lag[j] = 0u;
    }
    wscale = 1.0;
    wave_sync();

    converged = convergence_check(w, d.w_prev, KP, ctl.tol, lane);   // :367
    ++it_outer;
    wave_sync();
  } while (!converged && it_outer < ctl.max_epochs);                 // :371

  if (ctl.use_lds) {
    for (int64_t i = lane; i < KP; i += kWave) {
      d.w[i] = w[i];
      d.G[i] = G[i];
    }
  }
  if (lane == 0) {
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = converged;
  }
}

// --------------------------------------------------------------------------
// Dense variant: no lag, all-feature penalty every iteration (saga-dense.h:179-180),
// intercept without the 0.01 decay (:170-173).  Lanes stride the features for
// the three O(pK) passes; lane k owns class k.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void saga_dense_exact_kernel(SagaDev d, const LamParams* lamp,
                                                                 ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int K = d.K;
  const int64_t p = d.p;
  const int64_t KP = (int64_t)K * p;

  // LDS carve: [slp K][sgc K][xs p][w KP][G KP]
  double* slp = reinterpret_cast<double*>(smem);
  double* sgc = slp + K;
  double* xs = sgc + K;
  double* w = d.w;
  double* G = d.G;
  if (ctl.use_lds) {
    w = xs + p;
    G = w + KP;
    for (int64_t i = lane; i < KP; i += kWave) {
      w[i] = d.w[i];
      G[i] = d.G[i];
    }
  }
  for (int64_t i = lane; i < KP; i += kWave) d.w_prev[i] = w[i];     // saga-dense.h:142
  wave_sync();

  const int penalty = lamp->penalty;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :131
  const double n_d = d.n_total;
  const unsigned nit = (unsigned)ctl.nit;
  double wscale = 1.0;                                               // :129

  unsigned it_outer = 0;
  int converged = 0;
  int64_t t = ctl.stream_off;
  // software prefetch of the next sample's row (the stream is known in advance)
  constexpr int kPf = 4;
  double xn[kPf];
  {
    const uint32_t s0 = d.stream[t];
#pragma unroll
    for (int c = 0; c < kPf; ++c) {
      const int64_t j = lane + (int64_t)c * kWave;
      xn[c] = j < p ? d.xd[(int64_t)s0 * p + j] : 0.0;
    }
  }
  const int64_t t_end = ctl.stream_off + (int64_t)ctl.max_epochs * nit;
  do {
    for (unsigned it = 0; it < nit; ++it, ++t) {
      const uint32_t s = d.stream[t];                                // :152
#pragma unroll
      for (int c = 0; c < kPf; ++c) {
        const int64_t j = lane + (int64_t)c * kWave;
        if (j < p) xs[j] = xn[c];
      }
      for (int64_t j = lane + (int64_t)kPf * kWave; j < p; j += kWave) xs[j] = d.xd[(int64_t)s * p + j];
      if (t + 1 < t_end) {
        const uint32_t s1 = d.stream[t + 1];
#pragma unroll
        for (int c = 0; c < kPf; ++c) {
          const int64_t j = lane + (int64_t)c * kWave;
          xn[c] = j < p ? d.xd[(int64_t)s1 * p + j] : 0.0;
        }
      }
      wave_sync();

      for (int k = lane; k < K; k += kWave) {                        // :154
        double acc = 0.0;
        for (int64_t j = 0; j < p; ++j) acc += w[k + j * K] * xs[j];
        slp[k] = acc * wscale + d.b[k];
      }
      wave_sync();

      for (int k = lane; k < K; k += kWave) {                        // :156-159
        const double g = family_gradient_k(d.family, K, k, slp, d.y + (int64_t)s * d.Ky);
        const int64_t mi = k + (int64_t)s * K;
        sgc[k] = g - d.M[mi];
        d.M[mi] = g;
      }

      if (wscale < kSmall) {                                         // :162-166
        for (int64_t i = lane; i < KP; i += kWave) w[i] *= wscale;
        wscale = 1.0;
      }
      wscale *= wscale_update;                                       // :168
      wave_sync();

      if (d.fit_intercept) {                                         // :170-173
        for (int k = lane; k < K; k += kWave) {
          const double gck = sgc[k] / n_d;
          const double gbk = d.gb[k] + gck;
          d.gb[k] = gbk;
          d.b[k] -= gamma * (gbk + gck);
        }
      }

      {
        const double f = gamma / wscale;
        for (int64_t j = lane; j < p; j += kWave) {
          const double xj = xs[j];
          double* wj = w + j * K;
          double* gj = G + j * K;
          for (int k = 0; k < K; ++k) wj[k] -= sgc[k] * xj * f;      // :176
          penalty_apply(penalty, K, wj, gj, wscale, 1.0, gamma, beta);  // :179-180
          for (int k = 0; k < K; ++k) gj[k] += sgc[k] * xj / n_d;    // :183
        }
      }
      wave_sync();
    }

    for (int64_t i = lane; i < KP; i += kWave) w[i] *= wscale;       // :188-189
    wscale = 1.0;
    wave_sync();

    converged = convergence_check(w, d.w_prev, KP, ctl.tol, lane);   // :208
    ++it_outer;
    wave_sync();
  } while (!converged && it_outer < ctl.max_epochs);

  if (ctl.use_lds) {
    for (int64_t i = lane; i < KP; i += kWave) {
      d.w[i] = w[i];
      d.G[i] = G[i];
    }
  }
  if (lane == 0) {
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = converged;
  }
}

size_t sparse_exact_lds_bytes(const SagaDev& d, bool stage_state) {
  size_t b = sizeof(double) * 2 * (size_t)d.K;
  if (stage_state) b += sizeof(double) * 2 * (size_t)d.K * (size_t)d.p + sizeof(unsigned) * (size_t)d.p;
  return (b + 15) & ~size_t(15);
}

size_t dense_exact_lds_bytes(const SagaDev& d, bool stage_state) {
  size_t b = sizeof(double) * (2 * (size_t)d.K + (size_t)d.p);
  if (stage_state) b += sizeof(double) * 2 * (size_t)d.K * (size_t)d.p;
  return (b + 15) & ~size_t(15);
}

int launch_sparse_exact(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                        hipStream_t st) {
  SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_sparse_exact_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(saga_sparse_exact_kernel, dim3(1), dim3(kWave), lds_bytes, st, d, lam, ctl);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_dense_exact(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                       hipStream_t st) {
  SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_dense_exact_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(saga_dense_exact_kernel, dim3(1), dim3(kWave), lds_bytes, st, d, lam, ctl);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

}  // namespace sgdnet
