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
#define SGDNET_DET_MATH 1   // bit-identical family gradients (include/sgdnet_detmath.h)
#include "device_math.hpp"

namespace sgdnet {

namespace {

constexpr int kWave = 64;

// The workgroup is ONE wavefront, whose LDS operations execute in program order: when all
// shared state lives in LDS, ordering lanes against each other only needs the compiler to keep
// program order and the LDS queue to drain -- NOT `s_waitcnt vmcnt(0)`, which __syncthreads()
// would add and which would serialise the software-prefetched global loads of the next draws
// into every phase.  With state in global memory the full barrier is kept.
__device__ __forceinline__ void wave_sync(bool lds_only) {
  if (lds_only) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// ConvergenceCheck (src/utils.h:240-262), executed by the whole wave.
__device__ __forceinline__ int convergence_check(const double* w, double* w_prev, int64_t len,
                                                 double tol, int lane) {
  double max_change = 0.0, max_size = 0.0;
  bool finite = true;
  for (int64_t i = lane; i < len; i += kWave) {
    const double v = w[i];
    finite = finite && (fabs(v) <= 1.79769313486231570815e+308);   // fmax below would drop a NaN
    max_change = fmax(max_change, fabs(v - w_prev[i]));
    max_size = fmax(max_size, fabs(v));
    w_prev[i] = v;
  }
  // a run that blew up never satisfies the reference's test (comparisons with NaN are false):
  // it goes on to max_iter and reports return_code 1
  if (__ballot(!finite) != 0ull) return 0;
  max_change = wave_max(max_change);
  max_size = wave_max(max_size);
  const bool all_zero = (max_size == 0.0) && (max_change == 0.0);
  const bool no_change = (max_size != 0.0) && (max_change / max_size <= tol);
  return (all_zero || no_change) ? 1 : 0;
}

}  // namespace

// LDS carve of the sparse kernel (doubles unless noted):
//   [slp K][sgc K][sb K][sgb K][sval 64][LS cache kLsCache][sidx 64 int][w KP][G KP][lag p u32]
constexpr int kLsCache = 2048;

__global__ __launch_bounds__(kWave) void saga_sparse_exact_kernel(SagaDev d, const LamParams* lamp,
                                                                  ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int K = d.K;
  const int64_t p = d.p;
  const int64_t KP = (int64_t)K * p;
  const bool lds_only = ctl.use_lds != 0;

  double* slp = reinterpret_cast<double*>(smem);
  double* sgc = slp + K;
  double* sb = sgc + K;          // intercept, kept on chip for the whole launch
  double* sgb = sb + K;          // g_sum_intercept
  double* sval = sgb + K;        // the drawn row, staged for the ascending-order dot product
  double* sls = sval + kWave;    // first kLsCache entries of lag_scaling
  int* sidx = reinterpret_cast<int*>(sls + kLsCache);
  double* w = d.w;
  double* G = d.G;
  unsigned* lag = d.lag;
  if (ctl.use_lds) {
    w = reinterpret_cast<double*>(sidx + kWave);
    G = w + KP;
    lag = reinterpret_cast<unsigned*>(G + KP);
    for (int64_t i = lane; i < KP; i += kWave) {
      w[i] = d.w[i];
      G[i] = d.G[i];
    }
  }
  const unsigned nit = (unsigned)ctl.nit;
  const double* LS = ctl.LS;
  for (int i = lane; i < kLsCache && i <= (int64_t)nit; i += kWave) sls[i] = LS[i];
  for (int k = lane; k < K; k += kWave) {
    sb[k] = d.b[k];
    sgb[k] = d.gb[k];
  }
  for (int64_t j = lane; j < p; j += kWave) lag[j] = 0u;            // saga-sparse.h:225
  for (int64_t i = lane; i < KP; i += kWave) d.w_prev[i] = w[i];     // :251
  wave_sync(lds_only);

  auto ls_at = [&](unsigned m) -> double { return m < (unsigned)kLsCache ? sls[m] : LS[m]; };

  const int penalty = lamp->penalty;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :234
  const double n_d = d.n_total;
  double wscale = 1.0;                                               // :227

  unsigned it_outer = 0;
  int converged = 0;
  int64_t t = ctl.stream_off;
  const int64_t t_last = ctl.stream_off + (int64_t)ctl.max_epochs * nit - 1;
  auto clampt = [&](int64_t x) { return x < t_last ? x : t_last; };

  // Software pipeline over the known sample order: the stream entry is loaded three draws
  // ahead, the row pointers two ahead, the row (first 64 entries per lane), y and the old
  // gradient memory one ahead -- none of them depends on the solver state.
  uint32_t s1 = d.stream[clampt(t)], s2 = d.stream[clampt(t + 1)], s3 = d.stream[clampt(t + 2)];
  int64_t q0_1 = d.ptr[s1], q1_1 = d.ptr[s1 + 1];
  int64_t q0_2 = d.ptr[s2], q1_2 = d.ptr[s2 + 1];
  int idx_1 = 0;
  double val_1 = 0.0;
  if (q0_1 + lane < q1_1) {
    idx_1 = d.idx[q0_1 + lane];
    val_1 = d.val[q0_1 + lane];
  }
  // lane k carries class k (k < 64; further classes are re-read from memory)
  double y_1 = lane < d.Ky ? d.y[(int64_t)s1 * d.Ky + lane] : 0.0;
  double m_1 = lane < K ? d.M[lane + (int64_t)s1 * K] : 0.0;

  do {
    for (unsigned it = 0; it < nit; ++it, ++t) {
      // ---- rotate the pipeline -------------------------------------------------------
      const uint32_t s = s1;                                         // :261
      const int64_t q0 = q0_1, q1 = q1_1;
      const int idx_c = idx_1;
      const double val_c = val_1;
      const double y_c = y_1;
      const double m_c = m_1;
      s1 = s2;
      s2 = s3;
      s3 = d.stream[clampt(t + 3)];
      q0_1 = q0_2;
      q1_1 = q1_2;
      q0_2 = d.ptr[s2];
      q1_2 = d.ptr[s2 + 1];
      idx_1 = 0;
      val_1 = 0.0;
      if (q0_1 + lane < q1_1) {
        idx_1 = d.idx[q0_1 + lane];
        val_1 = d.val[q0_1 + lane];
      }
      y_1 = lane < d.Ky ? d.y[(int64_t)s1 * d.Ky + lane] : 0.0;
      m_1 = lane < K ? d.M[lane + (int64_t)s1 * K] : 0.0;

      const bool mine = q0 + lane < q1;                              // this lane holds entry q0+lane
      if (mine) {
        sidx[lane] = idx_c;
        sval[lane] = val_c;
      }

      // LaggedUpdate(it_inner): catch-up of the sample's features  :263-272
      if (mine) {
        const int64_t j = idx_c;
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
          lag[j] = it;
        }
      }
      for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
        const int64_t j = d.idx[q];
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
          lag[j] = it;
        }
      }
      wave_sync(lds_only);

      // linear predictor: lane k accumulates in ascending feature order  :274
      const int head = (q1 - q0) < kWave ? (int)(q1 - q0) : kWave;
      for (int k = lane; k < K; k += kWave) {
        double acc = 0.0;
        for (int e = 0; e < head; ++e) acc += sval[e] * w[k + (int64_t)sidx[e] * K];
        for (int64_t q = q0 + kWave; q < q1; ++q) acc += d.val[q] * w[k + (int64_t)d.idx[q] * K];
        slp[k] = acc * wscale + sb[k];
      }
      if (d.standardize) {                                           // :276-277
        wave_sync(lds_only);
        for (int k = 0; k < K; ++k) {
          double part = 0.0;
          for (int64_t j = lane; j < p; j += kWave) part += w[k + j * K] * d.c[j];
          part = wave_sum(part);
          if (lane == 0) slp[k] -= part * wscale;
        }
      }
      wave_sync(lds_only);

      // gradient, gradient memory  :279-282
      const double y_first = __shfl(y_c, 0, kWave);   // single-response families: y of the draw
      for (int k = lane; k < K; k += kWave) {
        const bool first = k < kWave;     // classes >= 64 are not register-carried
        double g;
        if (d.family == SGDNET_MGAUSSIAN)
          g = slp[k] - (first ? y_c : d.y[(int64_t)s * d.Ky + k]);
        else
          g = family_gradient_k(d.family, K, k, slp, &y_first);
        const int64_t mi = k + (int64_t)s * K;
        sgc[k] = g - (first ? m_c : d.M[mi]);
        d.M[mi] = g;
        if (first && s1 == s) m_1 = g;    // the next draw repeats this sample: forward its memory
      }

      // rescale + unlag whenever wscale becomes too small  :285-295
      if (wscale < kSmall) {
        wave_sync(lds_only);
        for (int64_t j = lane; j < p; j += kWave) {
          const unsigned lagged = it - lag[j];
          if (lagged != 0)
            penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
          for (int k = 0; k < K; ++k) w[k + j * K] *= wscale;
          lag[j] = it;
        }
        wscale = 1.0;
      }

      wscale *= wscale_update;                                       // :297
      wave_sync(lds_only);

      if (d.fit_intercept) {                                         // :300-304
        for (int k = lane; k < K; k += kWave) {
          const double gck = sgc[k] / n_d;
          const double gbk = sgb[k] + gck;
          sgb[k] = gbk;
          sb[k] -= gamma * (gbk * 0.01 + gck);
        }
      }

      // AddWeighted(w, ..., -gamma/wscale)  :306-313
      {
        const double scaling = -gamma / wscale;
        if (mine) {
          const int64_t j = idx_c;
          for (int k = 0; k < K; ++k) w[k + j * K] += val_c * sgc[k] * scaling;
        }
        for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const double v = d.val[q];
          for (int k = 0; k < K; ++k) w[k + j * K] += v * sgc[k] * scaling;
        }
        if (d.standardize) {
          wave_sync(lds_only);
          for (int64_t j = lane; j < p; j += kWave) {
            const double cj = d.c[j];
            for (int k = 0; k < K; ++k) w[k + j * K] -= cj * sgc[k] * scaling;
          }
          wave_sync(lds_only);
        }
      }

      // LaggedUpdate(it_inner + 1): the SAGA step on the sample's features  :316-325,
      // then AddWeighted(g_sum, ..., 1/n)  :328-335 (same lane, same feature)
      {
        const double scaling = 1.0 / n_d;
        if (mine) {
          const int64_t j = idx_c;
          const unsigned lagged = (it + 1) - lag[j];
          if (lagged != 0) {
            penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
            lag[j] = it + 1;
          }
          for (int k = 0; k < K; ++k) G[k + j * K] += val_c * sgc[k] * scaling;
        }
        for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const unsigned lagged = (it + 1) - lag[j];
          if (lagged != 0) {
            penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
            lag[j] = it + 1;
          }
          const double v = d.val[q];
          for (int k = 0; k < K; ++k) G[k + j * K] += v * sgc[k] * scaling;
        }
        if (d.standardize) {
          wave_sync(lds_only);
          for (int64_t j = lane; j < p; j += kWave) {
            const double cj = d.c[j];
            for (int k = 0; k < K; ++k) G[k + j * K] -= cj * sgc[k] * scaling;
          }
        }
      }
      wave_sync(lds_only);
    }

    // Reset(n_samples): unlag and rescale  :340-348
    for (int64_t j = lane; j < p; j += kWave) {
      const unsigned lagged = nit - lag[j];
      if (lagged != 0)
        penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
      for (int k = 0; k < K; ++k) w[k + j * K] *= wscale;
      lag[j] = 0u;
    }
    wscale = 1.0;
    wave_sync(lds_only);

    converged = convergence_check(w, d.w_prev, KP, ctl.tol, lane);   // :367
    ++it_outer;
    wave_sync(lds_only);
  } while (!converged && it_outer < ctl.max_epochs);                 // :371

  if (ctl.use_lds) {
    for (int64_t i = lane; i < KP; i += kWave) {
      d.w[i] = w[i];
      d.G[i] = G[i];
    }
  }
  for (int k = lane; k < K; k += kWave) {
    d.b[k] = sb[k];
    d.gb[k] = sgb[k];
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
  const bool lds_only = ctl.use_lds != 0;

  // LDS carve: [slp K][sgc K][sb K][sgb K][sy Ky][xs p][w KP][G KP]
  // (intercept and g_sum_intercept live in LDS for the whole launch and the next sample's
  // response and gradient memory are requested one iteration ahead, as in the sparse kernel:
  // with them in global memory every iteration was a chain of three dependent round trips)
  double* slp = reinterpret_cast<double*>(smem);
  double* sgc = slp + K;
  double* sb = sgc + K;
  double* sgb = sb + K;
  double* sy = sgb + K;
  double* xs = sy + d.Ky;
  const bool small_k = K <= kWave && d.Ky <= kWave;       // lane k owns class k in the prefetch
  for (int k = lane; k < K; k += kWave) {
    sb[k] = d.b[k];
    sgb[k] = d.gb[k];
  }
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
  wave_sync(lds_only);

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
  double yn = 0.0, mn = 0.0;
  {
    const uint32_t s0 = d.stream[t];
#pragma unroll
    for (int c = 0; c < kPf; ++c) {
      const int64_t j = lane + (int64_t)c * kWave;
      xn[c] = j < p ? d.xd[(int64_t)s0 * p + j] : 0.0;
    }
    if (small_k) {
      yn = lane < d.Ky ? d.y[(int64_t)s0 * d.Ky + lane] : 0.0;
      mn = lane < K ? d.M[lane + (int64_t)s0 * K] : 0.0;
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
      const double m_cur = mn;
      if (small_k && lane < d.Ky) sy[lane] = yn;
      const bool has_next = t + 1 < t_end;
      uint32_t s1 = s;
      if (has_next) {
        s1 = d.stream[t + 1];
#pragma unroll
        for (int c = 0; c < kPf; ++c) {
          const int64_t j = lane + (int64_t)c * kWave;
          xn[c] = j < p ? d.xd[(int64_t)s1 * p + j] : 0.0;
        }
        if (small_k) {
          yn = lane < d.Ky ? d.y[(int64_t)s1 * d.Ky + lane] : 0.0;
          mn = lane < K ? d.M[lane + (int64_t)s1 * K] : 0.0;     // stale if s1 == s: replaced below
        }
      }
      wave_sync(lds_only);

      for (int k = lane; k < K; k += kWave) {                        // :154
        double acc = 0.0;
        for (int64_t j = 0; j < p; ++j) acc += w[k + j * K] * xs[j];
        slp[k] = acc * wscale + sb[k];
      }
      wave_sync(lds_only);

      for (int k = lane; k < K; k += kWave) {                        // :156-159
        const double g = family_gradient_k(d.family, K, k, slp, small_k ? sy : d.y + (int64_t)s * d.Ky);
        const int64_t mi = k + (int64_t)s * K;
        sgc[k] = g - (small_k ? m_cur : d.M[mi]);
        d.M[mi] = g;
        if (small_k && has_next && s1 == s) mn = g;                  // k == lane here
      }

      if (wscale < kSmall) {                                         // :162-166
        for (int64_t i = lane; i < KP; i += kWave) w[i] *= wscale;
        wscale = 1.0;
      }
      wscale *= wscale_update;                                       // :168
      wave_sync(lds_only);

      if (d.fit_intercept) {                                         // :170-173
        for (int k = lane; k < K; k += kWave) {
          const double gck = sgc[k] / n_d;
          const double gbk = sgb[k] + gck;
          sgb[k] = gbk;
          sb[k] -= gamma * (gbk + gck);
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
      wave_sync(lds_only);
    }

    for (int64_t i = lane; i < KP; i += kWave) w[i] *= wscale;       // :188-189
    wscale = 1.0;
    wave_sync(lds_only);

    converged = convergence_check(w, d.w_prev, KP, ctl.tol, lane);   // :208
    ++it_outer;
    wave_sync(lds_only);
  } while (!converged && it_outer < ctl.max_epochs);

  if (ctl.use_lds) {
    for (int64_t i = lane; i < KP; i += kWave) {
      d.w[i] = w[i];
      d.G[i] = G[i];
    }
  }
  for (int k = lane; k < K; k += kWave) {
    d.b[k] = sb[k];
    d.gb[k] = sgb[k];
  }
  if (lane == 0) {
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = converged;
  }
}

// --------------------------------------------------------------------------
// Dense variant for SMALL problems (the reference's own data sets: iris, abalone, heart, wine,
// student -- K*p <= 64 coefficients, a few thousand samples): the same iteration in the same
// arithmetic order as saga_dense_exact_kernel, restructured around what bounds a one-wavefront
// sequential kernel -- latency:
//   * lane l = k + j*K owns coefficient (k, j): w and g_sum live in REGISTERS for the whole launch;
//   * g_memory, the response and the epoch's draws live in LDS (no global round trip inside an
//     iteration); the sample's row is requested kPf iterations ahead into a register ring;
//   * x_j reaches the class lanes by v_readlane (uniform source lane), the per-class gradient change
//     goes back by one shuffle; the only LDS traffic on the dependent chain is the K*p-entry copy of
//     w that the class lanes read for the linear predictor.
// One epoch of abalone (4177 x 9, gaussian): 9.1 ms with saga_dense_exact_kernel, ~0.6 ms here.
// --------------------------------------------------------------------------
constexpr int kSmallPf = 8;

__device__ __forceinline__ double readlane_d(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// kFamily / kPenalty: compile-time copies of d.family / the lambda's penalty, kK1: one class.  A single
// wavefront is bound by the number of instructions it has to issue per iteration (the generic body
// came to ~900, most of them the untaken branches of other families and penalties): every
// combination gets its own straight-line loop.
template <int kFamily, int kPenalty, bool kK1>
__global__ __launch_bounds__(kWave) void saga_dense_exact_small_kernel(SagaDev d, const LamParams* lamp,
                                                                       ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int K = kK1 ? 1 : d.K, Ky = kK1 ? 1 : d.Ky;
  const int p = (int)d.p;
  const int KP = K * p;
  const int n = (int)d.n;
  const unsigned nit = (unsigned)ctl.nit;
  // LDS carve: [Ml K*n][yl Ky*n][wl 64][slp 16][sgc 16][vl 64][sl nit + kSmallPf (uint32)]
  double* Ml = reinterpret_cast<double*>(smem);
  double* yl = Ml + (size_t)K * n;
  double* wl = yl + (size_t)Ky * n;
  double* slp = wl + kWave;
  double* sgcl = slp + 16;
  double* vl = sgcl + 16;
  uint32_t* sl = reinterpret_cast<uint32_t*>(vl + kWave);

  const bool active = lane < KP;
  const bool cls = lane < K;
  const int k = lane % K;
  const int j = lane / K;
  for (int i = lane; i < K * n; i += kWave) Ml[i] = d.M[i];
  for (int i = lane; i < Ky * n; i += kWave) yl[i] = d.y[i];
  double w = active ? d.w[lane] : 0.0;
  double G = active ? d.G[lane] : 0.0;
  double w_prev = w;                                                   // saga-dense.h:142
  double sb = cls ? d.b[lane] : 0.0;
  double sgb = cls ? d.gb[lane] : 0.0;
  if (lane < kWave) wl[lane] = w;
  if (active) d.w_prev[lane] = w;

  constexpr int penalty = kPenalty;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                    // :131
  const double n_d = d.n_total;
  double wscale = 1.0;                                                 // :129
  unsigned it_outer = 0;
  int converged = 0;
  int64_t t = ctl.stream_off;
  const int64_t t_end = ctl.stream_off + (int64_t)ctl.max_epochs * nit;

  do {
    // this epoch's draws (and the first kSmallPf of the next, for the row prefetch) into LDS
    const int64_t avail = t_end - t;
    const int want = (int)((int64_t)nit + kSmallPf < avail ? (int64_t)nit + kSmallPf : avail);
    wave_sync(true);
    for (int i = lane; i < want; i += kWave) sl[i] = d.stream[t + i];
    wave_sync(true);
    // register ring of the next kSmallPf rows: lane (k, j) holds x_j of each
    double xr[kSmallPf];
#pragma unroll
    for (int u = 0; u < kSmallPf; ++u) {
      const uint32_t su = (u < want) ? sl[u] : 0u;
      xr[u] = (active && u < want) ? d.xd[(int64_t)su * p + j] : 0.0;
    }
    // the current draw's sample, gradient memory and response are read from LDS one iteration ahead
    // (after the previous iteration's store to the gradient memory: LDS operations of a wavefront
    // execute in order, so a sample drawn twice in a row reads what was just written)
    uint32_t s = sl[0];
    double m_old = cls ? Ml[k + (size_t)s * K] : 0.0;
    double y_cur = yl[(size_t)s * Ky + ((Ky > 1 && cls) ? k : 0)];
    for (unsigned it0 = 0; it0 < nit; it0 += kSmallPf) {
#pragma unroll
      for (int u = 0; u < kSmallPf; ++u) {
        const unsigned it = it0 + u;
        if (it >= nit) break;
        const uint32_t s_next = sl[it + 1 < (unsigned)want ? it + 1 : it];   // :152 of the next iteration
        const double x = xr[u];
        {  // the row of iteration it + kSmallPf
          const int nx = (int)it + kSmallPf;
          const uint32_t sn = nx < want ? sl[nx] : 0u;
          xr[u] = (active && nx < want) ? d.xd[(int64_t)sn * p + j] : 0.0;
        }
        // ---- linear predictor on the class lanes: ascending-feature sum, :154 ----
        // (eight coefficients are read from LDS together, then added in order)
        double acc = 0.0;
        if (kK1) {
          // one class: lane j holds w_j and x_j, the ascending sum runs over v_readlane operands
          for (int jj = 0; jj < p; ++jj) acc += readlane_d(w, jj) * readlane_d(x, jj);
        } else {
          for (int j0 = 0; j0 < p; j0 += 8) {
            double wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[e] = wl[(j0 + e < p) ? k + (j0 + e) * K : 0];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int jj = (j0 + e < p) ? j0 + e : 0;
              const double term = wv[e] * readlane_d(x, jj * K);
              acc = (j0 + e < p) ? acc + term : acc;
            }
          }
        }
        const double lp = acc * wscale + sb;
        double g = 0.0;
        if (kFamily == SGDNET_GAUSSIAN) {
          g = lp - y_cur;
        } else if (kFamily == SGDNET_BINOMIAL) {
          g = 1.0 - y_cur - 1.0 / (1.0 + SGD_EXP(lp));
        } else if (kFamily == SGDNET_MGAUSSIAN) {
          g = lp - (cls ? y_cur : 0.0);
        } else {
          if (cls) slp[lane] = lp;
          wave_sync(true);
          if (cls) g = family_gradient_k(SGDNET_MULTINOMIAL, K, k, slp, &y_cur);
        }
        double gc = 0.0;
        if (cls) {                                                      // :156-159
          gc = g - m_old;
          Ml[k + (size_t)s * K] = g;
        }
        if (wscale < kSmall) {                                          // :162-166
          w *= wscale;
          wscale = 1.0;
        }
        wscale *= wscale_update;                                        // :168
        if (d.fit_intercept && cls) {                                   // :170-173
          const double gck = gc / n_d;
          const double gbk = sgb + gck;
          sgb = gbk;
          sb -= gamma * (gbk + gck);
        }
        // ---- all coefficients: gradient step, penalty, gradient average (:176-183) ----
        // class k's change on every (k, j) lane
        const double gck = kK1 ? readlane_d(gc, 0) : __shfl(gc, k, kWave);
        const double f = gamma / wscale;
        const double f2 = f;        // penalties.h: (gamma / w_scale) * scaling with scaling == 1.0: the same double
        w -= gck * x * f;                                               // :176
        if (penalty == SGDNET_RIDGE) {                                  // penalty(w, j, wscale, 1.0, g_sum), :179-180
          w -= f2 * G;
        } else if (penalty == SGDNET_ELASTICNET) {
          const double tau = beta * gamma * 1.0 / wscale;
          w = soft_threshold(w - f2 * G, tau);
        } else {
          const double v = w - f2 * G;
          vl[lane] = v;
          wave_sync(true);
          double nrm = 0.0;
          for (int kk = 0; kk < K; ++kk) {
            const double vv = vl[(active ? j : 0) * K + kk];
            nrm += vv * vv;
          }
          nrm = sqrt(nrm);
          const double factor = beta * gamma * 1.0 / nrm;
          w = factor < 1.0 ? v * (1.0 - factor / wscale) : 0.0;
          wave_sync(true);
        }
        G += gck * x / n_d;                                             // :183
        if (!active) { w = 0.0; G = 0.0; }
        if (!kK1) wl[lane] = w;
        // next iteration's operands (behind this iteration's gradient-memory store)
        s = s_next;
        m_old = cls ? Ml[k + (size_t)s * K] : 0.0;
        y_cur = yl[(size_t)s * Ky + ((Ky > 1 && cls) ? k : 0)];
        wave_sync(true);
      }
    }
    t += nit;
    w *= wscale;                                                        // :188-189
    wscale = 1.0;
    wl[lane] = w;
    // ConvergenceCheck (src/utils.h:240-262) on the registers
    {
      const bool finite = fabs(w) <= 1.79769313486231570815e+308;
      const double mc = wave_max(fabs(w - w_prev));
      const double ms = wave_max(fabs(w));
      w_prev = w;
      const bool all_zero = (ms == 0.0) && (mc == 0.0);
      const bool no_change = (ms != 0.0) && (mc / ms <= ctl.tol);
      converged = (__ballot(!finite) == 0ull) && (all_zero || no_change) ? 1 : 0;
    }
    ++it_outer;
  } while (!converged && it_outer < ctl.max_epochs);

  wave_sync(true);
  if (active) {
    d.w[lane] = w;
    d.G[lane] = G;
    d.w_prev[lane] = w_prev;
  }
  if (cls) {
    d.b[lane] = sb;
    d.gb[lane] = sgb;
  }
  for (int i = lane; i < K * n; i += kWave) d.M[i] = Ml[i];
  if (lane == 0) {
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = converged;
  }
}

// 0 when the small-problem kernel does not apply
size_t dense_exact_small_lds_bytes(const SagaDev& d, int64_t nit) {
  if (!d.xd || d.K > 16 || (int64_t)d.K * d.p > kWave || d.n > (1 << 20) || nit > (1 << 16)) return 0;
  const size_t b = sizeof(double) * ((size_t)d.K * d.n + (size_t)d.Ky * d.n + 2 * kWave + 32) +
                   sizeof(uint32_t) * ((size_t)nit + kSmallPf);
  return b <= 150 * 1024 ? ((b + 15) & ~size_t(15)) : 0;
}

template <int kFamily, int kPenalty, bool kK1>
static int launch_small_t(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, hipStream_t st) {
  SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_dense_exact_small_kernel<kFamily, kPenalty, kK1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((saga_dense_exact_small_kernel<kFamily, kPenalty, kK1>), dim3(1), dim3(kWave), lds_bytes, st, d, lam,
                     ctl);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_dense_exact_small(const SagaDev& d, int penalty, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                             hipStream_t st) {
#define SGD_SMALL(F, P, K1) return launch_small_t<F, P, K1>(d, lam, ctl, lds_bytes, st)
  if (d.K == 1) {
    if (d.family == SGDNET_GAUSSIAN) {
      if (penalty == SGDNET_RIDGE) SGD_SMALL(SGDNET_GAUSSIAN, SGDNET_RIDGE, true);
      SGD_SMALL(SGDNET_GAUSSIAN, SGDNET_ELASTICNET, true);
    }
    if (d.family == SGDNET_BINOMIAL) {
      if (penalty == SGDNET_RIDGE) SGD_SMALL(SGDNET_BINOMIAL, SGDNET_RIDGE, true);
      SGD_SMALL(SGDNET_BINOMIAL, SGDNET_ELASTICNET, true);
    }
  }
  if (d.family == SGDNET_MULTINOMIAL) {
    if (penalty == SGDNET_RIDGE) SGD_SMALL(SGDNET_MULTINOMIAL, SGDNET_RIDGE, false);
    if (penalty == SGDNET_ELASTICNET) SGD_SMALL(SGDNET_MULTINOMIAL, SGDNET_ELASTICNET, false);
    SGD_SMALL(SGDNET_MULTINOMIAL, SGDNET_GROUPLASSO, false);
  }
  if (d.family == SGDNET_MGAUSSIAN) {
    if (penalty == SGDNET_RIDGE) SGD_SMALL(SGDNET_MGAUSSIAN, SGDNET_RIDGE, false);
    if (penalty == SGDNET_ELASTICNET) SGD_SMALL(SGDNET_MGAUSSIAN, SGDNET_ELASTICNET, false);
    SGD_SMALL(SGDNET_MGAUSSIAN, SGDNET_GROUPLASSO, false);
  }
#undef SGD_SMALL
  set_error("small dense exact kernel: unsupported family / class count");
  return SGDNET_EUNSUPPORTED;
}

size_t sparse_exact_lds_bytes(const SagaDev& d, bool stage_state) {
  size_t b = sizeof(double) * (4 * (size_t)d.K + kWave + kLsCache) + sizeof(int) * kWave;
  if (stage_state) b += sizeof(double) * 2 * (size_t)d.K * (size_t)d.p + sizeof(unsigned) * (size_t)d.p;
  return (b + 15) & ~size_t(15);
}

size_t dense_exact_lds_bytes(const SagaDev& d, bool stage_state) {
  size_t b = sizeof(double) * (4 * (size_t)d.K + (size_t)d.Ky + (size_t)d.p);
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
