// Exact-order SAGA epoch kernels for gfx950 (the parity anchor).
//
// One 64-lane wavefront executes the reference iteration one draw at a time, in
// the order of the resident sample stream, so results agree with the CPU
// restatement to rounding of libm calls only:
//   saga_sparse_exact_kernel  <-  Saga(), src/saga-sparse.h:194-383
//                                 (LaggedUpdate :76-100, AddWeighted :114-130, Reset :132-155)
//   saga_dense_exact_kernel   <-  Saga(), src/saga-dense.h:99-224
// The chain through `intercept` makes every pair of consecutive iterations
// dependent, so these kernels are latency-bound by construction;
// the throughput path is saga_batched.hip.  Sparse x with one response has two
// faster forms of the same iteration, bit for bit (round 3):
//   saga_sparse_exact_k1x_kernel  one producer + one consumer wavefront, the drawn
//                                 row held in registers for the whole draw
//   saga_sparse_exact_k1m_kernel  two producers + six consumers taking the draws
//                                 round robin; registration of a draw's features
//                                 and the intercept chain run in draw order
// and for several classes the general iteration runs on eight wavefronts at once
// under the same protocol (saga_sparse_exact_mc_kernel).
//
// Lane roles: lanes stride the nonzeros of the drawn sample for the per-feature
// steps; lane k owns class k for the linear predictor / gradient / intercept.
// w, g_sum and lag are staged in LDS when they fit (160 KiB per CU).
#define SGDNET_DET_MATH 1   // bit-identical family gradients (include/sgdnet_detmath.h)
#include "device_math.hpp"

namespace sgdnet {

namespace {

constexpr int kWave = 64;

// Pointers into the LDS carry their address space in the type, and a kernel whose state lives either in the LDS or
// in memory is compiled once for each (StatePtr<kLds>): a pointer that may be either becomes a FLAT access, which
// waits for the vector-memory AND the LDS counter to drain -- i.e. for every prefetch and store in flight.
#define SGD_LDS(T) __attribute__((address_space(3))) T
template <bool kLds> struct K1State { using D = double*; using U = unsigned*; };
template <> struct K1State<true> { using D = SGD_LDS(double)*; using U = SGD_LDS(unsigned)*; };

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

// Lanes of ONE wavefront exchanging data through the LDS: the hardware runs the wavefront's LDS operations in
// program order, the compiler only has to emit them in that order (it reasons per thread and would otherwise
// forward a lane's own store to its next load, or drop a store that the same lane overwrites later).
__device__ __forceinline__ void lanes_publish() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ double readlane_d(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// a / n for a constant n (an integer count held as a double), rn = RN(1 / n): quotient estimate, exact
// remainder, one correction (Markstein) -- three instructions instead of the ~30 of an IEEE division
// sequence.  Correctly rounded whenever the remainder does not underflow; the wide kernel's parity
// is to rounding, not bit for bit.
__device__ __forceinline__ int64_t readlane_ll(int64_t v, int src) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
  return ((int64_t)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ double div_by_n(double a, double n, double rn) {
  const double q0 = a * rn;
  const double rem = __builtin_fma(-q0, n, a);
  return __builtin_fma(rem, rn, q0);
}

// the same, for the kernels that promise the reference's bits: below 1e-280 the remainder could underflow
// and the plain division is used (elsewhere the result IS the correctly rounded quotient: rn is the correctly
// rounded reciprocal of an integer-valued n, q0 is within an ulp, the remainder is exact)
__device__ __forceinline__ double div_by_n_exact(double a, double n, double rn) {
  double q = div_by_n(a, n, rn);
  q = a == 0.0 ? a : q;                                   // a zero keeps its sign, as a / n does
  // (lanes that hold no coefficient carry a = 0 through here every iteration: tested per lane, they dragged the
  // whole wavefront through the division sequence -- ~300 cycles of a lone wavefront's 1500 per iteration)
  const bool tiny = fabs(a) < 1e-280 && a != 0.0;
  if (__ballot(tiny) != 0ull) q = tiny ? a / n : q;
  return q;
}

// Multinomial gradient of class `lane` (families.h:244-260, math.h:25-33) when lane k < K holds the linear
// predictor of class k: one exp per class LANE instead of K + 1 per lane -- the other classes' terms arrive by
// v_readlane and are added in ascending class order, so every intermediate is the double the reference forms.
__device__ __forceinline__ double softmax_gradient_lanes(double lp, int K, int lane, double y_label) {
  const bool cls = lane < K;
  double mx = readlane_d(lp, 0);
  for (int kk = 1; kk < K; ++kk) {
    const double v = readlane_d(lp, kk);
    mx = v > mx ? v : mx;
  }
  const double e = SGD_EXP((cls ? lp : mx) - mx);
  double se = 0.0;
  for (int kk = 0; kk < K; ++kk) se += readlane_d(e, kk);
  const double lse = SGD_LOG(se) + mx;
  double g = SGD_EXP((cls ? lp : lse) - lse);
  if ((unsigned)lane == (unsigned)(y_label + 0.5)) g -= 1.0;
  return g;
}

// ConvergenceCheck (src/utils.h:240-262), executed by the whole wave.
template <typename WP>
__device__ __forceinline__ int convergence_check(WP w, double* w_prev, int64_t len, double tol, int lane) {
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

template <bool kLds>
__global__ __launch_bounds__(kWave) void saga_sparse_exact_kernel(SagaDev d, const LamParams* lamp,
                                                                  ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int K = d.K;
  const int64_t p = d.p;
  const int64_t KP = (int64_t)K * p;
  constexpr bool lds_only = kLds;

  SGD_LDS(double)* slp = (SGD_LDS(double)*)smem;
  SGD_LDS(double)* sgc = slp + K;
  SGD_LDS(double)* sb = sgc + K;          // intercept, kept on chip for the whole launch
  SGD_LDS(double)* sgb = sb + K;          // g_sum_intercept
  SGD_LDS(double)* sval = sgb + K;        // the drawn row, staged for the ascending-order dot product
  SGD_LDS(double)* sls = sval + kWave;    // first kLsCache entries of lag_scaling
  SGD_LDS(int)* sidx = (SGD_LDS(int)*)(sls + kLsCache);
  typename K1State<kLds>::D w;
  typename K1State<kLds>::D G;
  typename K1State<kLds>::U lag;
  if constexpr (kLds) {
    w = (SGD_LDS(double)*)(sidx + kWave);
    G = w + KP;
    lag = (SGD_LDS(unsigned)*)(G + KP);
    for (int64_t i = lane; i < KP; i += kWave) {
      w[i] = d.w[i];
      G[i] = d.G[i];
    }
  } else {
    w = d.w;
    G = d.G;
    lag = d.lag;
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

  auto ls_at = [&](unsigned m) -> double {   // (not a select between an LDS and a global address)
    double v = sls[m < (unsigned)kLsCache ? m : 0u];
    if (m >= (unsigned)kLsCache) v = LS[m];
    return v;
  };

  const int penalty = lamp->penalty;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :234
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
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
      const double q_before = gamma / wscale;     // gamma / w_scale of every penalty call until the scale moves
      if (mine) {
        const int64_t j = idx_c;
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_before, gamma, beta);
          lag[j] = it;
        }
      }
      for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
        const int64_t j = d.idx[q];
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_before, gamma, beta);
          lag[j] = it;
        }
      }
      wave_sync(lds_only);

      // linear predictor: lane k accumulates in ascending feature order  :274
      const int head = (q1 - q0) < kWave ? (int)(q1 - q0) : kWave;
      if (K == 1 && q1 - q0 <= kWave) {
        // one response, the whole row in the lanes: every lane forms its product (it caught up its own
        // feature just above), the ascending sum runs over v_readlane operands -- product rounded, then
        // added, the reference's operations -- instead of one lane walking the row through LDS
        const double wx = mine ? val_c * w[idx_c] : 0.0;
        double acc = 0.0;
        for (int e = 0; e < head; ++e) acc += readlane_d(wx, e);
        if (lane == 0) slp[0] = acc * wscale + sb[0];
      } else
      for (int k = lane; k < K; k += kWave) {
        double acc = 0.0;
        // (the coefficients of eight entries are requested together, then added in order: one at a time, every
        // entry paid a round trip of its own)
        for (int e0 = 0; e0 < head; e0 += 8) {
          double wv[8], xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ee = e0 + e < head ? e0 + e : head - 1;
            xv[e] = sval[ee];
            wv[e] = w[k + (int64_t)sidx[ee] * K];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e0 + e < head) acc += xv[e] * wv[e];
        }
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
      if (d.family == SGDNET_MULTINOMIAL && K <= kWave) {
        const double g = softmax_gradient_lanes(lane < K ? slp[lane] : 0.0, K, lane, y_first);
        if (lane < K) {
          const int64_t mi = lane + (int64_t)s * K;
          sgc[lane] = g - m_c;
          d.M[mi] = g;
          if (s1 == s) m_1 = g;
        }
      } else
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
      const double q_after = gamma / wscale;
      wave_sync(lds_only);

      if (d.fit_intercept) {                                         // :300-304
        for (int k = lane; k < K; k += kWave) {
          const double gck = div_by_n_exact(sgc[k], n_d, rn_d);
          const double gbk = sgb[k] + gck;
          sgb[k] = gbk;
          sb[k] -= gamma * (gbk * 0.01 + gck);
        }
      }

      // AddWeighted(w, ..., -gamma/wscale)  :306-313
      {
        const double scaling = -q_after;            // -gamma / wscale: the quotient's sign flipped, the same double
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
            penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_after, gamma, beta);
            lag[j] = it + 1;
          }
          for (int k = 0; k < K; ++k) G[k + j * K] += val_c * sgc[k] * scaling;
        }
        for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const unsigned lagged = (it + 1) - lag[j];
          if (lagged != 0) {
            penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_after, gamma, beta);
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

  if constexpr (kLds) {
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
// One response, explicit x (no implicit centring): the register-resident iteration (round 3).
//
// The general kernel above pays a memory round trip per phase of a draw -- lag, then w and g_sum, the
// store, w again for the dot product, w for AddWeighted, lag / w / g_sum for the SAGA step: ~7 dependent
// trips of 300-400 ns through L2 when the state does not fit the LDS.  Here lane e owns entry e of the drawn
// row for the WHOLE draw: it is handed w[j], g_sum[j], lag[j] and the lag scaling of its feature at the top
// of the iteration, runs catch-up (:263-272), its product of the ascending sum (:274), AddWeighted
// (:306-313), the SAGA step (:316-325) and the g_sum update (:328-335) on registers -- the same operations
// on the same doubles in the same order per feature, so the state stays bit for bit -- and stores once.
// w / g_sum / lag of the next draw are requested a draw ahead, before this draw's stores; a feature the two
// draws share is forwarded from its lane's registers (an LDS byte table, feature -> lane, finds it; the stores
// of earlier draws are older memory operations of this wavefront and are seen in order).  Rows longer than
// the wavefront, the `wscale < SMALL` reset and the epoch end work on memory, after which the one-ahead state
// is fetched again.
//
// The requests that do not depend on the solver are taken off the wavefront that iterates: a workgroup of TWO
// wavefronts.  Wavefront 1 (the producer) walks the sample stream ahead of the solver: stream entries, row
// pointers, responses and gradient memory are gathered sixteen draws at a time (one lane per draw), rows four
// draws at a time, and every draw's data-independent numbers -- w_scale before and after the draw and the two
// quotients by it, gamma / w_scale and the SAGA step's soft threshold (:234, :285-297; the sequence restarts
// every epoch) -- are formed there; each draw becomes one slot of an LDS ring.  Wavefront 0 (the consumer) runs
// the iteration of the kernel above from those slots: its only memory traffic is w / g_sum / lag of the next
// draw (requested a draw ahead, forwarded where two draws share a feature) and the stores, so its counter never
// queues behind a request to HBM.  (Measured, DESIGN.md 4.1: one wavefront doing everything issued 436 instructions
// and took 3900 cycles per draw, half of them instruction issue -- a lone wavefront retires one every ~4.5 cycles,
// an f64 one every 8 -- and half waits; with the producer the consumer never waits for a slot, 22 polls in 400 000
// draws, and the ring alone runs at 0.38 us per draw.)
// Gradient memory: the producer's copy of M[s] may predate the consumer's store for a draw of the same sample
// at most kRing + 32 draws back; the consumer keeps the last 64 (sample, gradient) pairs in registers, one per
// lane, and takes the latest match instead.
// --------------------------------------------------------------------------
constexpr int kOwnSlots = 8192;                                   // bytes: feature (hashed) -> lane of the draw in hand
constexpr int kK1Sum = kWave + 16;                                // products of the draw in hand, padded to whole blocks
constexpr size_t kK1FixedLds = sizeof(double) * (kK1Sum + 128) + kOwnSlots;

constexpr int kRing = 16;                                          // slots (a power of two)
constexpr int kSlotHdr = 10;                                       // doubles per slot header
// Every wait in these kernels gives up after kSpinLimitTicks of the 100 MHz wall clock (an exit every wavefront
// reaches) -- by the clock, not by a poll count: a wait may legitimately span another wavefront's serial pass over
// K * p entries (ConvergenceCheck, the rescale sweep), and a count of 2^24 polls (~1 s) could expire on ~1e8 of them.
constexpr long long kSpinLimitTicks = 3000000000ll;                 // 30 s
__device__ __forceinline__ bool spin_expired(unsigned& spins, long long& t0) {
  if ((++spins & 4095u) != 0u) return false;
  const long long now = wall_clock64();
  if (t0 == 0) {
    t0 = now;
    return false;
  }
  return now - t0 > kSpinLimitTicks;
}
constexpr size_t kK1xFixedLds = kK1FixedLds + kRing * (sizeof(int) * kWave + sizeof(double) * (kWave + kSlotHdr)) + 32;

// SoftThreshold (prox.h:32-39) of the register-resident kernels.  For s >= 0 (and for NaN operands)
// max(x - s, 0) - max(-x - s, 0) is |x| - s carrying x's sign where that is positive, +0 otherwise: the same double
// from one f64 operation instead of five (a lone wavefront pays 8 cycles for each); `plain` keeps the reference form.
__device__ __forceinline__ double k1_soft(double x, double s, bool plain) {
  if (plain) return soft_threshold(x, s);
  const double t = fabs(x) - s;
  return t > 0.0 ? copysign(t, x) : 0.0;
}

SGD_DEFINE_EXP(sgd_exp_lds, const SGD_LDS(double)*)
SGD_DEFINE_LOG(sgd_log_lds, const SGD_LDS(double)*)

// softmax_gradient_lanes with the exp / log tables in the LDS (the same functions on copies of the same tables: the
// same bits): in the multi-wavefront kernel the three table lookups sit on the serial path, and as global loads
// they queue behind every request the wavefront has in flight
__device__ __forceinline__ double softmax_gradient_lanes_lds(double lp, int K, int lane, double y_label,
                                                             const SGD_LDS(double)* etab, const SGD_LDS(double)* ltab) {
  const bool cls = lane < K;
  double mx = readlane_d(lp, 0);
  for (int kk = 1; kk < K; ++kk) {
    const double v = readlane_d(lp, kk);
    mx = v > mx ? v : mx;
  }
  const double e = sgd_exp_lds((cls ? lp : mx) - mx, etab);
  double se = 0.0;
  for (int kk = 0; kk < K; ++kk) se += readlane_d(e, kk);
  const double lse = sgd_log_lds(se, ltab) + mx;
  double g = sgd_exp_lds((cls ? lp : lse) - lse, etab);
  if ((unsigned)lane == (unsigned)(y_label + 0.5)) g -= 1.0;
  return g;
}

// ... and with the 64-entry table spread over the lanes of the wavefront (entry j in lane j): a lookup is four
// v_readlane instead of an LDS round trip -- for the serial part of the multi-consumer kernel.  The argument is the
// same in every lane there, so the index is too.
struct LaneExpTab {
  double hi, lo;
  __device__ __forceinline__ double operator[](int i) const {
    const int jj = __builtin_amdgcn_readfirstlane(i >> 1);
    return readlane_d((i & 1) ? lo : hi, jj);
  }
};
SGD_DEFINE_EXP(sgd_exp_lanes, const LaneExpTab&)

// one counter of the ring, the same value in every lane and known to be so
__device__ __forceinline__ unsigned long long ctrl_load(const volatile SGD_LDS(unsigned long long)* c) {
  const unsigned long long v = *c;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void wave_mem_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

template <bool kLds>
__global__ __launch_bounds__(2 * kWave) void saga_sparse_exact_k1x_kernel(SagaDev d, const LamParams* lamp,
                                                                          ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t p = d.p;
  const unsigned L = (unsigned)ctl.ls_cache;
  // (LDS pointers carry their address space in the type: a pointer the compiler cannot place becomes a FLAT access,
  // which waits for the vector-memory AND the LDS counter to drain -- i.e. for every store in flight)
  SGD_LDS(double)* sx = (SGD_LDS(double)*)smem;                       // [kK1Sum]
  SGD_LDS(double)* sexp = sx + kK1Sum;                                // [128]
  SGD_LDS(unsigned char)* owner = (SGD_LDS(unsigned char)*)(sexp + 128);
  SGD_LDS(double)* rval = (SGD_LDS(double)*)(owner + kOwnSlots);      // [kRing][kWave]
  SGD_LDS(double)* rhdr = rval + kRing * kWave;                       // [kRing][kSlotHdr]
  SGD_LDS(int)* ridx = (SGD_LDS(int)*)(rhdr + kRing * kSlotHdr);      // [kRing][kWave]
  volatile SGD_LDS(unsigned long long)* ctrl = (volatile SGD_LDS(unsigned long long)*)(ridx + kRing * kWave);  // produced, consumed, stop
  SGD_LDS(double)* sls = (SGD_LDS(double)*)(ridx + kRing * kWave) + 4;
  typename K1State<kLds>::D w;
  typename K1State<kLds>::D G;
  typename K1State<kLds>::U lag;
  if constexpr (kLds) {
    w = sls + L;
    G = w + p;
    lag = (typename K1State<kLds>::U)(G + p);
    for (int64_t i = tid; i < p; i += 2 * kWave) {
      w[i] = d.w[i];
      G[i] = d.G[i];
    }
  } else {
    w = d.w;
    G = d.G;
    lag = d.lag;
  }
  const unsigned nit = (unsigned)ctl.nit;
  const double* LS = ctl.LS;
  for (unsigned i = tid; i < L; i += 2 * kWave) sls[i] = LS[i];
  for (int i = tid; i < 128; i += 2 * kWave) sexp[i] = SGD_EXP_TABPTR[i];
  for (int i = tid; i < kOwnSlots / 4; i += 2 * kWave) ((SGD_LDS(unsigned)*)owner)[i] = 0u;
  for (int i = tid; i < kK1Sum; i += 2 * kWave) sx[i] = 0.0;
  if (tid < 4) ctrl[tid] = 0ull;
  for (int64_t j = tid; j < p; j += 2 * kWave) lag[j] = 0u;          // saga-sparse.h:225
  for (int64_t i = tid; i < p; i += 2 * kWave) d.w_prev[i] = w[i];    // :251
  __syncthreads();                                                    // the last barrier both wavefronts meet

  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :234
  const double bg = beta * gamma;                                    // penalties.h:49: beta * gamma * scaling / w_scale
  const double ls_one = 1u < L ? sls[1] : LS[1];                     // the SAGA step always lags by one
  const double bg_ls1 = bg * ls_one;
  const int64_t total = (int64_t)ctl.max_epochs * nit;               // draws of this launch, unless it converges first
  const int64_t t_last = ctl.stream_off + total - 1;

  if (wave == 1) {
    // ================================ producer ================================
    auto stream_at = [&](int64_t u) -> uint32_t {
      const int64_t x = ctl.stream_off + u;
      return d.stream[x < t_last ? x : t_last];
    };
    const bool gl = lane < 16;
    uint32_t sv = gl ? stream_at(lane) : 0u, sv_n = gl ? stream_at(16 + lane) : 0u;
    int64_t pa = 0, pe = 0;
    double yv = 0.0, mv = 0.0;
    if (gl) {
      pa = d.ptr[sv];
      pe = d.ptr[sv + 1];
      yv = d.y[sv];
      mv = d.M[sv];
    }
    double W = 1.0, q_prev = gamma / W;
    unsigned itp = 0;
    bool stop = false;
    int64_t freed = 0;                 // slots the consumer is known to have taken
#ifdef SGDNET_PHASE_TIMING
    unsigned long long prod_spins = 0, prod_waits = 0;
#endif
    for (int64_t ub = 0; ub <= total && !stop; ub += 16) {
      // the gathers of the next sixteen draws and the stream entries of the sixteen after them
      const uint32_t sv_nn = gl ? stream_at(ub + 32 + lane) : 0u;
      int64_t pa_n = 0, pe_n = 0;
      double yv_n = 0.0, mv_n = 0.0;
      if (gl) {
        pa_n = d.ptr[sv_n];
        pe_n = d.ptr[sv_n + 1];
        yv_n = d.y[sv_n];
        // (the consumer's stores of draws more than 64 back are long in L2; read past this CU's L1)
        mv_n = __hip_atomic_load(d.M + sv_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int i0 = 0; i0 < 16 && !stop; i0 += 4) {
        int idx4[4];
        double val4[4];
        int64_t a4[4], e4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a4[j] = readlane_ll(pa, i0 + j);
          e4[j] = readlane_ll(pe, i0 + j);
          idx4[j] = 0;
          val4[j] = 0.0;
          if (a4[j] + lane < e4[j]) {
            idx4[j] = d.idx[a4[j] + lane];
            val4[j] = d.val[a4[j] + lane];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t u = ub + i0 + j;
          if (u > total || stop) break;
          if (u - freed >= kRing) {                                  // ring full: the consumer frees a slot per draw
            unsigned spins = 0;
            long long spin_t0 = 0;
            for (;;) {
              freed = (int64_t)ctrl_load(ctrl + 1);
              if (u - freed < kRing) break;
              if (ctrl_load(ctrl + 2) != 0ull || spin_expired(spins, spin_t0)) {   // (the limit: an exit every wavefront reaches)
                stop = true;
                break;
              }
              __builtin_amdgcn_s_sleep(2);
            }
#ifdef SGDNET_PHASE_TIMING
            prod_spins += spins;
            ++prod_waits;
#endif
          }
          if (stop) break;
          const int slot = (int)(u & (kRing - 1));
          const uint32_t s_u = (uint32_t)__builtin_amdgcn_readlane((int)sv, i0 + j);
          const double y_u = readlane_d(yv, i0 + j), m_u = readlane_d(mv, i0 + j);
          const int64_t rlen = e4[j] - a4[j];
          const int len_u = (int)(rlen < (int64_t)(kWave + 1) ? rlen : (int64_t)(kWave + 1));
          // w_scale as the draw's catch-up sees it, what the draw leaves behind, and the quotients by the latter
          const double Wp = (W < kSmall ? 1.0 : W) * wscale_update;
          const double r = (lane == 0 ? gamma : bg_ls1) / Wp;
          const double q_t = readlane_d(r, 0), tau1 = readlane_d(r, 1);
          ridx[slot * kWave + lane] = idx4[j];
          rval[slot * kWave + lane] = val4[j];
          double hv = __longlong_as_double(((long long)len_u << 32) | (long long)s_u);
          hv = lane == 1 ? __longlong_as_double(a4[j]) : hv;
          hv = lane == 2 ? y_u : hv;
          hv = lane == 3 ? m_u : hv;
          hv = lane == 4 ? W : hv;
          hv = lane == 5 ? Wp : hv;
          hv = lane == 6 ? q_prev : hv;
          hv = lane == 7 ? q_t : hv;
          hv = lane == 8 ? tau1 : hv;
          hv = lane == 9 ? __longlong_as_double(e4[j]) : hv;
          if (lane < kSlotHdr) rhdr[slot * kSlotHdr + lane] = hv;
          lanes_publish();
          if (lane == 0) ctrl[0] = (unsigned long long)(u + 1);
          q_prev = q_t;
          W = Wp;
          if (++itp == nit) {                                      // Reset(n_samples) :340-348 leaves w_scale = 1
            itp = 0;
            W = 1.0;
            q_prev = gamma / W;
          }
        }
      }
      sv = sv_n;
      sv_n = sv_nn;
      pa = pa_n;
      pe = pe_n;
      yv = yv_n;
      mv = mv_n;
    }
#ifdef SGDNET_PHASE_TIMING
    if (d.dbg && lane == 0) {
      d.dbg[8] += prod_spins;
      d.dbg[9] += prod_waits;
    }
#endif
    return;
  }

  // ================================ consumer ================================
  auto ls_at = [&](unsigned m) -> double {   // (never a select between an LDS and a global address: that is a FLAT load)
    double v = sls[m < L ? m : 0u];
    if (m >= L) v = LS[m];
    return v;
  };
  const int penalty = lamp->penalty;
  const bool group = penalty == SGDNET_GROUPLASSO;                    // one response: a group of one (generic functor)
  const bool l1 = penalty == SGDNET_ELASTICNET;
  // the one-operation soft threshold needs a threshold that is never negative: true while w_scale stays positive
  const bool plain_soft = !(wscale_update > 0.0 && bg >= 0.0);
  const int family = d.family;
  const bool fit_intercept = d.fit_intercept != 0;
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
  const double g_scale = 1.0 / n_d;                                  // AddWeighted(g_sum, ..., 1/n)
  double b = d.b[0], gb = d.gb[0];                                   // intercept, g_sum_intercept: every lane the same
  uint32_t hs = 0xffffffffu;                                         // lane i: sample and gradient of draw u with u % 64 == i
  double hg = 0.0;

  // a slot, as registers
  int64_t avail = 0;                  // slots known to be filled
#ifdef SGDNET_PHASE_TIMING
  unsigned long long cons_spins = 0, cons_waits = 0;
#endif
  bool stalled = false;               // the producer stopped feeding the ring: give up with an error instead of spinning
  int idx_n;
  double val_n, h_sl, h_q0, h_q1, y_n, m_n, W_n, Wp_n, qp_n, qt_n, tau1_n;
  auto read_slot = [&](int64_t u) {
    if (u >= avail) {
      unsigned spins = 0;
      long long spin_t0 = 0;
      for (;;) {
        avail = (int64_t)ctrl_load(ctrl);
        if (u < avail) break;
        if (spin_expired(spins, spin_t0)) {
          stalled = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#ifdef SGDNET_PHASE_TIMING
      cons_spins += spins;
      ++cons_waits;
#endif
    }
    const int slot = (int)(u & (kRing - 1));
    lanes_publish();
    idx_n = ridx[slot * kWave + lane];
    val_n = rval[slot * kWave + lane];
    const SGD_LDS(double)* h = rhdr + slot * kSlotHdr;
    h_sl = h[0];
    h_q0 = h[1];
    y_n = h[2];
    m_n = h[3];
    W_n = h[4];
    Wp_n = h[5];
    qp_n = h[6];
    qt_n = h[7];
    tau1_n = h[8];
    h_q1 = h[9];
    lanes_publish();
    if (lane == 0) ctrl[1] = (unsigned long long)(u + 1);            // (the LDS runs these reads before this store)
  };
  double w0 = 0.0, G0 = 0.0, lsc0 = 0.0, tauc = 0.0;
  unsigned lag0 = 0u;
  // state of the draw whose slot is in the *_n registers, from memory
  auto fetch_state = [&](unsigned it_of_draw) {
    const int len_n = (int)(__double_as_longlong(h_sl) >> 32);
    w0 = 0.0;
    G0 = 0.0;
    lag0 = it_of_draw;
    if (lane < len_n && len_n <= kWave) {
      w0 = w[idx_n];
      G0 = G[idx_n];
      lag0 = lag[idx_n];
    }
    lsc0 = ls_at(it_of_draw - lag0);
    tauc = bg * lsc0 / W_n;
  };
  read_slot(0);
  fetch_state(0u);

  unsigned it_outer = 0;
  int converged = 0;
  int64_t u = 0;
#ifdef SGDNET_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define K1_STAMP(i) do { const unsigned long long now_ = clock64(); ph[i] += now_ - last_; last_ = now_; } while (0)
#else
#define K1_STAMP(i) ((void)0)
#endif
  do {
    double W_end = 1.0;
    for (unsigned it = 0; it < nit; ++it, ++u) {
#ifdef SGDNET_PHASE_TIMING
      unsigned long long last_ = clock64();
#endif
      // ---- the draw in hand -----------------------------------------------------------------------
      const long long sl = __double_as_longlong(h_sl);
      const uint32_t s = (uint32_t)(sl & 0xffffffffll);             // :261
      const int len = (int)(sl >> 32);
      const int64_t q0 = __double_as_longlong(h_q0), q1 = __double_as_longlong(h_q1);
      const int idx_c = idx_n;
      const double val_c = val_n, y_c = y_n;
      double m_c = m_n;
      const double W = W_n, Wp = Wp_n, q_prev = qp_n, q_t = qt_n, tau1_t = tau1_n;
      const bool shortrow = len <= kWave;
      const bool mine = lane < len && shortrow;
      // ---- the next draw's slot is asked for now and looked at after this draw's catch-up ----
      read_slot(u + 1);
      if (stalled) break;
      const int r0 = (int)(u & (kWave - 1));

      K1_STAMP(0);
      double wj = w0, Gj = G0;
      double acc = 0.0;
      int len_nx = 0;
      bool mine_n = false, collide = false;
      double w_n = 0.0, G_n = 0.0;
      unsigned lag_n = it + 1u, cand = 0u;
      if (shortrow) {
        // LaggedUpdate(it_inner): catch-up of the sample's features  :263-272
        const unsigned lagged = it - lag0;
        if (mine && lagged != 0u) {
          if (group) {
            penalty_apply_q(penalty, 1, &wj, &Gj, W, lsc0, q_prev, gamma, beta);
          } else {
            const double f = q_prev * lsc0;
            const double v = wj - f * Gj;
            wj = l1 ? k1_soft(v, tauc, plain_soft) : v;
          }
        }
        // linear predictor, ascending feature order  :274 (products past the row are +0.0: adding them changes nothing)
        sx[lane] = mine ? val_c * wj : 0.0;
        lanes_publish();
        // (the first sixteen products are asked back at once, and before the requests below: read four at a time
        // inside the adding loop, every block paid an LDS round trip of its own -- 1500 cycles for 30 entries)
        double pr[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) pr[e] = sx[e];
        lanes_publish();
        // ---- the next draw: w / g_sum / lag, requested before this draw's stores (forwarded below) ----
        len_nx = (int)(__double_as_longlong(h_sl) >> 32);
        mine_n = lane < len_nx && len_nx <= kWave;
        if (mine_n) {
          w_n = w[idx_n];
          G_n = G[idx_n];
          lag_n = lag[idx_n];
        }
        // which lane of this draw, if any, holds the feature a lane of the next draw asked for
        {
          const int h_c = idx_c & (kOwnSlots - 1), h_n = idx_n & (kOwnSlots - 1);
          if (mine) owner[h_c] = (unsigned char)(lane + 1);
          lanes_publish();
          const unsigned chk = mine ? owner[h_c] : (unsigned)(lane + 1);
          cand = mine_n ? owner[h_n] : 0u;
          lanes_publish();
          if (mine) owner[h_c] = 0;
          collide = __ballot(chk != (unsigned)(lane + 1)) != 0ull;   // two features of this row share a slot: compare instead
        }
        for (int base = 0;;) {
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) {
            if (base + 4 * blk < len) {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc += pr[4 * blk + e];
            }
          }
          base += 16;
          if (base >= len) break;
#pragma unroll
          for (int e = 0; e < 16; ++e) pr[e] = sx[base + e];
        }
      } else {
        // ---- the next draw: w / g_sum / lag, requested before this draw's stores (forwarded below) ----
        len_nx = (int)(__double_as_longlong(h_sl) >> 32);
        mine_n = lane < len_nx && len_nx <= kWave;
        if (mine_n) {
          w_n = w[idx_n];
          G_n = G[idx_n];
          lag_n = lag[idx_n];
        }
        wave_mem_sync();                     // the stores of the draws before it
        for (int64_t q = q0 + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const unsigned lagged = it - lag[j];
          if (lagged != 0u) {
            penalty_apply_q(penalty, 1, w + j, G + j, W, ls_at(lagged), q_prev, gamma, beta);
            lag[j] = it;
          }
        }
        wave_mem_sync();
        for (int64_t base = q0; base < q1; base += kWave) {
          const int64_t q = base + lane;
          const double wx = q < q1 ? d.val[q] * w[d.idx[q]] : 0.0;
          const int cnt = (q1 - base) < (int64_t)kWave ? (int)(q1 - base) : kWave;
          for (int e = 0; e < cnt; ++e) acc += readlane_d(wx, e);
        }
      }
      K1_STAMP(1);
      // gradient memory: a draw of the same sample among the last 64 supersedes the producer's copy
      {
        const unsigned long long mask = __ballot(hs == s);
        if (mask != 0ull) {
          const unsigned long long rot = r0 ? ((mask >> r0) | (mask << (kWave - r0))) : mask;   // bit k: lane (k + r0) % 64, age 64 - k
          const int kk = 63 - __builtin_clzll(rot);
          m_c = readlane_d(hg, (kk + r0) & (kWave - 1));
        }
      }
      const double lp = acc * W + b;

      // gradient, gradient memory  :279-282
      double g;
      if (family == SGDNET_BINOMIAL)
        g = 1.0 - y_c - 1.0 / (1.0 + sgd_exp_lds(lp, sexp));
      else
        g = lp - y_c;
      const double gc = g - m_c;
      if (lane == 0) d.M[s] = g;
      if (lane == r0) {
        hs = s;
        hg = g;
      }

      K1_STAMP(2);
      // rescale + unlag whenever wscale becomes too small  :285-295
      bool refetch = !shortrow;
      if (W < kSmall) {
        if (mine) {                          // the caught-up features go to memory first
          w[idx_c] = wj;
          lag[idx_c] = it;
        }
        wave_mem_sync();
        for (int64_t j = lane; j < p; j += kWave) {
          const unsigned lagged = it - lag[j];
          if (lagged != 0u) penalty_apply(penalty, 1, w + j, G + j, W, ls_at(lagged), gamma, beta);
          w[j] *= W;
          lag[j] = it;
        }
        wave_mem_sync();
        if (mine) wj = w[idx_c];
        refetch = true;
      }
      W_end = Wp;                                                    // :297 (after the reset, if any)

      if (fit_intercept) {                                           // :300-304
        const double gck = div_by_n_exact(gc, n_d, rn_d);
        gb = gb + gck;
        b -= gamma * (gb * 0.01 + gck);
      }

      // the next draw's lag scaling and catch-up threshold, from the lag as loaded: a feature this draw forwards
      // below lags by zero and needs neither (a division here is off the path from this draw's stores to the next)
      double lsc_e = 0.0, tauc_e = 0.0;
      if (!refetch && it + 1u < nit) {
        lsc_e = ls_at((it + 1u) - lag_n);
        tauc_e = bg * lsc_e / W_n;
      }

      K1_STAMP(3);
      if (shortrow) {
        if (mine) {
          wj += val_c * gc * (-q_t);                                 // AddWeighted(w, ..., -gamma/wscale)  :306-313
          // LaggedUpdate(it_inner + 1): the SAGA step (the feature was caught up to `it`, so it lags by one) :316-325
          if (group) {
            penalty_apply_q(penalty, 1, &wj, &Gj, Wp, ls_one, q_t, gamma, beta);
          } else {
            const double f = q_t * ls_one;
            const double v = wj - f * Gj;
            wj = l1 ? k1_soft(v, tau1_t, plain_soft) : v;
          }
          Gj += val_c * gc * g_scale;                                // AddWeighted(g_sum, ..., 1/n)  :328-335
          w[idx_c] = wj;
          G[idx_c] = Gj;
          lag[idx_c] = it + 1u;
        }
      } else {
        const double scaling = -q_t;
        for (int64_t q = q0 + lane; q < q1; q += kWave) w[d.idx[q]] += d.val[q] * gc * scaling;
        for (int64_t q = q0 + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const unsigned lagged = (it + 1u) - lag[j];
          if (lagged != 0u) {
            penalty_apply_q(penalty, 1, w + j, G + j, Wp, ls_at(lagged), q_t, gamma, beta);
            lag[j] = it + 1u;
          }
          G[j] += d.val[q] * gc * g_scale;
        }
      }

      K1_STAMP(4);
      // ---- hand the next draw its state ------------------------------------------------------------
      if (refetch || it + 1u == nit) {
        // memory was rewritten behind the request (or is about to be, by Reset): ask again once the stores are in
        if (it + 1u < nit) {
          wave_mem_sync();
          fetch_state(it + 1u);
        }
      } else {
        bool hit = false;
        int src = 0;
        if (!collide) {
          src = (int)cand - 1;
          hit = cand != 0u;
          if (__ballot(hit) != 0ull) {
            const int je = __shfl(idx_c, src & (kWave - 1), kWave);     // every lane takes part: the source lanes must be live
            hit = hit && je == idx_n;
          }
        } else {
          for (int e = 0; e < len; ++e) {
            const int je = __builtin_amdgcn_readlane(idx_c, e);
            if (mine_n && idx_n == je) {
              src = e;
              hit = true;
            }
          }
        }
        if (__ballot(hit) != 0ull) {
          const double wf = __shfl(wj, src & (kWave - 1), kWave);
          const double Gf = __shfl(Gj, src & (kWave - 1), kWave);
          if (hit) {
            w_n = wf;
            G_n = Gf;
            lag_n = it + 1u;
          }
        }
        w0 = w_n;
        G0 = G_n;
        lag0 = lag_n;
        lsc0 = lsc_e;
        tauc = tauc_e;
      }
      K1_STAMP(5);
    }

    if (stalled) break;
    // Reset(n_samples): unlag and rescale  :340-348
    wave_mem_sync();
    for (int64_t j = lane; j < p; j += kWave) {
      const unsigned lagged = nit - lag[j];
      if (lagged != 0u) penalty_apply(penalty, 1, w + j, G + j, W_end, ls_at(lagged), gamma, beta);
      w[j] *= W_end;
      lag[j] = 0u;
    }
    wave_mem_sync();

    converged = convergence_check(w, d.w_prev, p, ctl.tol, lane);    // :367
    ++it_outer;
    wave_mem_sync();
    fetch_state(0u);
  } while (!converged && it_outer < ctl.max_epochs);                 // :371
  if (lane == 0) ctrl[2] = 1ull;                                      // the producer may be waiting for a free slot
#ifdef SGDNET_PHASE_TIMING
  if (d.dbg && lane == 0) {
    d.dbg[10] += cons_spins;
    d.dbg[11] += cons_waits;
    for (int i = 0; i < 6; ++i) d.dbg[i] += ph[i];
  }
#endif
#undef K1_STAMP

  if constexpr (kLds) {
    for (int64_t i = lane; i < p; i += kWave) {
      d.w[i] = w[i];
      d.G[i] = G[i];
    }
  }
  if (lane == 0) {
    d.b[0] = b;
    d.gb[0] = gb;
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = stalled ? -2 : converged;
  }
}

// --------------------------------------------------------------------------
// Several consumers (round 3, last step): draws whose supports are disjoint only meet in the intercept, so kCons
// consumer wavefronts take the draws round robin (draw u -> wavefront u % kCons) and everything but the intercept
// chain of different draws overlaps.  What keeps the result the reference's, bit for bit:
//   * registration, in draw order: a draw stamps its features in an LDS table (feature, hashed -> index of the last
//     draw that holds it) and reads the stamps it replaces; a stamp inside the window of draws still in flight means
//     "wait until every draw up to that one has completed" (a collision of the hash only makes a draw wait for more);
//   * completion: a wavefront publishes a draw as complete once its stores of w / g_sum / lag are acknowledged
//     (s_waitcnt vmcnt(0), taken right after the NEXT draw's registration so that it costs nothing); readers load
//     the state past the L1;
//   * the chain, in draw order: wait for b(u-1), form the linear predictor, the gradient and b(u), publish; the last
//     64 (sample, gradient) pairs travel with it through the LDS for draws that repeat a sample;
//   * the epoch end (Reset, ConvergenceCheck) between barriers of the consumers.
// w / g_sum / lag stay in memory (L2): with them in the LDS the window p <= 4800 is also the one where neighbouring
// draws share features all the time.  A row longer than a wavefront works on memory and alone: it waits for every draw
// before it and every draw behind it waits for it (a stamp of its own at registration); so does the draw that finds
// w_scale below SMALL and rescales w (the producers know which: the sequence of w_scale does not depend on the data).
// Every wait loop gives up after kSpinLimitTicks (spin_expired) or when another wavefront has raised the abort flag.
// --------------------------------------------------------------------------
constexpr int kCons = 6;                // consumer wavefronts (8 and 10 measured slower: the draws meet in the chain)
constexpr int kProd = 2;               // producer wavefronts: draw i of a batch of sixteen belongs to producer i % kProd
constexpr int kDepSlots = 16384;
constexpr int kK1mCtrl = 8;         // produced, registered, chain_done, stop/abort, barrier count, converged, epochs, spare
constexpr int kHist = 2 * kWave;        // (sample, gradient) pairs of the latest draws: the producers read the gradient memory up to
                                        // kRing + 32 + kCons draws ahead of the draw in hand, and a store of the last few draws before
                                        // THAT read may not have landed: the history reaches past both
constexpr size_t kK1mFixedLds = sizeof(double) * ((size_t)kCons * kK1Sum + 128 + kRing * (kWave + kSlotHdr) + 4 + kHist) +
                                sizeof(unsigned long long) * (kK1mCtrl + 2 * kRing + kWave) +
                                sizeof(int) * (kRing * kWave) + sizeof(unsigned) * (kHist + kDepSlots);

__global__ __launch_bounds__((kCons + kProd) * kWave) void saga_sparse_exact_k1m_kernel(SagaDev d, const LamParams* lamp,
                                                                                    ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int T = (kCons + kProd) * kWave;
  const int64_t p = d.p;
  const unsigned L = (unsigned)ctl.ls_cache;
  SGD_LDS(double)* sx_all = (SGD_LDS(double)*)smem;                           // [kCons][kK1Sum]
  SGD_LDS(double)* sexp = sx_all + kCons * kK1Sum;                            // [128]
  SGD_LDS(double)* rval = sexp + 128;                                         // [kRing][kWave]
  SGD_LDS(double)* rhdr = rval + kRing * kWave;                               // [kRing][kSlotHdr]
  SGD_LDS(double)* chainv = rhdr + kRing * kSlotHdr;                          // b, g_sum_intercept, w_scale at the epoch end, spare
  SGD_LDS(double)* hist_g = chainv + 4;                                       // [kHist]
  volatile SGD_LDS(unsigned long long)* ctrl = (volatile SGD_LDS(unsigned long long)*)(hist_g + kHist);   // [kK1mCtrl]
  volatile SGD_LDS(unsigned long long)* readslot = ctrl + kK1mCtrl;           // [kRing]: slot u % kRing was read for draw u
  volatile SGD_LDS(unsigned long long)* filled = readslot + kRing;            // [kRing]: slot u % kRing holds draw u
  volatile SGD_LDS(unsigned long long)* done_slot = filled + kRing;           // [kWave]: draw u complete -> [u % 64] = u + 1
  SGD_LDS(int)* ridx = (SGD_LDS(int)*)(done_slot + kWave);                    // [kRing][kWave]
  SGD_LDS(unsigned)* hist_s = (SGD_LDS(unsigned)*)(ridx + kRing * kWave);     // [kHist]
  SGD_LDS(unsigned)* lastw = hist_s + kHist;                                  // [kDepSlots]
  SGD_LDS(double)* sls = (SGD_LDS(double)*)(lastw + kDepSlots);               // [L]
  double* w = d.w;
  double* G = d.G;
  unsigned* lag = d.lag;
  const unsigned nit = (unsigned)ctl.nit;
  const double* LS = ctl.LS;
  for (unsigned i = tid; i < L; i += T) sls[i] = LS[i];
  for (int i = tid; i < 128; i += T) sexp[i] = SGD_EXP_TABPTR[i];
  for (int i = tid; i < kDepSlots; i += T) lastw[i] = 0u;
  for (int i = tid; i < kCons * kK1Sum; i += T) sx_all[i] = 0.0;
  if (tid < kK1mCtrl) ctrl[tid] = 0ull;
  if (tid < kRing) {
    readslot[tid] = 0ull;
    filled[tid] = 0ull;
  }
  if (tid < kWave) done_slot[tid] = 0ull;
  if (tid < kHist) {
    hist_s[tid] = 0xffffffffu;
    hist_g[tid] = 0.0;
  }
  if (tid == 0) {
    chainv[0] = d.b[0];
    chainv[1] = d.gb[0];
    chainv[2] = 1.0;
  }
  for (int64_t j = tid; j < p; j += T) lag[j] = 0u;                   // saga-sparse.h:225
  for (int64_t i = tid; i < p; i += T) d.w_prev[i] = w[i];            // :251
  __syncthreads();                                                    // the last barrier all wavefronts meet

  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :234
  const double bg = beta * gamma;                                    // penalties.h:49: beta * gamma * scaling / w_scale
  const double ls_one = 1u < L ? sls[1] : LS[1];                     // the SAGA step always lags by one
  const double bg_ls1 = bg * ls_one;
  const int64_t total = (int64_t)ctl.max_epochs * nit;               // draws of this launch, unless it converges first
  const int64_t t_last = ctl.stream_off + total - 1;
  // waits: true when the condition came true, false when the launch is being abandoned
  auto aborted = [&]() -> bool { return ctrl_load(ctrl + 3) != 0ull; };
  unsigned long long spin_count[5] = {0, 0, 0, 0, 0};   // slot, registration, dependency, chain, barrier (phase-timing builds report them)
  (void)spin_count;
  int spin_kind = 0;
  auto wait_ge = [&](volatile SGD_LDS(unsigned long long)* c, unsigned long long target, bool eager = false) -> bool {
    unsigned spins = 0;
    long long spin_t0 = 0;
    for (;;) {
      const unsigned long long now = ctrl_load(c);
      if (now >= target) break;
      const bool next_in_line = eager && now + 1ull >= target;       // the wavefront whose turn comes next does not sleep
#ifdef SGDNET_PHASE_TIMING
      ++spin_count[spin_kind];
#endif
      if ((spins & 15u) == 15u && aborted()) return false;
      if (spin_expired(spins, spin_t0)) {
        if (lane == 0) ctrl[3] = 2ull;
        return false;
      }
      if (!next_in_line) __builtin_amdgcn_s_sleep(1);
    }
    return true;
  };

  if (wave >= kCons) {
    // ================================ producers ================================
    // kProd of them: draw i of every batch of sixteen belongs to producer i % kProd.  A producer works four batches
    // deep: the stream entries of batch b + 3, the row pointers of b + 2, the rows, responses and gradient memories
    // of b + 1 (one lane per draw for the gathers, a load per row) are requested before the slots of batch b are
    // filled, as the consumers free them.  Both run the whole w_scale sequence (sixteen multiplications a batch).
    const int q = wave - kCons;
    constexpr int kOwn = 16 / kProd;                                  // my draws per batch
    auto stream_at = [&](int64_t u) -> uint32_t {
      const int64_t x = ctl.stream_off + u;
      return d.stream[x < t_last ? x : t_last];
    };
    const bool gl = lane < kOwn;
    const int mine_i = (lane & (kOwn - 1)) * kProd + q;               // my lane's draw inside a batch
    uint32_t sv0 = gl ? stream_at(mine_i) : 0u, sv1 = gl ? stream_at(16 + mine_i) : 0u,
             sv2 = gl ? stream_at(32 + mine_i) : 0u;
    int64_t pa0 = 0, pe0 = 0, pa1 = 0, pe1 = 0;
    double yv0 = 0.0, mv0 = 0.0;
    if (gl) {
      pa0 = d.ptr[sv0];
      pe0 = d.ptr[sv0 + 1];
      pa1 = d.ptr[sv1];
      pe1 = d.ptr[sv1 + 1];
      yv0 = d.y[sv0];
      mv0 = d.M[sv0];
    }
    int idx0[kOwn];
    double val0[kOwn];
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
      const int64_t a = readlane_ll(pa0, j), e = readlane_ll(pe0, j);
      idx0[j] = 0;
      val0[j] = 0.0;
      if (a + lane < e) {
        idx0[j] = d.idx[a + lane];
        val0[j] = d.val[a + lane];
      }
    }
    double W = 1.0;                    // w_scale before the first draw of the batch in hand
    unsigned itp = 0;
    bool stop = false;
    for (int64_t ub = 0; ub < total && !stop; ub += 16) {
      // ---- requests of the batches behind this one ----
      const uint32_t sv3 = gl ? stream_at(ub + 48 + mine_i) : 0u;
      int64_t pa2 = 0, pe2 = 0;
      double yv1 = 0.0, mv1 = 0.0;
      if (gl) {
        pa2 = d.ptr[sv2];
        pe2 = d.ptr[sv2 + 1];
        yv1 = d.y[sv1];
        // (the consumers' stores of draws more than 64 back are long in L2; read past this CU's L1)
        mv1 = __hip_atomic_load(d.M + sv1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int idx1[kOwn];
      double val1[kOwn];
#pragma unroll
      for (int j = 0; j < kOwn; ++j) {
        const int64_t a = readlane_ll(pa1, j), e = readlane_ll(pe1, j);
        idx1[j] = 0;
        val1[j] = 0.0;
        if (a + lane < e) {
          idx1[j] = d.idx[a + lane];
          val1[j] = d.val[a + lane];
        }
      }
      // ---- the data-independent numbers of my draws at once: lane l works for my draw l % kOwn; lanes [0, kOwn) divide
      // gamma by w_scale before the draw, [kOwn, 2 kOwn) gamma by w_scale after it, [2 kOwn, 3 kOwn) the SAGA
      // step's threshold by the latter ----
      double Wb = W, Wa = W;
      double Wnext = W;
      unsigned itnext = itp;
#pragma unroll
      for (int t16 = 0; t16 < 16; ++t16) {
        const double before = Wnext;
        const double after = (before < kSmall ? 1.0 : before) * wscale_update;     // :285-297
        if (t16 == mine_i) {
          Wb = before;
          Wa = after;
        }
        Wnext = after;
        if (++itnext == nit) {                                     // Reset(n_samples) :340-348 leaves w_scale = 1
          itnext = 0;
          Wnext = 1.0;
        }
      }
      const int kind = lane / kOwn;
      const double quot = (kind == 2 ? bg_ls1 : gamma) / (kind == 0 ? Wb : Wa);
      // ---- the slots of this batch ----
#pragma unroll
      for (int j = 0; j < kOwn; ++j) {
        const int64_t u = ub + (int64_t)j * kProd + q;
        const int slot = (int)(u & (kRing - 1));
        if (u < total && !stop && u >= kRing && !wait_ge(readslot + slot, (unsigned long long)(u - kRing + 1))) stop = true;
        if (u < total && !stop) {
          const int64_t a_u = readlane_ll(pa0, j), e_u = readlane_ll(pe0, j);
          const uint32_t s_u = (uint32_t)__builtin_amdgcn_readlane((int)sv0, j);
          const double y_u = readlane_d(yv0, j), m_u = readlane_d(mv0, j);
          const int64_t rlen = e_u - a_u;
          const int len_u = (int)(rlen < (int64_t)(kWave + 1) ? rlen : (int64_t)(kWave + 1));
          const double W_u = readlane_d(Wb, j), Wp_u = readlane_d(Wa, j);
          const double qp_u = readlane_d(quot, j), qt_u = readlane_d(quot, kOwn + j), tau1_u = readlane_d(quot, 2 * kOwn + j);
          ridx[slot * kWave + lane] = idx0[j];
          rval[slot * kWave + lane] = val0[j];
          double hv = __longlong_as_double(((long long)len_u << 32) | (long long)s_u);
          hv = lane == 1 ? __longlong_as_double(a_u) : hv;
          hv = lane == 2 ? y_u : hv;
          hv = lane == 3 ? m_u : hv;
          hv = lane == 4 ? W_u : hv;
          hv = lane == 5 ? Wp_u : hv;
          hv = lane == 6 ? qp_u : hv;
          hv = lane == 7 ? qt_u : hv;
          hv = lane == 8 ? tau1_u : hv;
          hv = lane == 9 ? __longlong_as_double(e_u) : hv;
          if (lane < kSlotHdr) rhdr[slot * kSlotHdr + lane] = hv;
          lanes_publish();
          if (lane == 0) filled[slot] = (unsigned long long)(u + 1);
        }
      }
      W = Wnext;
      itp = itnext;
      sv0 = sv1;
      sv1 = sv2;
      sv2 = sv3;
      pa0 = pa1;
      pe0 = pe1;
      pa1 = pa2;
      pe1 = pe2;
      yv0 = yv1;
      mv0 = mv1;
#pragma unroll
      for (int j = 0; j < kOwn; ++j) {
        idx0[j] = idx1[j];
        val0[j] = val1[j];
      }
    }
    return;
  }

  // ================================ consumers ================================
  const int c = wave;
  SGD_LDS(double)* sx = sx_all + c * kK1Sum;
  auto ls_at = [&](unsigned m) -> double {
    double v = sls[m < L ? m : 0u];
    if (m >= L) v = LS[m];
    return v;
  };
  const int penalty = lamp->penalty;
  const bool group = penalty == SGDNET_GROUPLASSO;
  const bool l1 = penalty == SGDNET_ELASTICNET;
  const bool plain_soft = !(wscale_update > 0.0 && bg >= 0.0);
  const int family = d.family;
  const bool fit_intercept = d.fit_intercept != 0;
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
  const double g_scale = 1.0 / n_d;
  const LaneExpTab exp_tab{SGD_EXP_TABPTR[2 * lane], SGD_EXP_TABPTR[2 * lane + 1]};   // lane j: 2^(j/64) = hi + lo
  unsigned long long bar_target = 0ull;
  bool ok = true;
  // a barrier of the consumers (the producer does not take part)
  auto consumers_meet = [&]() {
    lanes_publish();
    bar_target += kCons;
    if (lane == 0) __hip_atomic_fetch_add((SGD_LDS(unsigned long long)*)(ctrl + 4), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    spin_kind = 4;
    if (ok) ok = wait_ge(ctrl + 4, bar_target);
    lanes_publish();
  };

  unsigned it_outer = 0;
  int converged = 0;
  int64_t base = 0;                   // draws before this epoch
  int64_t u_prev = -1;                // my last draw whose completion is not published yet
  do {
    for (unsigned it = (unsigned)c; it < nit && ok; it += kCons) {
      const int64_t u = base + it;
      // ---- the slot ----
      spin_kind = 0;
      const int slot = (int)(u & (kRing - 1));
      if (!(ok = wait_ge(filled + slot, (unsigned long long)(u + 1)))) break;
      lanes_publish();
      const int idx_c = ridx[slot * kWave + lane];
      const double val_c = rval[slot * kWave + lane];
      const SGD_LDS(double)* h = rhdr + slot * kSlotHdr;
      const long long sl = __double_as_longlong(h[0]);
      const double y_c = h[2];
      double m_c = h[3];
      const double W = h[4], Wp = h[5], q_prev = h[6], q_t = h[7], tau1_t = h[8];
      const double h1 = h[1], h9 = h[9];
      lanes_publish();
      if (lane == 0) readslot[slot] = (unsigned long long)(u + 1);
      const uint32_t s = (uint32_t)(sl & 0xffffffffll);             // :261
      const int len = (int)(sl >> 32);
      const bool longrow = len > kWave;                              // works on memory, alone: see below
      const bool rescale = W < kSmall;                               // the draw that rescales w (:285-295) is alone as well
      const bool mine = lane < len && !longrow;
      const int64_t q0 = __double_as_longlong(h1), q1 = __double_as_longlong(h9);
      // ---- registration, in draw order ----
      spin_kind = 1;
      if (!(ok = wait_ge(ctrl + 1, (unsigned long long)u))) break;
      const int hsl = idx_c & (kDepSlots - 1);
      unsigned prev = mine ? lastw[hsl] : 0u;
      // a row longer than the wavefront is a dependency of everything behind it and depends on everything before it
      const unsigned long_before = (unsigned)ctrl_load(ctrl + 6);
      prev = prev > long_before ? prev : long_before;
      if (longrow || rescale) prev = it;
      lanes_publish();
      if (mine) lastw[hsl] = it + 1u;
      if ((longrow || rescale) && lane == 0) ctrl[6] = (unsigned long long)(it + 1u);
      // ... and the sample: the latest of the 64 draws before this one that drew it too, if any, hands its gradient on
      const int r0 = (int)(u & (kHist - 1));
      const unsigned long long same_lo = __ballot(hist_s[lane] == s), same_hi = __ballot(hist_s[kWave + lane] == s);
      int repeat_lane = -1;                                            // entry of the LATEST of the kHist draws before this one
      if ((same_lo | same_hi) != 0ull) {
        // entries in order of age: r0 - 1 down to 0, then kHist - 1 down to r0 (all wave-uniform arithmetic)
        const int h0 = r0 & (kWave - 1);
        const unsigned long long below = h0 ? ((1ull << h0) - 1ull) : 0ull;
        const unsigned long long cur = r0 < kWave ? same_lo : same_hi, oth = r0 < kWave ? same_hi : same_lo;
        const int cur_base = r0 < kWave ? 0 : kWave, oth_base = r0 < kWave ? kWave : 0;
        if (cur & below) repeat_lane = cur_base + 63 - __builtin_clzll(cur & below);
        else if (oth) repeat_lane = oth_base + 63 - __builtin_clzll(oth);
        else repeat_lane = cur_base + 63 - __builtin_clzll(cur & ~below);
      }
      lanes_publish();
      if (lane == 0) {
        hist_s[r0] = s;
        ctrl[1] = (unsigned long long)(u + 1);
      }
      // ---- my previous draw is complete once its stores of w / g_sum / lag are acknowledged: everything but the
      // youngest vector-memory operation, which is its gradient-memory store (a line in HBM: ~2 us to acknowledge,
      // and nobody waits for it -- the chain hands the gradient on through the LDS) ----
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (u_prev >= 0 && lane == 0) done_slot[u_prev & (kWave - 1)] = (unsigned long long)(u_prev + 1);
      // ---- draws in flight that hold one of my features ----
      {
        unsigned need = prev;
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned o = (unsigned)__shfl_xor((int)need, off, kWave);
          need = o > need ? o : need;
        }
        need = (unsigned)__builtin_amdgcn_readfirstlane((int)need);
        if (need != 0u) {
          const unsigned v_it = need - 1u;                           // the latest earlier draw sharing a feature (or a hash slot)
          // (a wavefront flags draw u complete only after it has REGISTERED its next draw u + kCons, so a draw that
          //  has passed the registration wait may still find the stores of draws back to it - 2 kCons unacknowledged:
          //  the window covers them; the older flags are almost always set already)
          const unsigned lo = it + 1u > 2u * (unsigned)kCons ? it + 1u - 2u * (unsigned)kCons : 0u;
          spin_kind = 2;
          for (unsigned dd = lo; dd <= v_it && dd < it && ok; ++dd)
            ok = wait_ge(done_slot + ((base + dd) & (kWave - 1)), (unsigned long long)(base + dd + 1));
          if (!ok) break;
        }
      }
      double acc = 0.0;
      if (longrow) {
        // every earlier draw is complete and acknowledged: catch-up and the ordered sum on memory (:263-274), past the L1
        for (int64_t qq = q0 + lane; qq < q1; qq += kWave) {
          const int64_t jf = d.idx[qq];
          const unsigned lagged_l = it - __hip_atomic_load(lag + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lagged_l != 0u) {
            double wv = __hip_atomic_load(w + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double Gv = __hip_atomic_load(G + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            penalty_apply_q(penalty, 1, &wv, &Gv, W, ls_at(lagged_l), q_prev, gamma, beta);
            w[jf] = wv;
            lag[jf] = it;
          }
        }
        wave_mem_sync();
        for (int64_t base_q = q0; base_q < q1; base_q += kWave) {
          const int64_t qq = base_q + lane;
          const double wx = qq < q1 ? d.val[qq] * __hip_atomic_load(w + d.idx[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
          const int cnt = (q1 - base_q) < (int64_t)kWave ? (int)(q1 - base_q) : kWave;
          for (int e = 0; e < cnt; ++e) acc += readlane_d(wx, e);
        }
      }
      // ---- w / g_sum / lag of my features (past the L1: other wavefronts of this CU wrote them) ----
      double wj = 0.0, Gj = 0.0;
      unsigned lag0 = it;
      if (mine) {
        wj = __hip_atomic_load(w + idx_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Gj = __hip_atomic_load(G + idx_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lag0 = __hip_atomic_load(lag + idx_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // LaggedUpdate(it_inner): catch-up of the sample's features  :263-272
      const unsigned lagged = it - lag0;
      if (mine && lagged != 0u) {
        const double lsc0 = ls_at(lagged);
        if (group) {
          penalty_apply_q(penalty, 1, &wj, &Gj, W, lsc0, q_prev, gamma, beta);
        } else {
          const double f = q_prev * lsc0;
          const double v = wj - f * Gj;
          wj = l1 ? k1_soft(v, bg * lsc0 / W, plain_soft) : v;
        }
      }
      // linear predictor, ascending feature order  :274
      sx[lane] = mine ? val_c * wj : 0.0;
      lanes_publish();
      if (!longrow) {
        double pr[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) pr[e] = sx[e];
        for (int eb = 0;;) {
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) {
            if (eb + 4 * blk < len) {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc += pr[4 * blk + e];
            }
          }
          eb += 16;
          if (eb >= len) break;
#pragma unroll
          for (int e = 0; e < 16; ++e) pr[e] = sx[eb + e];
        }
      }
      // ---- the chain, in draw order ----
      const double aw = acc * W;
      spin_kind = 3;
      if (!(ok = wait_ge(ctrl + 2, (unsigned long long)u, true))) break;
      lanes_publish();
      double b = chainv[0], gb = chainv[1];
      if (repeat_lane >= 0) m_c = hist_g[repeat_lane];               // (that draw's chain step is behind us)
      const double lp = aw + b;
      double g;                                                      // :279-282
      if (family == SGDNET_BINOMIAL)
        g = 1.0 - y_c - 1.0 / (1.0 + sgd_exp_lanes(lp, exp_tab));
      else
        g = lp - y_c;
      const double gc = g - m_c;
      if (fit_intercept) {                                           // :300-304
        const double gck = div_by_n_exact(gc, n_d, rn_d);
        gb = gb + gck;
        b -= gamma * (gb * 0.01 + gck);
      }
      if (lane == 0) {
        chainv[0] = b;
        chainv[1] = gb;
        if (it + 1u == nit) chainv[2] = Wp;                          // w_scale as Reset finds it
        hist_g[r0] = g;
      }
      lanes_publish();
      if (lane == 0) ctrl[2] = (unsigned long long)(u + 1);
      // ---- the rest of the draw ----
      if (rescale) {
        // rescale + unlag  :285-295: every earlier draw is complete, every later one waits for this one
        if (mine) {                          // the caught-up features go to memory first
          w[idx_c] = wj;
          lag[idx_c] = it;
        }
        wave_mem_sync();
        for (int64_t jf = lane; jf < p; jf += kWave) {
          double wv = __hip_atomic_load(w + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double Gv = __hip_atomic_load(G + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned lagged_r = it - __hip_atomic_load(lag + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lagged_r != 0u) penalty_apply(penalty, 1, &wv, &Gv, W, ls_at(lagged_r), gamma, beta);
          w[jf] = wv * W;
          lag[jf] = it;
        }
        wave_mem_sync();
        if (mine) wj = __hip_atomic_load(w + idx_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (mine) {
        wj += val_c * gc * (-q_t);                                   // AddWeighted(w, ..., -gamma/wscale)  :306-313
        if (group) {                                                 // LaggedUpdate(it_inner + 1)  :316-325
          penalty_apply_q(penalty, 1, &wj, &Gj, Wp, ls_one, q_t, gamma, beta);
        } else {
          const double f = q_t * ls_one;
          const double v = wj - f * Gj;
          wj = l1 ? k1_soft(v, tau1_t, plain_soft) : v;
        }
        Gj += val_c * gc * g_scale;                                  // AddWeighted(g_sum, ..., 1/n)  :328-335
        w[idx_c] = wj;
        G[idx_c] = Gj;
        lag[idx_c] = it + 1u;
      }
      if (longrow) {
        const double scaling = -q_t;                                 // :306-335 on memory; the same lane holds an entry in both passes
        for (int64_t qq = q0 + lane; qq < q1; qq += kWave) {
          const int64_t jf = d.idx[qq];
          const double xv = d.val[qq];
          double wv = __hip_atomic_load(w + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          double Gv = __hip_atomic_load(G + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wv += xv * gc * scaling;
          const unsigned lagged_l = (it + 1u) - __hip_atomic_load(lag + jf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lagged_l != 0u) penalty_apply_q(penalty, 1, &wv, &Gv, Wp, ls_at(lagged_l), q_t, gamma, beta);
          Gv += xv * gc * g_scale;
          w[jf] = wv;
          G[jf] = Gv;
          lag[jf] = it + 1u;
        }
      }
      lanes_publish();                                               // (the compiler keeps the stores above above)
      if (lane == 0) d.M[s] = g;                                     // the youngest store of the draw (see vmcnt(1) above)
      u_prev = u;
    }
    // my last draw of the epoch
    wave_mem_sync();
    if (u_prev >= 0 && lane == 0) done_slot[u_prev & (kWave - 1)] = (unsigned long long)(u_prev + 1);
    u_prev = -1;
    consumers_meet();                                                // every draw of the epoch is in memory
    if (!ok) break;
    // Reset(n_samples): unlag and rescale  :340-348 (features dealt round robin to the consumers)
    {
      const double W_end = chainv[2];
      for (int64_t j = (int64_t)c * kWave + lane; j < p; j += kCons * kWave) {
        double wv = __hip_atomic_load(w + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double Gv = __hip_atomic_load(G + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned lagged = nit - __hip_atomic_load(lag + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lagged != 0u) penalty_apply(penalty, 1, &wv, &Gv, W_end, ls_at(lagged), gamma, beta);
        w[j] = wv * W_end;
        lag[j] = 0u;
      }
      for (int i = c * kWave + lane; i < kDepSlots; i += kCons * kWave) lastw[i] = 0u;
      if (c == 0 && lane == 0) ctrl[6] = 0ull;
    }
    wave_mem_sync();
    consumers_meet();
    if (!ok) break;
    ++it_outer;
    if (c == 0) {
      // ConvergenceCheck  :367 (w as the other consumers left it: past the L1)
      double max_change = 0.0, max_size = 0.0;
      bool finite = true;
      for (int64_t i = lane; i < p; i += kWave) {
        const double v = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        finite = finite && (fabs(v) <= 1.79769313486231570815e+308);
        max_change = fmax(max_change, fabs(v - d.w_prev[i]));
        max_size = fmax(max_size, fabs(v));
        d.w_prev[i] = v;
      }
      int conv = 0;
      if (__ballot(!finite) == 0ull) {
        max_change = wave_max(max_change);
        max_size = wave_max(max_size);
        const bool all_zero = (max_size == 0.0) && (max_change == 0.0);
        const bool no_change = (max_size != 0.0) && (max_change / max_size <= ctl.tol);
        conv = (all_zero || no_change) ? 1 : 0;
      }
      if (lane == 0) ctrl[5] = (unsigned long long)conv;
    }
    consumers_meet();
    if (!ok) break;
    converged = (int)ctrl_load(ctrl + 5);
    base += nit;
  } while (!converged && it_outer < ctl.max_epochs);                 // :371
  if (lane == 0 && ctrl_load(ctrl + 3) == 0ull) ctrl[3] = 1ull;       // the producer may be waiting for a free slot

#ifdef SGDNET_PHASE_TIMING
  if (d.dbg && lane == 0)
    for (int q = 0; q < 5; ++q) atomicAdd(d.dbg + 16 + q, spin_count[q]);
#endif
  if (c == 0 && lane == 0) {
    d.b[0] = chainv[0];
    d.gb[0] = chainv[1];
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = (!ok || ctrl_load(ctrl + 3) == 2ull) ? -2 : converged;
  }
}

// --------------------------------------------------------------------------
// Any number of classes up to 64, rows of any length (round 3): the general iteration of saga_sparse_exact_kernel run
// by kMc wavefronts at once, draw t on wavefront t % kMc, under the protocol of the kernel above -- registration of a
// draw's features AND of its sample in draw order (two stamp tables), completion flags, the intercepts handed on in
// draw order.  No producers: every wavefront keeps the general kernel's own pipeline of requests for its next draw,
// kMc draws ahead.  The state stays in memory; a wavefront invalidates the L1 once it may touch the state of a draw
// (buffer_inv sc1: the other wavefronts of this CU wrote it) and publishes a draw as complete when all its stores are
// acknowledged.  What is serial: the turn at registration (two LDS round trips) and, in draw order, intercept ->
// linear predictor -> gradient -> intercept; catch-up, the x.w sums, AddWeighted, the SAGA step and the gradient
// averages of different draws overlap.  w_scale does not depend on the data, so every wavefront steps its own copy
// kMc draws at a time; the draw that finds it below SMALL rescales w alone, as above.
// --------------------------------------------------------------------------
constexpr int kMc = 8;                  // wavefronts (12 and 16 measured: multinomial slower, mgaussian 10 % faster)
constexpr int kMcDep = 16384;        // feature stamps
constexpr int kMcSam = 8192;         // sample stamps
constexpr int kMcScratch = 3 * kWave;   // doubles per wavefront: slp[64], sgc[64], sval[64] (+ sidx[64] ints behind them)
constexpr size_t kMcFixedLds = sizeof(double) * ((size_t)kMc * kMcScratch + 2 * kWave + kLsCache + 2 + 128 + 258) + sizeof(int) * (kMc * kWave) +
                               sizeof(unsigned long long) * (8 + kWave) + sizeof(unsigned) * (kMcDep + kMcSam);

__global__ __launch_bounds__(kMc * kWave) void saga_sparse_exact_mc_kernel(SagaDev d, const LamParams* lamp, ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int T = kMc * kWave;
  const int K = d.K;
  const int64_t p = d.p;
  const int64_t KP = (int64_t)K * p;
  SGD_LDS(double)* scratch = (SGD_LDS(double)*)smem;                          // [kMc][kMcScratch]
  SGD_LDS(double)* sb = scratch + kMc * kMcScratch;                           // [kWave]: intercepts (the chain)
  SGD_LDS(double)* sgb = sb + kWave;                                          // [kWave]: g_sum_intercept
  SGD_LDS(double)* sls = sgb + kWave;                                         // [kLsCache]
  SGD_LDS(double)* wend = sls + kLsCache;                                     // [2]: w_scale as the epoch's last draw leaves it
  SGD_LDS(double)* etab = wend + 2;                                           // [128]: sgd_exp_tab
  SGD_LDS(double)* ltab = etab + 128;                                         // [258]: sgd_log_tab
  SGD_LDS(int)* sidx_all = (SGD_LDS(int)*)(ltab + 258);                       // [kMc][kWave]
  // registered, chain_done, abort, barrier count, converged, alone stamp, spare, spare
  volatile SGD_LDS(unsigned long long)* ctrl = (volatile SGD_LDS(unsigned long long)*)(sidx_all + kMc * kWave);
  volatile SGD_LDS(unsigned long long)* done_slot = ctrl + 8;                 // [kWave]
  SGD_LDS(unsigned)* lastw = (SGD_LDS(unsigned)*)(done_slot + kWave);         // [kMcDep]
  SGD_LDS(unsigned)* lasts = lastw + kMcDep;                                  // [kMcSam]
  SGD_LDS(double)* slp = scratch + c * kMcScratch;
  SGD_LDS(double)* sgc = slp + kWave;
  SGD_LDS(double)* sval = sgc + kWave;
  SGD_LDS(int)* sidx = sidx_all + c * kWave;
  double* w = d.w;
  double* G = d.G;
  unsigned* lag = d.lag;
  const unsigned nit = (unsigned)ctl.nit;
  const double* LS = ctl.LS;
  for (int i = tid; i < kLsCache && i <= (int64_t)nit; i += T) sls[i] = LS[i];
  for (int i = tid; i < 128; i += T) etab[i] = SGD_EXP_TABPTR[i];
  for (int i = tid; i < 258; i += T) ltab[i] = SGD_LOG_TABPTR[i];
  for (int i = tid; i < kMcDep; i += T) lastw[i] = 0u;
  for (int i = tid; i < kMcSam; i += T) lasts[i] = 0u;
  if (tid < 8) ctrl[tid] = 0ull;
  if (tid < kWave) {
    done_slot[tid] = 0ull;
    sb[tid] = tid < K ? d.b[tid] : 0.0;
    sgb[tid] = tid < K ? d.gb[tid] : 0.0;
  }
  for (int64_t j = tid; j < p; j += T) lag[j] = 0u;                   // saga-sparse.h:225
  for (int64_t i = tid; i < KP; i += T) d.w_prev[i] = w[i];           // :251
  __syncthreads();                                                    // the last barrier all wavefronts meet

  auto ls_at = [&](unsigned m) -> double {
    double v = sls[m < (unsigned)kLsCache ? m : 0u];
    if (m >= (unsigned)kLsCache) v = LS[m];
    return v;
  };
  auto aborted = [&]() -> bool { return ctrl_load(ctrl + 2) != 0ull; };
  auto wait_ge = [&](volatile SGD_LDS(unsigned long long)* cnt, unsigned long long target, bool eager) -> bool {
    unsigned spins = 0;
    long long spin_t0 = 0;
    for (;;) {
      const unsigned long long now = ctrl_load(cnt);
      if (now >= target) break;
      if ((spins & 15u) == 15u && aborted()) return false;
      if (spin_expired(spins, spin_t0)) {
        if (lane == 0) ctrl[2] = 2ull;
        return false;
      }
      if (!(eager && now + 1ull >= target)) __builtin_amdgcn_s_sleep(1);
    }
    return true;
  };
  unsigned long long bar_target = 0ull;
  bool ok = true;
  auto meet = [&]() {
    lanes_publish();
    bar_target += kMc;
    if (lane == 0) __hip_atomic_fetch_add((SGD_LDS(unsigned long long)*)(ctrl + 3), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ok) ok = wait_ge(ctrl + 3, bar_target, false);
    lanes_publish();
  };

  const int penalty = lamp->penalty;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :234
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
  const int64_t t0 = ctl.stream_off;
  const int64_t t_last = ctl.stream_off + (int64_t)ctl.max_epochs * nit - 1;
  auto clampt = [&](int64_t x) { return x < t_last ? x : t_last; };

  unsigned it_outer = 0;
  int converged = 0;
  int64_t base = 0;                   // draws before this epoch
  int64_t u_prev = -1;                // my last draw, complete but not published as such
  do {
    // my own pipeline of requests, kMc draws apart: stream entry two of my draws ahead, row pointers and row one ahead
    int64_t tm = t0 + base + c;                                       // stream position of my draw in hand
    uint32_t s1 = d.stream[clampt(tm)], s2 = d.stream[clampt(tm + kMc)];
    int64_t q0_1 = d.ptr[s1], q1_1 = d.ptr[s1 + 1];
    int idx_1 = 0;
    double val_1 = 0.0;
    if (q0_1 + lane < q1_1) {
      idx_1 = d.idx[q0_1 + lane];
      val_1 = d.val[q0_1 + lane];
    }
    double y_1 = lane < d.Ky ? d.y[(int64_t)s1 * d.Ky + lane] : 0.0;
    double wscale = 1.0;                                              // :227; before draw 0 of the epoch
    unsigned ws_it = 0;                                               // the draw `wscale` stands before
    for (unsigned it = (unsigned)c; it < nit && ok; it += kMc, tm += kMc) {
      const int64_t u = base + it;
      // ---- rotate the pipeline ----
      const uint32_t s = s1;                                          // :261
      const int64_t q0 = q0_1, q1 = q1_1;
      const int idx_c = idx_1;
      const double val_c = val_1;
      const double y_c = y_1;
      s1 = s2;
      s2 = d.stream[clampt(tm + 2 * kMc)];
      q0_1 = d.ptr[s1];
      q1_1 = d.ptr[s1 + 1];
      idx_1 = 0;
      val_1 = 0.0;
      if (q0_1 + lane < q1_1) {
        idx_1 = d.idx[q0_1 + lane];
        val_1 = d.val[q0_1 + lane];
      }
      y_1 = lane < d.Ky ? d.y[(int64_t)s1 * d.Ky + lane] : 0.0;
      // ---- w_scale before this draw: the data-independent sequence, stepped from where my last draw left it ----
      for (; ws_it < it; ++ws_it) {
        if (wscale < kSmall) wscale = 1.0;                            // (the draw that found it there rescaled w)
        wscale *= wscale_update;
      }
      const bool rescale = wscale < kSmall;
      const bool mine = q0 + lane < q1;                               // this lane holds entry q0 + lane
      // ---- registration, in draw order: features, sample, and whether the draw runs alone ----
      if (!(ok = wait_ge(ctrl + 0, (unsigned long long)u, false))) break;
      unsigned prev = 0u;
      for (int64_t q = q0 + lane; q < q1; q += kWave) {
        const int hsl = (q < q0 + kWave ? idx_c : d.idx[q]) & (kMcDep - 1);
        const unsigned o = lastw[hsl];
        prev = o > prev ? o : prev;
      }
      const unsigned prev_s = lasts[s & (kMcSam - 1)];
      prev = prev_s > prev ? prev_s : prev;
      const unsigned alone_before = (unsigned)ctrl_load(ctrl + 5);
      prev = alone_before > prev ? alone_before : prev;
      if (rescale) prev = it;
      lanes_publish();
      for (int64_t q = q0 + lane; q < q1; q += kWave) lastw[(q < q0 + kWave ? idx_c : d.idx[q]) & (kMcDep - 1)] = it + 1u;
      if (lane == 0) {
        lasts[s & (kMcSam - 1)] = it + 1u;
        if (rescale) ctrl[5] = (unsigned long long)(it + 1u);
      }
      lanes_publish();
      if (lane == 0) ctrl[0] = (unsigned long long)(u + 1);
      // ---- my previous draw is complete once its stores are acknowledged ----
      wave_mem_sync();
      if (u_prev >= 0 && lane == 0) done_slot[u_prev & (kWave - 1)] = (unsigned long long)(u_prev + 1);
      // ---- draws in flight that hold one of my features or my sample ----
      {
        unsigned need = prev;
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned o = (unsigned)__shfl_xor((int)need, off, kWave);
          need = o > need ? o : need;
        }
        need = (unsigned)__builtin_amdgcn_readfirstlane((int)need);
        if (need != 0u) {
          const unsigned v_it = need - 1u;
          const unsigned lo = it + 1u > 2u * (unsigned)kMc ? it + 1u - 2u * (unsigned)kMc : 0u;   // (two rounds back: see the k1m kernel)
          for (unsigned dd = lo; dd <= v_it && dd < it && ok; ++dd)
            ok = wait_ge(done_slot + ((base + dd) & (kWave - 1)), (unsigned long long)(base + dd + 1), false);
          if (!ok) break;
        }
      }
      asm volatile("buffer_inv sc1" ::: "memory");                    // the state other wavefronts of this CU stored
      const double m_c = lane < K ? d.M[lane + (int64_t)s * K] : 0.0;
      if (mine && lane < kWave) {
        sidx[lane] = idx_c;
        sval[lane] = val_c;
      }

      // LaggedUpdate(it_inner): catch-up of the sample's features  :263-272
      const double q_before = gamma / wscale;
      if (mine) {
        const int64_t j = idx_c;
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_before, gamma, beta);
          lag[j] = it;
        }
      }
      for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
        const int64_t j = d.idx[q];
        const unsigned lagged = it - lag[j];
        if (lagged != 0) {
          penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_before, gamma, beta);
          lag[j] = it;
        }
      }
      wave_mem_sync();
      asm volatile("buffer_inv sc1" ::: "memory");                    // (my own stores, read back by other lanes)

      // x.w per class, ascending feature order  :274 -- everything of the linear predictor but the intercept
      const int head = (q1 - q0) < kWave ? (int)(q1 - q0) : kWave;
      double acc = 0.0;
      if (lane < K) {
        // (the coefficients of eight entries are requested together, then added in order: one at a time, every
        // entry paid a round trip of its own)
        for (int e0 = 0; e0 < head; e0 += 8) {
          double wv[8], xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ee = e0 + e < head ? e0 + e : head - 1;
            xv[e] = sval[ee];
            wv[e] = w[lane + (int64_t)sidx[ee] * K];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e0 + e < head) acc += xv[e] * wv[e];
        }
        for (int64_t q = q0 + kWave; q < q1; ++q) acc += d.val[q] * w[lane + (int64_t)d.idx[q] * K];
      }

      // (what the draw leaves w_scale at, and gamma over it, do not depend on the data: formed ahead of the turn)
      const double wscale_after = (rescale ? 1.0 : wscale) * wscale_update;          // :285-297
      const double q_after = gamma / wscale_after;
      // ---- the chain, in draw order: intercept -> linear predictor -> gradient -> intercept  :274-304 ----
      if (!(ok = wait_ge(ctrl + 1, (unsigned long long)u, true))) break;
      lanes_publish();
      if (lane < K) slp[lane] = acc * wscale + sb[lane];
      lanes_publish();
      const double y_first = __shfl(y_c, 0, kWave);
      double g = 0.0;
      if (d.family == SGDNET_MULTINOMIAL) {
        g = softmax_gradient_lanes_lds(lane < K ? slp[lane] : 0.0, K, lane, y_first, etab, ltab);
      } else if (lane < K) {
        if (d.family == SGDNET_MGAUSSIAN)
          g = slp[lane] - y_c;
        else
          g = family_gradient_k(d.family, K, lane, slp, &y_first);
      }
      if (lane < K) sgc[lane] = g - m_c;
      if (rescale) {
        // rescale + unlag  :285-295: every earlier draw is complete, every later one waits for this one
        wave_mem_sync();
        asm volatile("buffer_inv sc1" ::: "memory");
        for (int64_t j = lane; j < p; j += kWave) {
          const unsigned lagged = it - lag[j];
          if (lagged != 0) penalty_apply(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), gamma, beta);
          for (int k = 0; k < K; ++k) w[k + j * K] *= wscale;
          lag[j] = it;
        }
        wave_mem_sync();
        asm volatile("buffer_inv sc1" ::: "memory");
      }
      wscale = wscale_after;                                         // :297
      ws_it = it + 1u;
      if (it + 1u == nit && lane == 0) wend[0] = wscale;             // Reset(n_samples) finds w_scale here
      lanes_publish();
      if (d.fit_intercept && lane < K) {                             // :300-304
        const double gck = div_by_n_exact(sgc[lane], n_d, rn_d);
        const double gbk = sgb[lane] + gck;
        sgb[lane] = gbk;
        sb[lane] -= gamma * (gbk * 0.01 + gck);
      }
      lanes_publish();
      if (lane == 0) ctrl[1] = (unsigned long long)(u + 1);
      if (lane < K) d.M[lane + (int64_t)s * K] = g;                  // (a draw of the same sample waits for this draw to complete)

      // AddWeighted(w, ..., -gamma/wscale)  :306-313
      {
        const double scaling = -q_after;
        if (mine) {
          const int64_t j = idx_c;
          for (int k = 0; k < K; ++k) w[k + j * K] += val_c * sgc[k] * scaling;
        }
        for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const double v = d.val[q];
          for (int k = 0; k < K; ++k) w[k + j * K] += v * sgc[k] * scaling;
        }
      }
      // LaggedUpdate(it_inner + 1): the SAGA step  :316-325, then AddWeighted(g_sum, ..., 1/n)  :328-335
      {
        const double scaling = 1.0 / n_d;
        if (mine) {
          const int64_t j = idx_c;
          const unsigned lagged = (it + 1) - lag[j];
          if (lagged != 0) {
            penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_after, gamma, beta);
            lag[j] = it + 1;
          }
          for (int k = 0; k < K; ++k) G[k + j * K] += val_c * sgc[k] * scaling;
        }
        for (int64_t q = q0 + kWave + lane; q < q1; q += kWave) {
          const int64_t j = d.idx[q];
          const unsigned lagged = (it + 1) - lag[j];
          if (lagged != 0) {
            penalty_apply_q(penalty, K, w + j * K, G + j * K, wscale, ls_at(lagged), q_after, gamma, beta);
            lag[j] = it + 1;
          }
          const double v = d.val[q];
          for (int k = 0; k < K; ++k) G[k + j * K] += v * sgc[k] * scaling;
        }
      }
      u_prev = u;
    }
    // my last draw of the epoch
    wave_mem_sync();
    if (u_prev >= 0 && lane == 0) done_slot[u_prev & (kWave - 1)] = (unsigned long long)(u_prev + 1);
    u_prev = -1;
    meet();                                                          // every draw of the epoch is in memory
    if (!ok) break;
    // Reset(n_samples): unlag and rescale  :340-348 (features dealt round robin to the wavefronts)
    {
      const double W_end = wend[0];
      asm volatile("buffer_inv sc1" ::: "memory");
      for (int64_t j = (int64_t)c * kWave + lane; j < p; j += kMc * kWave) {
        const unsigned lagged = nit - lag[j];
        if (lagged != 0) penalty_apply(penalty, K, w + j * K, G + j * K, W_end, ls_at(lagged), gamma, beta);
        for (int k = 0; k < K; ++k) w[k + j * K] *= W_end;
        lag[j] = 0u;
      }
      for (int i = c * kWave + lane; i < kMcDep; i += kMc * kWave) lastw[i] = 0u;
      for (int i = c * kWave + lane; i < kMcSam; i += kMc * kWave) lasts[i] = 0u;
      if (c == 0 && lane == 0) ctrl[5] = 0ull;
    }
    wave_mem_sync();
    meet();
    if (!ok) break;
    ++it_outer;
    if (c == 0) {
      asm volatile("buffer_inv sc1" ::: "memory");
      const int conv = convergence_check(w, d.w_prev, KP, ctl.tol, lane);    // :367
      if (lane == 0) ctrl[4] = (unsigned long long)conv;
    }
    meet();
    if (!ok) break;
    converged = (int)ctrl_load(ctrl + 4);
    base += nit;
  } while (!converged && it_outer < ctl.max_epochs);                 // :371

  if (c == 0) {
    for (int k = lane; k < K; k += kWave) {
      d.b[k] = sb[k];
      d.gb[k] = sgb[k];
    }
    if (lane == 0) {
      ctl.out[0] = (int)it_outer;
      ctl.out[1] = (!ok || ctrl_load(ctrl + 2) == 2ull) ? -2 : converged;
    }
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

      if (d.family == SGDNET_MULTINOMIAL && small_k) {               // :156-159, one exp per class lane
        const double g = softmax_gradient_lanes(lane < K ? slp[lane] : 0.0, K, lane, sy[0]);
        if (lane < K) {
          const int64_t mi = lane + (int64_t)s * K;
          sgc[lane] = g - m_cur;
          d.M[mi] = g;
          if (has_next && s1 == s) mn = g;
        }
      } else
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
// Dense variant for WIDE rows (K * p > 64, K <= 16): the same iteration, one draw at a time in stream
// order, executed by a whole workgroup.  An iteration is O(K p) work that is parallel over the
// features; only consecutive ITERATIONS depend on each other (through the linear predictor).  One
// wavefront walking 10 000 features -- and one lane adding up the dot product in ascending order, as
// saga_dense_exact_kernel does -- needs 0.8 ms per draw; here
//   * thread t owns features t, t + T, ...: it alone reads and writes their w and g_sum (LDS when
//     2 K p doubles fit, else global/L2), so those need no synchronisation at all;
//   * the linear predictor is the one cross-thread sum: per-thread partial sums, wave_sum, one LDS
//     slot per wavefront, barrier, and thread k (< K, all in wavefront 0) adds the slots in a fixed
//     order, evaluates the gradient, owns intercept k and gradient-memory entry k of every sample
//     (it alone loads and stores them: program order of one thread, no cross-wavefront visibility
//     question), and publishes the gradient change through LDS; second barrier; every thread
//     updates its features;
//   * the barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also
//     wait for the next draw's row, which is requested one iteration ahead (sample ids two ahead).
// Difference to the reference order: the dot product is summed as a tree instead of feature by
// feature (rounding only, ~1e-16 relative; the parity tests hold at 1e-10).
// --------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Wavefront sum through the DPP lane network (v_mov_b32_dpp: a few cycles per step, against an LDS
// crossbar round trip per __shfl_xor): quad swaps, half-row and row mirrors, then the row totals
// travel down the rows (row_bcast15 / row_bcast31, GFX9); lane 63 ends up with the sum of all 64.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_get(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), kCtrl, kRowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), kCtrl, kRowMask, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_get<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_get<0x140, 0xf>(v);   // row_mirror: every lane holds its row's sum
  v += dpp_get<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_get<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// penalty(w, j, w_scale, 1, g_sum) on one feature column held in registers: the arithmetic of
// penalty_apply (device_math.hpp) with f = gamma / w_scale and bg = beta * gamma formed once per iteration.
template <int KMAX>
__device__ __forceinline__ void penalty_regs(int penalty, int K, double (&wj)[KMAX], const double (&gj)[KMAX], double f,
                                             double bg, double tau, double w_scale) {
  if (penalty == SGDNET_RIDGE) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) wj[k] -= f * gj[k];
  } else if (penalty == SGDNET_ELASTICNET) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) wj[k] = soft_threshold(wj[k] - f * gj[k], tau);
  } else {
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const double v = wj[k] - f * gj[k];
        wj[k] = v;
        nrm += v * v;
      }
    }
    nrm = sqrt(nrm);
    const double factor = bg / nrm;
    if (factor < 1.0) {
      const double m = 1.0 - factor / w_scale;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) wj[k] *= m;
    } else {
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) wj[k] = 0.0;
    }
  }
}

// kU: feature chunks handled per batch (loads of a batch are issued together; the first batch of the NEXT
// draw's row is requested one iteration ahead).  kStage: w and g_sum live in LDS (typed pointers: ds_read /
// ds_write instead of flat accesses that wait on both memory counters).
template <int KMAX, int kT, int kU, bool kStage>
__global__ __launch_bounds__(kT) void saga_dense_exact_wide_kernel(SagaDev d, const LamParams* lamp, ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) double wide_smem[];
  constexpr int kNW = kT / kWave;
  const int T = (int)blockDim.x;                                      // multiple of 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
  const int K = KMAX == 1 ? 1 : d.K, Ky = KMAX == 1 ? 1 : d.Ky;
  const int64_t p = d.p, KP = (int64_t)K * p;
  // LDS carve: [part kNW][KMAX] [sgc KMAX] [red 3][kNW] [w KP][G KP]
  double* part = wide_smem;
  double* sgc = part + kNW * KMAX;
  double* red = sgc + KMAX;
  // staged state is padded to whole chunks of T features (padding stays zero: x is 0 there), so its LDS
  // accesses need no per-thread guard, only the uniform "chunk exists" test
  const int64_t C = (p + T - 1) / T;
  const int64_t KPpad = (int64_t)K * C * T;
  double* wl = red + 3 * kNW;
  double* Gl = wl + KPpad;
  (void)KP;
  // w(j, k) / g_sum(j, k) of an own feature
  auto wld = [&](int64_t i) -> double { return kStage ? wl[i] : d.w[i]; };
  auto gld = [&](int64_t i) -> double { return kStage ? Gl[i] : d.G[i]; };
  auto wst = [&](int64_t i, double v) { if (kStage) wl[i] = v; else d.w[i] = v; };
  auto gst = [&](int64_t i, double v) { if (kStage) Gl[i] = v; else d.G[i] = v; };
  if (kStage) {
    // element (k, j) is staged by the thread that owns feature j
    for (int64_t j = tid; j < C * T; j += T)
      for (int k = 0; k < K; ++k) {
        wl[j * K + k] = j < p ? d.w[j * K + k] : 0.0;
        Gl[j * K + k] = j < p ? d.G[j * K + k] : 0.0;
      }
  }
  // x(sample, j) without a branch: clamped address, zero beyond the row
  auto xload = [&](uint32_t smp, int64_t j) -> double {
    const double v = d.xd[(int64_t)smp * p + (j < p ? j : p - 1)];
    return j < p ? v : 0.0;
  };
  for (int64_t j = tid; j < p; j += T)
    for (int k = 0; k < K; ++k) d.w_prev[j * K + k] = d.w[j * K + k];   // saga-dense.h:142
  const bool cls = tid < K;                                            // thread k owns class k
  const bool ycls = tid < Ky;
  double sb_own = 0.0, sgb_own = 0.0;
  if (cls) {
    sb_own = d.b[tid];
    sgb_own = d.gb[tid];
  }
  __syncthreads();

  const int penalty = lamp->penalty;
  const int family = d.family;
  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                  // :131
  const double bg = beta * gamma * 1.0;
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
  const unsigned nit = (unsigned)ctl.nit;
  double wscale = 1.0;                                               // :129
  unsigned it_outer = 0;
  int converged = 0;
  int64_t t = ctl.stream_off;
  const int64_t t_last = ctl.stream_off + (int64_t)ctl.max_epochs * nit - 1;
  auto clampt = [&](int64_t x) { return x < t_last ? x : t_last; };

  // sample ids two draws ahead; the next draw's row (first kU chunks), response and gradient memory one ahead
  uint32_t s1 = d.stream[clampt(t)], s2 = d.stream[clampt(t + 1)];
  double xn[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t j = tid + (int64_t)u * T;
    xn[u] = xload(s1, j);
  }
  double y_n = ycls ? d.y[(int64_t)s1 * Ky + tid] : 0.0;
  double m_n = cls ? d.M[tid + (int64_t)s1 * K] : 0.0;

#ifdef SGDNET_PHASE_TIMING
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#define WIDE_STAMP(i) do { const unsigned long long now_ = clock64(); ph[i] += now_ - last_; last_ = now_; } while (0)
#else
#define WIDE_STAMP(i) ((void)0)
#endif
  do {
    for (unsigned it = 0; it < nit; ++it, ++t) {
#ifdef SGDNET_PHASE_TIMING
      unsigned long long last_ = clock64();
#endif
      const uint32_t s = s1;                                         // :152
      s1 = s2;
      s2 = d.stream[clampt(t + 2)];
      // (response and gradient memory are requested before the row: the memory counter retires in order,
      // so whoever waits for them does not also wait for the row)
      const double y_c = y_n, m_c = m_n;
      y_n = ycls ? d.y[(int64_t)s1 * Ky + tid] : 0.0;
      m_n = cls ? d.M[tid + (int64_t)s1 * K] : 0.0;                  // stale if s1 == s: replaced below
      double xc[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) xc[u] = xn[u];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t j = tid + (int64_t)u * T;
        xn[u] = xload(s1, j);
      }

      // ---- x . w: own features, wavefront sum, one slot per wavefront (:154) ----------------
      double acc[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) acc[k] = 0.0;
      for (int64_t c0 = 0; c0 < C; c0 += kU) {
        double xv[kU], wv[kU][KMAX];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int64_t j = tid + (c0 + u) * T;
          const bool on = kStage ? (c0 + u < C) : (j < p);           // staged: uniform
          xv[u] = (c0 == 0) ? xc[u] : xload(s, j);
#pragma unroll
          for (int k = 0; k < KMAX; ++k) wv[u][k] = 0.0;
          if (on) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
              if (k < K) wv[u][k] = wld(j * K + k);
          }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) acc[k] += wv[u][k] * xv[u];
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const double tot = wave_sum_dpp(acc[k]);
          if (lane == 0) part[wave * KMAX + k] = tot;
        }
      }
      WIDE_STAMP(0);
      // the scale bookkeeping does not depend on the gradient: done while the class threads work
      const double wscale_lp = wscale;
      if (wscale < kSmall) {                                         // :162-166
        for (int64_t j = tid; j < p; j += T)
          for (int k = 0; k < K; ++k) wst(j * K + k, wld(j * K + k) * wscale);
        wscale = 1.0;
      }
      wscale *= wscale_update;                                       // :168
      const double f = gamma / wscale;
      const double tau = bg / wscale;
      WIDE_STAMP(1);
      lds_barrier();
      WIDE_STAMP(2);

      if (cls) {
        // ---- linear predictor and gradient of class tid (:154-156); the other classes' values come by
        //      readlane (the class threads are lanes 0..K-1 of wavefront 0), in ascending class order
        double tot = 0.0;
        for (int wv = 0; wv < nw; ++wv) tot += part[wv * KMAX + tid];
        const double lp_own = tot * wscale_lp + sb_own;
        double g;
        if (family == SGDNET_GAUSSIAN) {
          g = lp_own - y_c;
        } else if (family == SGDNET_BINOMIAL) {
          g = 1.0 - y_c - 1.0 / (1.0 + SGD_EXP(lp_own));
        } else if (family == SGDNET_MULTINOMIAL) {
          double mx = readlane_d(lp_own, 0);
#pragma unroll
          for (int k = 1; k < KMAX; ++k) {
            const double v = readlane_d(lp_own, k);
            if (k < K) mx = v > mx ? v : mx;
          }
          const double e = SGD_EXP(lp_own - mx);
          double se = 0.0;
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            const double v = readlane_d(e, k);
            if (k < K) se += v;
          }
          const double lse = SGD_LOG(se) + mx;
          const double y0 = readlane_d(y_c, 0);                      // the class label lives in thread 0
          g = SGD_EXP(lp_own - lse);
          if ((unsigned)tid == (unsigned)(y0 + 0.5)) g -= 1.0;
        } else {
          g = lp_own - y_c;
        }
        double gck = g - m_c;                                        // :157
        d.M[tid + (int64_t)s * K] = g;                               // :158
        if (s1 == s) m_n = g;
        sgc[tid] = gck;
        if (d.fit_intercept) {                                       // :170-173
          gck = div_by_n(gck, n_d, rn_d);
          const double gbk = sgb_own + gck;
          sgb_own = gbk;
          sb_own -= gamma * (gbk + gck);
        }
      }
      WIDE_STAMP(3);
      lds_barrier();
      WIDE_STAMP(4);
      double gc[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) gc[k] = k < K ? sgc[k] : 0.0;

      for (int64_t c0 = 0; c0 < C; c0 += kU) {
        double xv[kU], wv[kU][KMAX], gv[kU][KMAX];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int64_t j = tid + (c0 + u) * T;
          const bool on = kStage ? (c0 + u < C) : (j < p);
          xv[u] = (c0 == 0) ? xc[u] : xload(s, j);
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            wv[u][k] = 0.0;
            gv[u][k] = 0.0;
          }
          if (on) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
              if (k < K) {
                wv[u][k] = wld(j * K + k);
                gv[u][k] = gld(j * K + k);
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int64_t j = tid + (c0 + u) * T;
          const bool on = kStage ? (c0 + u < C) : (j < p);
          if (on) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
              if (k < K) wv[u][k] -= gc[k] * xv[u] * f;              // :176
            penalty_regs<KMAX>(penalty, K, wv[u], gv[u], f, bg, tau, wscale);   // :179-180
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
              if (k < K) {
                gv[u][k] += div_by_n(gc[k] * xv[u], n_d, rn_d);      // :183
                wst(j * K + k, wv[u][k]);
                gst(j * K + k, gv[u][k]);
              }
            }
          }
        }
      }
      WIDE_STAMP(5);
    }

    // ---- end of the epoch: fold the scale (:188-189), ConvergenceCheck (:208) ----------------------
    double max_change = 0.0, max_size = 0.0;
    bool finite = true;
    for (int64_t j = tid; j < p; j += T) {
      for (int k = 0; k < K; ++k) {
        const int64_t i = j * K + k;
        const double v = wld(i) * wscale;
        wst(i, v);
        finite = finite && (fabs(v) <= 1.79769313486231570815e+308);
        max_change = fmax(max_change, fabs(v - d.w_prev[i]));
        max_size = fmax(max_size, fabs(v));
        d.w_prev[i] = v;
      }
    }
    wscale = 1.0;
    max_change = wave_max(max_change);
    max_size = wave_max(max_size);
    const bool wave_bad = __ballot(!finite) != 0ull;
    if (lane == 0) {
      red[wave] = max_change;
      red[kNW + wave] = max_size;
      red[2 * kNW + wave] = wave_bad ? 1.0 : 0.0;
    }
    lds_barrier();
    double mc = 0.0, ms = 0.0, bad = 0.0;
    for (int wv = 0; wv < nw; ++wv) {
      mc = fmax(mc, red[wv]);
      ms = fmax(ms, red[kNW + wv]);
      bad += red[2 * kNW + wv];
    }
    const bool all_zero = (ms == 0.0) && (mc == 0.0);
    const bool no_change = (ms != 0.0) && (mc / ms <= ctl.tol);
    converged = (bad == 0.0 && (all_zero || no_change)) ? 1 : 0;
    ++it_outer;
    lds_barrier();                                                   // red is rewritten at the end of the next epoch
  } while (!converged && it_outer < ctl.max_epochs);

  if (kStage) {
    for (int64_t j = tid; j < p; j += T)
      for (int k = 0; k < K; ++k) {
        d.w[j * K + k] = wl[j * K + k];
        d.G[j * K + k] = Gl[j * K + k];
      }
  }
  if (cls) {
    d.b[tid] = sb_own;
    d.gb[tid] = sgb_own;
  }
  if (tid == 0) {
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = converged;
#ifdef SGDNET_PHASE_TIMING
    if (d.dbg)
      for (int i = 0; i < 6; ++i) d.dbg[i] += ph[i];
#endif
  }
#undef WIDE_STAMP
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
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
  const double bg = beta * gamma * 1.0;                                // penalties.h: beta * gamma * scaling
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
      // The scale w_scale, and with it gamma / w_scale and beta gamma / w_scale, follow a data-independent
      // sequence (:162-168): lane l < kSmallPf runs the block's l + 1 multiplications (the same products in
      // the same order as one lane would form them) and the two divisions of iteration l -- two division
      // sequences per block instead of two per iteration.  A block in which the reset (:162-166) would
      // fire takes the step-by-step path.
      double ws_l = wscale;
#pragma unroll
      for (int q = 0; q < kSmallPf; ++q) ws_l = (q <= lane) ? ws_l * wscale_update : ws_l;
      const bool block_plain = (wscale >= kSmall) && (__ballot(lane < kSmallPf - 1 && ws_l < kSmall) == 0ull);
      const double f_l = gamma / ws_l;
      const double tau_l = bg / ws_l;
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
          // one class: lane j holds w_j and x_j; every lane forms its product, the ascending sum runs over
          // v_readlane operands (product rounded, then added: the reference's operations)
          const double wx = w * x;
          for (int jj = 0; jj < p; ++jj) acc += readlane_d(wx, jj);
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
          g = softmax_gradient_lanes(lp, K, lane, y_cur);
        }
        double gc = 0.0;
        if (cls) {                                                      // :156-159
          gc = g - m_old;
          Ml[k + (size_t)s * K] = g;
        }
        double f, tau;
        if (block_plain) {
          wscale = readlane_d(ws_l, u);                                 // :168, formed at the block's head
          f = readlane_d(f_l, u);
          tau = readlane_d(tau_l, u);
        } else {
          if (wscale < kSmall) {                                        // :162-166
            w *= wscale;
            wscale = 1.0;
          }
          wscale *= wscale_update;                                      // :168
          f = gamma / wscale;
          tau = bg / wscale;
        }
        if (d.fit_intercept && cls) {                                   // :170-173
          const double gck = div_by_n_exact(gc, n_d, rn_d);
          const double gbk = sgb + gck;
          sgb = gbk;
          sb -= gamma * (gbk + gck);
        }
        // ---- all coefficients: gradient step, penalty, gradient average (:176-183) ----
        // class k's change on every (k, j) lane
        const double gck = kK1 ? readlane_d(gc, 0) : __shfl(gc, k, kWave);
        const double f2 = f;        // penalties.h: (gamma / w_scale) * scaling with scaling == 1.0: the same double
        w -= gck * x * f;                                               // :176
        if (penalty == SGDNET_RIDGE) {                                  // penalty(w, j, wscale, 1.0, g_sum), :179-180
          w -= f2 * G;
        } else if (penalty == SGDNET_ELASTICNET) {
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
          const double factor = bg / nrm;
          w = factor < 1.0 ? v * (1.0 - factor / wscale) : 0.0;
          wave_sync(true);
        }
        G += div_by_n_exact(gck * x, n_d, rn_d);                        // :183
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

// --------------------------------------------------------------------------
// The small dense kernel for one response with a feeder (round 3): the iteration of saga_dense_exact_small_kernel
// <family, penalty, one class> with everything that does not depend on the coefficients done by a second wavefront.
// The producer walks the stream ahead: the drawn row (lane j: x_j), the response, the gradient memory as it stands
// and the data-independent numbers of the draw -- w_scale before and after it (with the reset of saga-dense.h:162-166),
// gamma / w_scale and beta gamma / w_scale, sixteen draws per division -- go into a ring of slots in the LDS.  The
// consumer keeps w_j, g_sum_j in registers and runs dot product (ascending order over v_readlane operands), gradient,
// intercept, step, penalty and gradient average from the slot: ~90 instructions per draw instead of ~250.
// Gradient memory and responses live in the LDS as before; the producer's copy of M[s] may predate the consumer's
// store for one of the last kSmallRing + 4 draws, so the consumer keeps the last 32 (sample, gradient) pairs in
// registers, one per lane, and takes the latest match.
// --------------------------------------------------------------------------
constexpr int kSmallRing = 16;
constexpr int kSmallHdr = 8;        // doubles per slot header: sample, y, m, w_scale before, after, f, tau, spare
constexpr size_t kSmall2FixedLds = sizeof(double) * (kSmallRing * (kWave + kSmallHdr) + kWave + 16) + 32;

template <int kFamily, int kPenalty>
__global__ __launch_bounds__(2 * kWave) void saga_dense_exact_small2_kernel(SagaDev d, const LamParams* lamp,
                                                                            ExactCtl ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = (int)d.p;
  const int n = (int)d.n;
  const unsigned nit = (unsigned)ctl.nit;
  // LDS carve: [Ml n][yl n][rx kSmallRing*64][rh kSmallRing*kSmallHdr][sx 64+16][ctrl 4 x u64]
  SGD_LDS(double)* Ml = (SGD_LDS(double)*)smem;
  SGD_LDS(double)* yl = Ml + n;
  SGD_LDS(double)* rx = yl + n;
  SGD_LDS(double)* rh = rx + kSmallRing * kWave;
  SGD_LDS(double)* sx = rh + kSmallRing * kSmallHdr;                  // [kWave + 16]: the products of the draw in hand
  volatile SGD_LDS(unsigned long long)* ctrl = (volatile SGD_LDS(unsigned long long)*)(sx + kWave + 16);   // produced, consumed, stop
  for (int i = tid; i < n; i += 2 * kWave) {
    Ml[i] = d.M[i];
    yl[i] = d.y[i];
  }
  if (tid < 4) ctrl[tid] = 0ull;
  if (tid < kWave + 16) sx[tid] = 0.0;
  __syncthreads();                                                    // the last barrier both wavefronts meet

  const double gamma = lamp->gamma, alpha = lamp->alpha, beta = lamp->beta;
  const double wscale_update = 1.0 - alpha * gamma;                   // saga-dense.h:131
  const double bg = beta * gamma * 1.0;                               // penalties.h: beta * gamma * scaling
  const int64_t total = (int64_t)ctl.max_epochs * nit;

  if (wave == 1) {
    // ================================ producer ================================
    // A batch of sixteen draws ahead: the rows of the next batch are requested before the slots of this one are
    // filled, so a row has sixteen slot fills to arrive in.
    const bool gl = lane < 16;
    const bool act = lane < p;
    auto stream_batch = [&](int64_t ub) -> uint32_t {
      const int64_t uu = ub + (lane & 15);
      return (gl && uu < total) ? d.stream[ctl.stream_off + uu] : 0u;
    };
    auto row_of = [&](uint32_t sv, int i) -> double {
      const uint32_t si = (uint32_t)__builtin_amdgcn_readlane((int)sv, i);
      return act ? d.xd[(int64_t)si * p + lane] : 0.0;
    };
    double W = 1.0;
    unsigned itp = 0;
    bool stop = false;
    int64_t freed = 0;
    uint32_t sv = stream_batch(0), sv_n = stream_batch(16);
    double xc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xc[i] = row_of(sv, i);
    for (int64_t ub = 0; ub < total && !stop; ub += 16) {
      const uint32_t sv_nn = stream_batch(ub + 32);
      double xn[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) xn[i] = row_of(sv_n, i);
      // w_scale before / after each of the sixteen draws, then gamma / after (lanes 0..15) and beta gamma / after (16..31)
      const int di = lane & 15;
      double Wb = W, Wa = W, Wnext = W;
      unsigned itnext = itp;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const double before = Wnext;
        const double after = (before < kSmall ? 1.0 : before) * wscale_update;      // :162-168
        if (q == di) {
          Wb = before;
          Wa = after;
        }
        Wnext = after;
        if (++itnext == nit) {                                      // :188-189: the epoch ends with w_scale = 1
          itnext = 0;
          Wnext = 1.0;
        }
      }
      const double quot = (lane < 16 ? gamma : bg) / Wa;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t u = ub + i;
        if (u < total && !stop) {
        if (u - freed >= kSmallRing) {
          unsigned spins = 0;
          long long spin_t0 = 0;
          for (;;) {
            freed = (int64_t)ctrl_load(ctrl + 1);
            if (u - freed < kSmallRing) break;
            if (ctrl_load(ctrl + 2) != 0ull || spin_expired(spins, spin_t0)) {
              stop = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        }
        if (u < total && !stop) {
        const int slot = (int)(u & (kSmallRing - 1));
        const uint32_t s_u = (uint32_t)__builtin_amdgcn_readlane((int)sv, i);
        const double y_u = yl[s_u], m_u = Ml[s_u];
        const double Wb_u = readlane_d(Wb, i), Wa_u = readlane_d(Wa, i);
        const double f_u = readlane_d(quot, i), tau_u = readlane_d(quot, 16 + i);
        rx[slot * kWave + lane] = xc[i];
        double hv = __longlong_as_double((long long)s_u);
        hv = lane == 1 ? y_u : hv;
        hv = lane == 2 ? m_u : hv;
        hv = lane == 3 ? Wb_u : hv;
        hv = lane == 4 ? Wa_u : hv;
        hv = lane == 5 ? f_u : hv;
        hv = lane == 6 ? tau_u : hv;
        if (lane < kSmallHdr) rh[slot * kSmallHdr + lane] = hv;
        lanes_publish();
        if (lane == 0) ctrl[0] = (unsigned long long)(u + 1);
        }
      }
      W = Wnext;
      itp = itnext;
      sv = sv_n;
      sv_n = sv_nn;
#pragma unroll
      for (int i = 0; i < 16; ++i) xc[i] = xn[i];
    }
    return;
  }

  // ================================ consumer ================================
  const bool active = lane < p;
  double w = active ? d.w[lane] : 0.0;
  double G = active ? d.G[lane] : 0.0;
  double w_prev = w;                                                  // saga-dense.h:142
  double sb = d.b[0], sgb = d.gb[0];                                  // every lane the same
  if (active) d.w_prev[lane] = w;
  const double n_d = d.n_total, rn_d = 1.0 / n_d;
  const bool fit_intercept = d.fit_intercept != 0;
  const bool plain_soft = !(wscale_update > 0.0 && bg >= 0.0);
  const LaneExpTab exp_tab{SGD_EXP_TABPTR[2 * lane], SGD_EXP_TABPTR[2 * lane + 1]};   // lane j: 2^(j/64) = hi + lo
  uint32_t hs = 0xffffffffu;                                          // lane i < 32: sample and gradient of draw u with u % 32 == i
  double hg = 0.0;
  int64_t avail = 0;
  bool stalled = false;
  // a slot, as registers
  double x_n, h_s, y_n, m_n, Wb_n, Wa_n, f_n, tau_n;
  auto read_slot = [&](int64_t u) {
    if (u >= avail) {
      unsigned spins = 0;
      long long spin_t0 = 0;
      for (;;) {
        avail = (int64_t)ctrl_load(ctrl);
        if (u < avail) break;
        if (spin_expired(spins, spin_t0)) {
          stalled = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const int slot = (int)(u & (kSmallRing - 1));
    lanes_publish();
    x_n = rx[slot * kWave + lane];
    const SGD_LDS(double)* h = rh + slot * kSmallHdr;
    h_s = h[0];
    y_n = h[1];
    m_n = h[2];
    Wb_n = h[3];
    Wa_n = h[4];
    f_n = h[5];
    tau_n = h[6];
    lanes_publish();
    if (lane == 0) ctrl[1] = (unsigned long long)(u + 1);
  };
  read_slot(0);

  unsigned it_outer = 0;
  int converged = 0;
  int64_t u = 0;
#ifdef SGDNET_PHASE_TIMING
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#define SM2_STAMP(i) do { const unsigned long long now_ = clock64(); ph[i] += now_ - last_; last_ = now_; } while (0)
#else
#define SM2_STAMP(i) ((void)0)
#endif
  do {
    double W_end = 1.0;
    for (unsigned it = 0; it < nit; ++it, ++u) {
#ifdef SGDNET_PHASE_TIMING
      unsigned long long last_ = clock64();
#endif
      const uint32_t s = (uint32_t)__double_as_longlong(h_s);        // :152
      const double x = x_n, y_cur = y_n, Wb = Wb_n, Wa = Wa_n, f = f_n, tau = tau_n;
      double m_old = m_n;
      if (u + 1 < total) read_slot(u + 1);
      if (stalled) break;
      // gradient memory: a draw of the same sample among the last 32 supersedes the producer's copy
      const int r0 = (int)(u & 31);
      {
        const unsigned mask = (unsigned)(__ballot(lane < 32 && hs == s) & 0xffffffffull);
        if (mask != 0u) {
          const unsigned rot = r0 ? ((mask >> r0) | (mask << (32 - r0))) : mask;     // bit k: lane (k + r0) % 32, age 32 - k
          const int kk = 31 - __builtin_clz(rot);
          m_old = readlane_d(hg, (kk + r0) & 31);
        }
      }
      SM2_STAMP(0);
      // linear predictor: ascending-feature sum over v_readlane operands (lanes past the last feature hold zeros), :154
      // (through the LDS, sixteen at a time: a loop of v_readlane cost ~75 cycles per feature; the +0.0 products of
      // the lanes past the last feature change no partial sum)
      sx[lane] = w * x;
      lanes_publish();
      double acc = 0.0;
      {
        double pr[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) pr[e] = sx[e];
        for (int eb = 0;;) {
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) {
            if (eb + 4 * blk < p) {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc += pr[4 * blk + e];
            }
          }
          eb += 16;
          if (eb >= p) break;
#pragma unroll
          for (int e = 0; e < 16; ++e) pr[e] = sx[eb + e];
        }
      }
      SM2_STAMP(1);
      const double lp = acc * Wb + sb;
      double g;
      if (kFamily == SGDNET_BINOMIAL)
        g = 1.0 - y_cur - 1.0 / (1.0 + sgd_exp_lanes(lp, exp_tab));
      else
        g = lp - y_cur;
      const double gc = g - m_old;                                    // :156-159
      if (lane == 0) Ml[s] = g;
      if (lane == r0) {
        hs = s;
        hg = g;
      }
      SM2_STAMP(2);
      if (Wb < kSmall) w *= Wb;                                       // :162-166 (w_scale becomes 1, then :168 -- in Wa)
      W_end = Wa;
      if (fit_intercept) {                                            // :170-173
        const double gck = div_by_n_exact(gc, n_d, rn_d);
        const double gbk = sgb + gck;
        sgb = gbk;
        sb -= gamma * (gbk + gck);
      }
      SM2_STAMP(3);
      // all coefficients: gradient step, penalty, gradient average (:176-183)
      w -= gc * x * f;                                                // :176
      if (kPenalty == SGDNET_RIDGE)
        w -= f * G;                                                   // penalty(w, j, wscale, 1.0, g_sum), :179-180
      else
        w = k1_soft(w - f * G, tau, plain_soft);
      G += div_by_n_exact(gc * x, n_d, rn_d);                         // :183
      if (!active) {
        w = 0.0;
        G = 0.0;
      }
      SM2_STAMP(4);
    }
    if (stalled) break;
    w *= W_end;                                                       // :188-189
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
  if (lane == 0) ctrl[2] = 1ull;                                      // the producer may be waiting for a free slot
#ifdef SGDNET_PHASE_TIMING
  if (d.dbg && lane == 0)
    for (int i = 0; i < 5; ++i) d.dbg[24 + i] += ph[i];
#endif
#undef SM2_STAMP

  if (active) {
    d.w[lane] = w;
    d.G[lane] = G;
    d.w_prev[lane] = w_prev;
  }
  lanes_publish();
  for (int i = lane; i < n; i += kWave) d.M[i] = Ml[i];
  if (lane == 0) {
    d.b[0] = sb;
    d.gb[0] = sgb;
    ctl.out[0] = (int)it_outer;
    ctl.out[1] = stalled ? -2 : converged;
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

template <typename Kern>
static int launch_k1_t(Kern kern, int threads, const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                       hipStream_t st) {
  SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds_bytes, st, d, lam, ctl);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// 0 when the feeder form does not apply: one response, gaussian or binomial, ridge or elastic net
size_t dense_exact_small2_lds_bytes(const SagaDev& d, int penalty, int64_t nit) {
  if (!d.xd || d.K != 1 || d.Ky != 1 || d.p > kWave || d.n > (1 << 20) || nit < 64 ||
      (d.family != SGDNET_GAUSSIAN && d.family != SGDNET_BINOMIAL) ||
      (penalty != SGDNET_RIDGE && penalty != SGDNET_ELASTICNET))
    return 0;
  const size_t b = sizeof(double) * 2 * (size_t)d.n + kSmall2FixedLds;
  return b <= 150 * 1024 ? ((b + 15) & ~size_t(15)) : 0;
}

int launch_dense_exact_small2(const SagaDev& d, int penalty, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                              hipStream_t st) {
  if (d.family == SGDNET_GAUSSIAN)
    return penalty == SGDNET_RIDGE
               ? launch_k1_t(saga_dense_exact_small2_kernel<SGDNET_GAUSSIAN, SGDNET_RIDGE>, 2 * kWave, d, lam, ctl, lds_bytes, st)
               : launch_k1_t(saga_dense_exact_small2_kernel<SGDNET_GAUSSIAN, SGDNET_ELASTICNET>, 2 * kWave, d, lam, ctl, lds_bytes, st);
  return penalty == SGDNET_RIDGE
             ? launch_k1_t(saga_dense_exact_small2_kernel<SGDNET_BINOMIAL, SGDNET_RIDGE>, 2 * kWave, d, lam, ctl, lds_bytes, st)
             : launch_k1_t(saga_dense_exact_small2_kernel<SGDNET_BINOMIAL, SGDNET_ELASTICNET>, 2 * kWave, d, lam, ctl, lds_bytes, st);
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

// The register-resident kernel: one response, explicit x.  LDS = the lag_scaling cache, plus w, g_sum and lag
// when they fit beside at least 2048 entries of it; `*ls_cache` / `*stage_state` tell the launch what was chosen.
bool sparse_exact_k1_eligible(const SagaDev& d) {
  return d.K == 1 && d.Ky == 1 && !d.standardize && d.family != SGDNET_MULTINOMIAL && d.ptr && d.idx && d.val;
}

size_t sparse_exact_k1_lds_bytes(const SagaDev& d, int64_t nit, bool allow_stage, int* ls_cache, int* stage_state) {
  const size_t fixed = kK1xFixedLds;
  const size_t cap = 160 * 1024 - 256 - fixed;
  const size_t state = (sizeof(double) * 2 + sizeof(unsigned)) * (size_t)d.p;
  const size_t want = (size_t)nit + 1;
  const size_t floor_entries = want < 2048 ? want : 2048;
  const bool stage = allow_stage && state + sizeof(double) * floor_entries <= cap;
  size_t entries = (cap - (stage ? state : 0)) / sizeof(double);
  if (entries > want) entries = want;
  *ls_cache = (int)entries;
  *stage_state = stage ? 1 : 0;
  return (fixed + sizeof(double) * entries + (stage ? state : 0) + 15) & ~size_t(15);
}


int launch_sparse_exact_k1(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, hipStream_t st) {
  return ctl.use_lds ? launch_k1_t(saga_sparse_exact_k1x_kernel<true>, 2 * kWave, d, lam, ctl, lds_bytes, st)
                     : launch_k1_t(saga_sparse_exact_k1x_kernel<false>, 2 * kWave, d, lam, ctl, lds_bytes, st);
}

// The multi-consumer kernel: LDS = its fixed part + the lag_scaling cache (w, g_sum, lag stay in memory).
size_t sparse_exact_k1m_lds_bytes(int64_t nit, int* ls_cache) {
  const size_t cap = 160 * 1024 - 256 - kK1mFixedLds;
  size_t entries = cap / sizeof(double);
  if (entries > (size_t)nit + 1) entries = (size_t)nit + 1;
  *ls_cache = (int)entries;
  return (kK1mFixedLds + sizeof(double) * entries + 15) & ~size_t(15);
}

int sparse_exact_k1m_consumers() { return kCons; }

int launch_sparse_exact_k1m(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, hipStream_t st) {
  return launch_k1_t(saga_sparse_exact_k1m_kernel, (kCons + kProd) * kWave, d, lam, ctl, lds_bytes, st);
}

// The multi-wavefront general kernel: up to 64 classes, explicit x.
bool sparse_exact_mc_eligible(const SagaDev& d) {
  return d.K <= kWave && d.Ky <= kWave && !d.standardize && d.ptr && d.idx && d.val;
}
size_t sparse_exact_mc_lds_bytes() { return (kMcFixedLds + 15) & ~size_t(15); }
int sparse_exact_mc_wavefronts() { return kMc; }
int launch_sparse_exact_mc(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, hipStream_t st) {
  return launch_k1_t(saga_sparse_exact_mc_kernel, kMc * kWave, d, lam, ctl, sparse_exact_mc_lds_bytes(), st);
}

int launch_sparse_exact(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                        hipStream_t st) {
  return ctl.use_lds ? launch_k1_t(saga_sparse_exact_kernel<true>, kWave, d, lam, ctl, lds_bytes, st)
                     : launch_k1_t(saga_sparse_exact_kernel<false>, kWave, d, lam, ctl, lds_bytes, st);
}

// Workgroup size and LDS of the wide dense kernel; 0 threads: not eligible (more than 16 classes).
// One class and 2..4 classes: 512 threads (256 VGPRs each); 5..16 classes: 256 threads (K-vectors of
// w, g_sum and the gradient change in registers).
static int wide_kmax(const SagaDev& d) { return d.K == 1 ? 1 : (d.K <= 4 ? 4 : 16); }
static int wide_cap(const SagaDev& d) { return d.K <= 4 ? 512 : 256; }

int dense_exact_wide_threads(const SagaDev& d) {
  if (!d.xd || d.K > 16) return 0;
  int64_t t = (d.p + kWave - 1) / kWave * kWave;
  if (t > wide_cap(d)) t = wide_cap(d);
  return (int)t;
}

size_t dense_exact_wide_lds_bytes(const SagaDev& d, bool stage_state) {
  const int kmax = wide_kmax(d), nwmax = wide_cap(d) / kWave;
  size_t b = sizeof(double) * (size_t)(nwmax * kmax + kmax + 3 * nwmax);
  if (stage_state) {   // padded to whole chunks of T features
    const size_t T = (size_t)dense_exact_wide_threads(d);
    const size_t ppad = ((size_t)d.p + T - 1) / T * T;
    b += sizeof(double) * 2 * (size_t)d.K * ppad;
  }
  return (b + 15) & ~size_t(15);
}

template <int KMAX, int kT, int kU, bool kStage>
static int launch_wide_t(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes, int T,
                         hipStream_t st) {
  SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_dense_exact_wide_kernel<KMAX, kT, kU, kStage>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((saga_dense_exact_wide_kernel<KMAX, kT, kU, kStage>), dim3(1), dim3(T), lds_bytes, st, d, lam, ctl);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_dense_exact_wide(const SagaDev& d, const LamParams* lam, const ExactCtl& ctl, size_t lds_bytes,
                            hipStream_t st) {
  const int T = dense_exact_wide_threads(d);
  if (T <= 0) {
    set_error("wide dense exact kernel: more than 16 classes");
    return SGDNET_EUNSUPPORTED;
  }
  const bool stage = ctl.use_lds != 0;
  if (d.K == 1) return stage ? launch_wide_t<1, 512, 8, true>(d, lam, ctl, lds_bytes, T, st)
                             : launch_wide_t<1, 512, 8, false>(d, lam, ctl, lds_bytes, T, st);
  if (d.K <= 4) return stage ? launch_wide_t<4, 512, 2, true>(d, lam, ctl, lds_bytes, T, st)
                             : launch_wide_t<4, 512, 2, false>(d, lam, ctl, lds_bytes, T, st);
  return stage ? launch_wide_t<16, 256, 1, true>(d, lam, ctl, lds_bytes, T, st)
               : launch_wide_t<16, 256, 1, false>(d, lam, ctl, lds_bytes, T, st);
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
