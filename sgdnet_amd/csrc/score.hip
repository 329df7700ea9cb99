// Prediction and prediction error along the lambda path on the device.
//
// The reference computes both on the host in R: predict.sgdnet (R/predict.sgdnet.R:347-402) is
// cbind2(1, newx) %*% beta for every lambda, score.sgdnet_<family> (R/score.R:55-186) turns the
// linear predictors into one loss per (sample, lambda) and averages over the samples.
// cv_sgdnet (R/cv_sgdnet.R:161-199) calls them n_alpha * n_folds times.  Here it is one SpMM-shaped
// kernel: a wavefront owns a sample, lane c owns the (lambda, class) pair c, the coefficient
// matrix is re-laid (p, lambda, class) so that one non-zero of the sample meets 64 contiguous
// coefficients, and the per-sample losses are reduced per lambda without leaving the kernel.
#include <cmath>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.hpp"
#include "device_math.hpp"

namespace sgdnet {
namespace {

constexpr int kScoreBlock = 256;           // 4 wavefronts, one sample each at a time
constexpr int kMaxPairs = 1024;            // (lambda, class) pairs per call (LDS: 4 x 8 KB)
constexpr int kMaxLambda = 256;            // lambdas per call (4 running sums per lane)
constexpr double kProbMin = 1e-05;         // R/score.R:88, 133

struct ScoreArgs {
  int64_t n, p;
  int family, K, Ky, L, measure;
  const int64_t* ptr;      // sparse, sample-major
  const int32_t* idx;
  const double* val;
  const double* xd;        // dense, sample-major n x p
  const double* y;         // Ky x n
  const double* a0;        // K x L  (index l * K + k)
  const double* B;         // p x L x K
  double* link;            // n x L x K or nullptr
  double* out;             // L sums or nullptr
};

// beta[k + K * (j + p * l)]  ->  B[(j * L + l) * K + k]
__global__ __launch_bounds__(256) void relayout_beta_kernel(const double* beta, int64_t p, int K, int L, double* B) {
  const int64_t total = p * (int64_t)L * K;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int k = (int)(t % K);
    const int64_t jl = t / K;
    const int l = (int)(jl % L);
    const int64_t j = jl / L;
    B[t] = beta[k + (int64_t)K * (j + p * l)];
  }
}

// one (sample, lambda) loss from the K linear predictors lp[0..K)
__device__ __forceinline__ double sample_loss(const ScoreArgs& a, const double* lp, const double* ys) {
  const int K = a.K;
  if (a.family == SGDNET_GAUSSIAN) {                       // R/score.R:55-70
    const double d = lp[0] - ys[0];
    return a.measure == SGDNET_MEASURE_MAE ? fabs(d) : d * d;
  }
  if (a.family == SGDNET_MGAUSSIAN) {                      // :172-186
    double e = 0.0;
    for (int k = 0; k < K; ++k) {
      const double d = lp[k] - ys[k];
      e += a.measure == SGDNET_MEASURE_MAE ? fabs(d) : d * d;
    }
    return e;
  }
  if (a.family == SGDNET_BINOMIAL) {                       // :74-115
    double ph = 1.0 / (1.0 + exp(-lp[0]));
    const double y2 = ys[0] > 0.5 ? 1.0 : 0.0, y1 = 1.0 - y2;
    switch (a.measure) {
      case SGDNET_MEASURE_MSE: return (ph + y1 - 1.0) * (ph + y1 - 1.0) + (ph - y2) * (ph - y2);
      case SGDNET_MEASURE_MAE: return fabs(ph + y1 - 1.0) + fabs(ph - y2);
      case SGDNET_MEASURE_CLASS: return y1 * (ph > 0.5 ? 1.0 : 0.0) + y2 * (ph <= 0.5 ? 1.0 : 0.0);
      default:
        ph = fmin(fmax(ph, kProbMin), 1.0 - kProbMin);
        return 2.0 * (0.0 - (y1 * log(1.0 - ph) + y2 * log(ph)));
    }
  }
  // multinomial, :119-168: response = exp(lp) / sum exp(lp) (R/predict.sgdnet.R:395-399)
  const int cls = (int)(ys[0] + 0.5);
  double den = 0.0;
  for (int k = 0; k < K; ++k) den += exp(lp[k]);
  double e = 0.0;
  if (a.measure == SGDNET_MEASURE_CLASS) {
    int best = 0;                                          // first maximum wins
    for (int k = 1; k < K; ++k)
      if (lp[k] > lp[best]) best = k;
    return best == cls ? 0.0 : 1.0;
  }
  for (int k = 0; k < K; ++k) {
    double ph = exp(lp[k]) / den;
    const double yk = k == cls ? 1.0 : 0.0;
    if (a.measure == SGDNET_MEASURE_MSE) {
      e += (yk - ph) * (yk - ph);
    } else if (a.measure == SGDNET_MEASURE_MAE) {
      e += fabs(yk - ph);
    } else {
      ph = fmin(fmax(ph, kProbMin), 1.0 - kProbMin);
      e += 2.0 * (0.0 - yk * log(ph));
    }
  }
  return e;
}

template <bool kSparse>
__global__ __launch_bounds__(kScoreBlock) void score_kernel(ScoreArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int LK = a.L * a.K;
  double* lp = lds + (size_t)wave * LK;
  double sums[kMaxLambda / 64] = {0.0, 0.0, 0.0, 0.0};
  const int64_t wave_id = (int64_t)blockIdx.x * (kScoreBlock / 64) + wave;
  const int64_t n_waves = (int64_t)gridDim.x * (kScoreBlock / 64);
  for (int64_t i = wave_id; i < a.n; i += n_waves) {
    for (int c0 = 0; c0 < LK; c0 += 64) {
      const int c = c0 + lane;
      const bool on = c < LK;
      double acc = 0.0;
      if (kSparse) {
        const int64_t q1 = a.ptr[i + 1];
        for (int64_t q = a.ptr[i]; q < q1; ++q) {
          const double v = a.val[q];
          const int64_t j = a.idx[q];
          if (on) acc += v * a.B[j * LK + c];
        }
      } else {
        const double* xs = a.xd + i * a.p;
        for (int64_t j = 0; j < a.p; ++j) {
          const double v = xs[j];
          if (on && v != 0.0) acc += v * a.B[j * LK + c];
        }
      }
      if (on) {
        acc += a.a0[c];
        lp[c] = acc;
        if (a.link) a.link[i * LK + c] = acc;
      }
    }
    if (a.out) {
      // the wave's own LDS row: LDS operations of one wave are served in issue order; the fence
      // keeps the compiler from moving the reads above the writes
      __threadfence_block();
      const double* ys = a.y + i * a.Ky;
#pragma unroll
      for (int r = 0; r < kMaxLambda / 64; ++r) {
        const int l = lane + 64 * r;
        if (l < a.L) sums[r] += sample_loss(a, lp + (size_t)l * a.K, ys);
      }
      __threadfence_block();
    }
  }
  if (a.out) {
#pragma unroll
    for (int r = 0; r < kMaxLambda / 64; ++r) {
      const int l = lane + 64 * r;
      if (l < a.L) __hip_atomic_fetch_add(a.out + l, sums[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- AUC (R/score.R:98-99, auc() :203-233) -------------------------------------------------------------
// score.sgdnet_binomial hands auc() the two-column indicator matrix, so the reference takes its weighted
// branch: the 2n stacked entries (prob, prob) with weights (y == class 0, y == class 1) are ordered by
// (prob, runif(2n)) and the area is sum over class-1 entries of the class-0 weight sorted before them,
// divided by n0 n1.  Entries of weight zero change nothing, so sample i takes part once, with the tie
// breaker tie[i] when it is of class 0 and tie[n + i] when it is of class 1 (no tie vector: position in the
// stacked vector, i.e. class-0 entries first).
// Two stable radix sorts (tie breaker, then probability), an exclusive scan of the class-0 flags, and an
// integer sum: exact, whatever the order of the atomic adds.
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void auc_keys_kernel(const double* link, int64_t n, int L, int l, const double* y,
                                                       const double* tie, unsigned long long* key_tie,
                                                       unsigned* idx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  idx[i] = (unsigned)i;
  // no tie vector: the order of the stacked vector itself (every class-0 entry before every class-1 entry)
  key_tie[i] = tie ? ordered_bits(y[i] < 0.5 ? tie[i] : tie[n + i])
                   : (((y[i] < 0.5) ? 0ull : 1ull) << 32) | (unsigned long long)i;
  (void)link; (void)L; (void)l;
}

// probability of the samples in the order of idx: R/predict.sgdnet.R type = "response", 1 / (1 + exp(-eta))
__global__ __launch_bounds__(256) void auc_prob_kernel(const double* link, int64_t n, int L, int l, const unsigned* idx,
                                                       unsigned long long* key_prob) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const double eta = link[(int64_t)idx[q] * L + l];
  key_prob[q] = ordered_bits(1.0 / (1.0 + exp(-eta)));
}

__global__ __launch_bounds__(256) void auc_flags_kernel(const double* y, int64_t n, const unsigned* idx, unsigned* neg) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) neg[q] = y[idx[q]] < 0.5 ? 1u : 0u;
}

__global__ __launch_bounds__(256) void auc_sum_kernel(const unsigned* neg, const unsigned* before, int64_t n,
                                                      unsigned long long* u) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long v = (q < n && neg[q] == 0u) ? (unsigned long long)before[q] : 0ull;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(u, v);
}

struct DevBufs {
  std::vector<void*> all;
  ~DevBufs() {
    for (void* q : all) (void)hipFree(q);
  }
  template <typename T>
  int upload(T** out, const T* host, size_t count, hipStream_t st) {
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(T) * (count ? count : 1)) != hipSuccess) {
      set_error("hipMalloc(%zu bytes) failed", sizeof(T) * count);
      return SGDNET_ENOMEM;
    }
    all.push_back(q);
    if (host && count) SGD_HIP_TRY(hipMemcpyAsync(q, host, sizeof(T) * count, hipMemcpyHostToDevice, st));
    *out = static_cast<T*>(q);
    return SGDNET_OK;
  }
};

int run_score(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
              const double* x_dense, const double* y, int y_rows, int family, int n_classes, const double* a0,
              const double* beta, int n_lambda, int measure, int device, double* out, double* link,
              const double* tie = nullptr, sgdnet_rng* rng = nullptr) {
  const bool auc = measure == SGDNET_MEASURE_AUC;
  if (n < 1 || p < 1 || n_lambda < 1 || n_classes < 1 || !a0 || !beta || (!out && !link) ||
      (out && (!y || y_rows < 1)) || family < SGDNET_GAUSSIAN || family > SGDNET_MGAUSSIAN ||
      measure < SGDNET_MEASURE_DEVIANCE || measure > SGDNET_MEASURE_AUC ||
      (measure == SGDNET_MEASURE_CLASS && family != SGDNET_BINOMIAL && family != SGDNET_MULTINOMIAL) ||
      (auc && (family != SGDNET_BINOMIAL || n_classes != 1 || !out || link || n >= (1ll << 31))) ||
      (!x_dense && !(rowptr && colidx && values))) {
    set_error("sgdnet_score/predict: bad argument");
    return SGDNET_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    set_error("no HIP device");
    return SGDNET_ENODEVICE;
  }
  if (device < 0 || device >= ndev) {
    set_error("device %d out of range (%d devices)", device, ndev);
    return SGDNET_EINVAL;
  }
  SGD_HIP_TRY(hipSetDevice(device));
  hipStream_t st = nullptr;
  SGD_HIP_TRY(hipStreamCreate(&st));
  struct StreamGuard {
    hipStream_t s;
    ~StreamGuard() { (void)hipStreamDestroy(s); }
  } guard{st};
  DevBufs bufs;
  ScoreArgs a{};
  a.n = n;
  a.p = p;
  a.family = family;
  a.K = n_classes;
  a.Ky = y_rows > 0 ? y_rows : 1;
  a.measure = measure;
  int rc;
  if (x_dense) {
    double* xd;
    if ((rc = bufs.upload(&xd, x_dense, (size_t)n * p, st))) return rc;
    a.xd = xd;
  } else {
    int64_t* ptr;
    int32_t* idx;
    double* val;
    const size_t nnz = (size_t)rowptr[n];
    if ((rc = bufs.upload(&ptr, rowptr, (size_t)n + 1, st)) || (rc = bufs.upload(&idx, colidx, nnz, st)) ||
        (rc = bufs.upload(&val, values, nnz, st)))
      return rc;
    a.ptr = ptr;
    a.idx = idx;
    a.val = val;
  }
  if (y) {
    double* yd;
    if ((rc = bufs.upload(&yd, y, (size_t)n * a.Ky, st))) return rc;
    a.y = yd;
  }
  // lambdas in chunks that fit the kernel's LDS rows and running sums
  int chunk = kMaxPairs / n_classes;
  if (chunk > kMaxLambda) chunk = kMaxLambda;
  if (chunk < 1) {
    set_error("sgdnet_score/predict: more than %d classes", kMaxPairs);
    return SGDNET_EUNSUPPORTED;
  }
  double *beta_d, *B, *a0_d, *out_d = nullptr, *link_d = nullptr;
  const size_t K = (size_t)n_classes;
  if ((rc = bufs.upload(&beta_d, beta, K * p * n_lambda, st)) || (rc = bufs.upload(&a0_d, a0, K * n_lambda, st)) ||
      (rc = bufs.upload<double>(&B, nullptr, K * p * chunk, st)))
    return rc;
  if (out && (rc = bufs.upload<double>(&out_d, nullptr, (size_t)n_lambda, st))) return rc;
  if (out) SGD_HIP_TRY(hipMemsetAsync(out_d, 0, sizeof(double) * n_lambda, st));
  if ((link || auc) && (rc = bufs.upload<double>(&link_d, nullptr, (size_t)n * K * chunk, st))) return rc;
  // AUC scratch: keys, permutation, flags, prefix counts, one integer sum per lambda, radix sort / scan storage
  unsigned long long *key_a = nullptr, *key_b = nullptr, *u_d = nullptr;
  unsigned *idx_a = nullptr, *idx_b = nullptr, *neg_d = nullptr, *before_d = nullptr;
  double* tie_d = nullptr;
  uint32_t *rng_d = nullptr, *raw_d = nullptr;
  void* tmp_d = nullptr;
  size_t tmp_bytes = 0;
  int64_t n1 = 0;
  if (auc) {
    for (int64_t i = 0; i < n; ++i) {
      if (y[i] != 0.0 && y[i] != 1.0) {
        set_error("sgdnet_score: auc needs class codes 0 / 1 in y");
        return SGDNET_EINVAL;
      }
      n1 += y[i] == 1.0;
    }
    if ((rc = bufs.upload<unsigned long long>(&key_a, nullptr, (size_t)n, st)) ||
        (rc = bufs.upload<unsigned long long>(&key_b, nullptr, (size_t)n, st)) ||
        (rc = bufs.upload<unsigned long long>(&u_d, nullptr, (size_t)n_lambda, st)) ||
        (rc = bufs.upload<unsigned>(&idx_a, nullptr, (size_t)n, st)) || (rc = bufs.upload<unsigned>(&idx_b, nullptr, (size_t)n, st)) ||
        (rc = bufs.upload<unsigned>(&neg_d, nullptr, (size_t)n, st)) || (rc = bufs.upload<unsigned>(&before_d, nullptr, (size_t)n, st)))
      return rc;
    if ((tie || rng) && (rc = bufs.upload<double>(&tie_d, nullptr, (size_t)2 * n, st))) return rc;
    // tie breakers drawn here: the caller's generator moves to the device and back (R/score.R:221: runif(2 n) per lambda)
    if (rng && ((rc = bufs.upload<uint32_t>(&rng_d, reinterpret_cast<const uint32_t*>(rng), sizeof(sgdnet_rng) / 4, st)) ||
                (rc = bufs.upload<uint32_t>(&raw_d, nullptr, (size_t)2 * n, st))))
      return rc;
    SGD_HIP_TRY(hipMemsetAsync(u_d, 0, sizeof(unsigned long long) * n_lambda, st));
    size_t b1 = 0, b2 = 0;
    SGD_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, key_a, key_b, idx_a, idx_b, (int)n, 0, 64, st));
    SGD_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, neg_d, before_d, (int)n, st));
    tmp_bytes = b1 > b2 ? b1 : b2;
    char* tq = nullptr;
    if ((rc = bufs.upload<char>(&tq, nullptr, tmp_bytes, st))) return rc;
    tmp_d = tq;
  }
  int64_t grid = (n + 3) / 4;
  if (grid > 8192) grid = 8192;
  for (int l0 = 0; l0 < n_lambda; l0 += chunk) {
    const int L = n_lambda - l0 < chunk ? n_lambda - l0 : chunk;
    int rgrid = (int)((K * p * L + 255) / 256);
    if (rgrid > 4096) rgrid = 4096;
    hipLaunchKernelGGL(relayout_beta_kernel, dim3(rgrid), dim3(256), 0, st, beta_d + K * p * l0, p, n_classes, L, B);
    a.L = L;
    a.a0 = a0_d + K * l0;
    a.B = B;
    a.out = (out && !auc) ? out_d + l0 : nullptr;
    a.link = link_d;
    const size_t lds = sizeof(double) * (size_t)(kScoreBlock / 64) * L * K;
    if (x_dense)
      hipLaunchKernelGGL(score_kernel<false>, dim3((unsigned)grid), dim3(kScoreBlock), lds, st, a);
    else
      hipLaunchKernelGGL(score_kernel<true>, dim3((unsigned)grid), dim3(kScoreBlock), lds, st, a);
    SGD_HIP_TRY(hipGetLastError());
    if (auc) {
      const unsigned g = (unsigned)((n + 255) / 256);
      for (int l = 0; l < L; ++l) {
        if (rng) {
          if ((rc = launch_rng_unif(rng_d, rng_d, raw_d, tie_d, 2 * n, st))) return rc;
        } else if (tie) {
          SGD_HIP_TRY(hipMemcpyAsync(tie_d, tie + (size_t)2 * n * (l0 + l), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
        }
        hipLaunchKernelGGL(auc_keys_kernel, dim3(g), dim3(256), 0, st, link_d, n, L, l, a.y, tie_d, key_a, idx_a);
        // stable order by the tie breaker first, then by the probability
        SGD_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp_d, tmp_bytes, key_a, key_b, idx_a, idx_b, (int)n, 0, 64, st));
        const unsigned* order = idx_b;
        hipLaunchKernelGGL(auc_prob_kernel, dim3(g), dim3(256), 0, st, link_d, n, L, l, order, key_a);
        unsigned* sorted = idx_a;
        SGD_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp_d, tmp_bytes, key_a, key_b, order, sorted, (int)n, 0, 64, st));
        hipLaunchKernelGGL(auc_flags_kernel, dim3(g), dim3(256), 0, st, a.y, n, sorted, neg_d);
        SGD_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_d, tmp_bytes, neg_d, before_d, (int)n, st));
        hipLaunchKernelGGL(auc_sum_kernel, dim3(g), dim3(256), 0, st, neg_d, before_d, n, u_d + l0 + l);
        SGD_HIP_TRY(hipGetLastError());
      }
    }
    if (link) {
      // host layout: link[(i * n_lambda + l) * K + k]
      SGD_HIP_TRY(hipMemcpy2DAsync(link + K * l0, sizeof(double) * K * n_lambda, link_d, sizeof(double) * K * L,
                                   sizeof(double) * K * L, (size_t)n, hipMemcpyDeviceToHost, st));
      SGD_HIP_TRY(hipStreamSynchronize(st));
    }
  }
  if (auc) {
    std::vector<unsigned long long> u((size_t)n_lambda);
    if (rng) SGD_HIP_TRY(hipMemcpyAsync(rng, rng_d, sizeof(sgdnet_rng), hipMemcpyDeviceToHost, st));
    SGD_HIP_TRY(hipMemcpyAsync(u.data(), u_d, sizeof(unsigned long long) * n_lambda, hipMemcpyDeviceToHost, st));
    SGD_HIP_TRY(hipStreamSynchronize(st));
    // exp(log(sum) - log(n1) - log(n0)), as R/score.R:225-227 forms it
    for (int l = 0; l < n_lambda; ++l)
      out[l] = std::exp(std::log((double)u[(size_t)l]) - std::log((double)n1) - std::log((double)(n - n1)));
  } else if (out) {
    SGD_HIP_TRY(hipMemcpyAsync(out, out_d, sizeof(double) * n_lambda, hipMemcpyDeviceToHost, st));
    SGD_HIP_TRY(hipStreamSynchronize(st));
    // R/score.R: mean over the samples; mgaussian: colSums over the samples, mean over the responses
    const double scale = family == SGDNET_MGAUSSIAN ? 1.0 / (double)n_classes : 1.0 / (double)n;
    for (int l = 0; l < n_lambda; ++l) out[l] *= scale;
  }
  SGD_HIP_TRY(hipStreamSynchronize(st));
  return SGDNET_OK;
}

}  // namespace
}  // namespace sgdnet

extern "C" {

int sgdnet_score_sparse(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                        const double* y, int y_rows, int family, int n_classes, const double* a0,
                        const double* beta, int n_lambda, int measure, int device, double* out) {
  return sgdnet::run_score(n, p, rowptr, colidx, values, nullptr, y, y_rows, family, n_classes, a0, beta, n_lambda,
                           measure, device, out, nullptr);
}

int sgdnet_score_dense(const double* x, int64_t n, int64_t p, const double* y, int y_rows, int family,
                       int n_classes, const double* a0, const double* beta, int n_lambda, int measure, int device,
                       double* out) {
  return sgdnet::run_score(n, p, nullptr, nullptr, nullptr, x, y, y_rows, family, n_classes, a0, beta, n_lambda,
                           measure, device, out, nullptr);
}

int sgdnet_predict_sparse(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                          int n_classes, const double* a0, const double* beta, int n_lambda, int device,
                          double* link) {
  return sgdnet::run_score(n, p, rowptr, colidx, values, nullptr, nullptr, 0, SGDNET_GAUSSIAN, n_classes, a0, beta,
                           n_lambda, SGDNET_MEASURE_DEVIANCE, device, nullptr, link);
}

int sgdnet_predict_dense(const double* x, int64_t n, int64_t p, int n_classes, const double* a0, const double* beta,
                         int n_lambda, int device, double* link) {
  return sgdnet::run_score(n, p, nullptr, nullptr, nullptr, x, nullptr, 0, SGDNET_GAUSSIAN, n_classes, a0, beta,
                           n_lambda, SGDNET_MEASURE_DEVIANCE, device, nullptr, link);
}

int sgdnet_auc_sparse(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                      const double* y, const double* a0, const double* beta, int n_lambda, const double* tie,
                      int device, double* out) {
  return sgdnet::run_score(n, p, rowptr, colidx, values, nullptr, y, 1, SGDNET_BINOMIAL, 1, a0, beta, n_lambda,
                           SGDNET_MEASURE_AUC, device, out, nullptr, tie);
}

int sgdnet_auc_dense(const double* x, int64_t n, int64_t p, const double* y, const double* a0, const double* beta,
                     int n_lambda, const double* tie, int device, double* out) {
  return sgdnet::run_score(n, p, nullptr, nullptr, nullptr, x, y, 1, SGDNET_BINOMIAL, 1, a0, beta, n_lambda,
                           SGDNET_MEASURE_AUC, device, out, nullptr, tie);
}

int sgdnet_auc_sparse_rng(int64_t n, int64_t p, const int64_t* rowptr, const int32_t* colidx, const double* values,
                          const double* y, const double* a0, const double* beta, int n_lambda, sgdnet_rng* rng,
                          int device, double* out) {
  if (!rng) {
    sgdnet::set_error("sgdnet_auc_sparse_rng: no generator");
    return SGDNET_EINVAL;
  }
  return sgdnet::run_score(n, p, rowptr, colidx, values, nullptr, y, 1, SGDNET_BINOMIAL, 1, a0, beta, n_lambda,
                           SGDNET_MEASURE_AUC, device, out, nullptr, nullptr, rng);
}

int sgdnet_auc_dense_rng(const double* x, int64_t n, int64_t p, const double* y, const double* a0, const double* beta,
                         int n_lambda, sgdnet_rng* rng, int device, double* out) {
  if (!rng) {
    sgdnet::set_error("sgdnet_auc_dense_rng: no generator");
    return SGDNET_EINVAL;
  }
  return sgdnet::run_score(n, p, nullptr, nullptr, nullptr, x, y, 1, SGDNET_BINOMIAL, 1, a0, beta, n_lambda,
                           SGDNET_MEASURE_AUC, device, out, nullptr, nullptr, rng);
}

}  // extern "C"
