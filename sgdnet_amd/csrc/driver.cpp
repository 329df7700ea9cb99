// Lambda-path driver behind sgdnet_fit_sparse / sgdnet_fit_dense.
//
// Mirrors SetupSgdnet (reference src/sgdnet.cpp:119-285): preprocess, lambda
// path, step sizes, then for every lambda {SAGA loop, deviance, rescale} with
// the solver state kept resident in HBM across the path (warm starts,
// src/sgdnet.cpp:187-198).  For sparse x the per-fit O(nnz) setup passes run on
// the device (setup_device.hip; SGDNET_HOST_SETUP=1 keeps the host loops below for
// A/B checks); the SAGA loop and the per-lambda deviance pass run on the GPU and
// there is no CPU fallback.
#include <math.h>
#include <string.h>

#include <cmath>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "common.hpp"
#include "setup_device.hpp"

using namespace sgdnet;

namespace {

// SGDNET_TRACE=1: wall-clock of the driver's phases on stderr
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  PhaseTimer() : on(getenv("SGDNET_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sgdnet] %-28s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
};

// Shortest staleness window the driver uses.  The exported rule (sgdnet_auto_batch) floors at 64 draws, and
// for dense x with a dominant common factor (or few, strongly scaled features) 2 L_max / L_F is well below
// that: a 64-draw window then oscillates or settles on a wrong point without tripping a guard (a random
// sweep of 30-lambda paths found deviance ratios off by 0.06-0.6).  Below 8 draws a batch is no longer
// worth its launch: mode = auto takes the exact iteration there.
constexpr int64_t kWindowFloor = 8;
constexpr int64_t kMaxBatchesPerEpoch = 16384;
constexpr int64_t kRetryWindowMin = kWindowFloor;     // shortest window the divergence restarts go down to
thread_local bool t_batched_diverged = false;   // set when a batched fit gave up: mode = auto then reruns the fit in exact mode

struct Features {
  bool sparse = false;
  int64_t n = 0, p = 0;
  // feature-major (as passed by R), preprocessed values
  const int32_t* colptr = nullptr;
  const int32_t* rowidx = nullptr;
  std::vector<double> val;      // sparse values (scaled if standardize)
  std::vector<double> xd;       // dense n x p column-major (standardised if requested)
  // sample-major
  std::vector<int64_t> sptr;
  std::vector<int32_t> sidx;
  std::vector<double> sval;
  std::vector<double> xt;       // dense p x n
  std::vector<double> x_center, x_scale, x_center_scaled;
  // device-side setup (sparse): the O(nnz) passes run in setup_device.hip and the vectors
  // above other than x_center / x_scale stay empty
  DeviceSetup* dev = nullptr;
  hipStream_t st = nullptr;
  double dev_max_mean_sq = 0.0;
  bool dense_dev = false;       // dense x prepared by dense_setup_* (large matrices)
};

// math.h:66-79 Mean / :114-130 StandardDeviation (population sd, 0 -> 1)
// Host-side passes over dense x (R hands over a column-major matrix): per-column work is split
// over a few threads.  Every column is still reduced by one thread in the reference's order, so
// the results are bitwise those of the serial loops.
template <class F>
void parallel_for(int64_t count, double work, F f) {
  unsigned T = std::thread::hardware_concurrency();
  if (T > 16) T = 16;
  if (work < 4e6 || T < 2 || count < 2) {
    f((int64_t)0, count);
    return;
  }
  if ((int64_t)T > count) T = (unsigned)count;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) {
    const int64_t lo = count * t / T, hi = count * (t + 1) / T;
    th.emplace_back([=, &f] { f(lo, hi); });
  }
  for (auto& x : th) x.join();
}

void col_mean_sd(const double* x, int64_t n, int64_t m, double* mean, double* sd) {
  parallel_for(m, (double)n * (double)m, [&](int64_t j0, int64_t j1) {
  for (int64_t j = j0; j < j1; ++j) {
    const double* col = x + j * n;
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += col[i];
    mean[j] = s / (double)n;
    double v = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      const double dlt = col[i] - mean[j];
      v += dlt * dlt;
    }
    v /= (double)n;
    sd[j] = (v == 0.0) ? 1.0 : sqrt(v);
  }
  });
}

void standardize_cols(double* x, int64_t n, int64_t m, const double* mean, const double* sd) {
  parallel_for(m, (double)n * (double)m, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; ++j)
      for (int64_t i = 0; i < n; ++i) x[i + j * n] = (x[i + j * n] - mean[j]) / sd[j];
  });
}

// sample-major copy of a column-major n x p matrix (utils.h:283-288), in cache-sized tiles
void transpose_to_sample_major(const double* xd, int64_t n, int64_t p, double* xt) {
  constexpr int64_t kTile = 64;
  const int64_t row_tiles = (n + kTile - 1) / kTile;
  parallel_for(row_tiles, (double)n * (double)p, [&](int64_t t0, int64_t t1) {
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t i0 = t * kTile, i1 = std::min(n, i0 + kTile);
      for (int64_t j0 = 0; j0 < p; j0 += kTile) {
        const int64_t j1 = std::min(p, j0 + kTile);
        for (int64_t i = i0; i < i1; ++i)
          for (int64_t j = j0; j < j1; ++j) xt[j + i * p] = xd[i + j * n];
      }
    }
  });
}

// x^T v for every feature column; v is n x cols column-major
int xt_times(const Features& X, const double* v, int cols, double* out) {
  if (X.dev) return X.dense_dev ? dense_xt_times(*X.dev, v, cols, out, X.st) : device_xt_times(*X.dev, v, cols, out, X.st);
  for (int c = 0; c < cols; ++c) {
    const double* vc = v + (int64_t)c * X.n;
    parallel_for(X.p, X.sparse ? 0.0 : (double)X.n * (double)X.p, [&](int64_t j0, int64_t j1) {
      for (int64_t j = j0; j < j1; ++j) {
        double s = 0.0;
        if (X.sparse) {
          for (int64_t q = X.colptr[j]; q < X.colptr[j + 1]; ++q) s += X.val[(size_t)q] * vc[X.rowidx[q]];
        } else {
          const double* col = X.xd.data() + j * X.n;
          for (int64_t i = 0; i < X.n; ++i) s += col[i] * vc[i];
        }
        out[j + (int64_t)c * X.p] = s;
      }
    });
  }
  return SGDNET_OK;
}

double log_sum_exp_host(const double* x, int K) {
  double mx = x[0];
  for (int k = 1; k < K; ++k) mx = std::max(mx, x[k]);
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += exp(x[k] - mx);
  return log(s) + mx;
}

double binomial_link(double ybar) {     // families.h:141-150
  const double pmin = 1e-9, pmax = 1.0 - pmin;
  const double z = ybar > pmax ? pmax : (ybar < pmin ? pmin : ybar);
  return log(z / (1.0 - z));
}

// Family::FitNullModel (families.h:112-117,190-201,287-298,380-385); yt is Ky x n
void fit_null_model(int family, int K, const double* yt, int Ky, int64_t n, bool fit_intercept,
                    double* b0) {
  if (family == SGDNET_GAUSSIAN || family == SGDNET_MGAUSSIAN) {
    for (int k = 0; k < Ky; ++k) {
      double s = 0.0;
      for (int64_t i = 0; i < n; ++i) s += yt[k + i * Ky];
      b0[k] = s / (double)n;
    }
  } else if (family == SGDNET_BINOMIAL) {
    if (fit_intercept) {
      double s = 0.0;
      for (int64_t i = 0; i < n; ++i) s += yt[i];
      b0[0] = binomial_link(s / (double)n);
    } else {
      b0[0] = 0.0;
    }
  } else {
    if (fit_intercept) {
      for (int k = 0; k < K; ++k) b0[k] = 0.0;
      for (int64_t i = 0; i < n; ++i) b0[(int64_t)(yt[i] + 0.5)] += 1.0 / (double)n;
    } else {
      for (int k = 0; k < K; ++k) b0[k] = 1.0 / (double)K;
    }
    double ls = 0.0;
    for (int k = 0; k < K; ++k) ls += log(b0[k]);
    for (int k = 0; k < K; ++k) b0[k] = log(b0[k]) - ls / (double)K;
  }
}

// Family::NullDeviance (families.h:98-110,170-188,262-285,367-378); yt is Ky x n
double null_deviance(int family, int K, const double* yt, int Ky, int64_t n, bool fit_intercept) {
  std::vector<double> lp((size_t)std::max(K, Ky));
  double loss = 0.0;
  if (family == SGDNET_GAUSSIAN || family == SGDNET_MGAUSSIAN) {
    fit_null_model(family, K, yt, Ky, n, true, lp.data());
    for (int64_t i = 0; i < n; ++i) {
      double s = 0.0;
      for (int k = 0; k < Ky; ++k) {
        const double dlt = lp[(size_t)k] - yt[k + i * Ky];
        s += dlt * dlt;
      }
      loss += 0.5 * s;
    }
  } else if (family == SGDNET_BINOMIAL) {
    fit_null_model(family, K, yt, Ky, n, fit_intercept, lp.data());
    for (int64_t i = 0; i < n; ++i) loss += log(1.0 + exp(lp[0])) - yt[i] * lp[0];
  } else {
    fit_null_model(family, K, yt, Ky, n, fit_intercept, lp.data());
    const double lse = log_sum_exp_host(lp.data(), K);
    for (int64_t i = 0; i < n; ++i) loss += lse - lp[(size_t)(unsigned)(yt[i] + 0.5)];
  }
  return 2.0 * loss;
}

// Family::LambdaMax (families.h:119-126,203-220,300-325,387-406); y is n x Ky, preprocessed
double lambda_max(int family, int K, const Features& X, const double* y, int Ky, const double* y_scale) {
  const int64_t n = X.n, p = X.p;
  double best = 0.0;
  if (family == SGDNET_GAUSSIAN) {
    std::vector<double> xty((size_t)p);
    if (xt_times(X, y, 1, xty.data())) return NAN;
    for (int64_t j = 0; j < p; ++j) best = std::max(best, fabs(xty[(size_t)j]));
    return y_scale[0] * best / (double)n;
  }
  if (family == SGDNET_BINOMIAL) {
    double ybar, ystd;
    col_mean_sd(y, n, 1, &ybar, &ystd);
    std::vector<double> ymap((size_t)n), xty((size_t)p);
    for (int64_t i = 0; i < n; ++i) ymap[(size_t)i] = (y[i] - ybar) / ystd;
    if (xt_times(X, ymap.data(), 1, xty.data())) return NAN;
    for (int64_t j = 0; j < p; ++j) best = std::max(best, fabs(xty[(size_t)j]));
    return ystd * best / (double)n;
  }
  if (family == SGDNET_MULTINOMIAL) {
    std::vector<double> ymap((size_t)(n * K), 0.0), xty((size_t)(p * K)), ybar((size_t)K), ystd((size_t)K);
    for (int64_t i = 0; i < n; ++i) ymap[(size_t)(i + (int64_t)(unsigned)(y[i] + 0.5) * n)] = 1.0;
    col_mean_sd(ymap.data(), n, K, ybar.data(), ystd.data());
    standardize_cols(ymap.data(), n, K, ybar.data(), ystd.data());
    if (xt_times(X, ymap.data(), K, xty.data())) return NAN;
    for (int k = 0; k < K; ++k)
      for (int64_t j = 0; j < p; ++j)
        best = std::max(best, fabs(xty[(size_t)(j + (int64_t)k * p)] * ystd[(size_t)k]));
    return best / (double)n;
  }
  std::vector<double> ymap(y, y + n * Ky), xty((size_t)(p * Ky)), ybar((size_t)Ky), ystd((size_t)Ky);
  col_mean_sd(y, n, Ky, ybar.data(), ystd.data());
  standardize_cols(ymap.data(), n, Ky, ybar.data(), ystd.data());
  if (xt_times(X, ymap.data(), Ky, xty.data())) return NAN;
  for (int64_t j = 0; j < p; ++j) {
    double s = 0.0;
    for (int k = 0; k < Ky; ++k) {
      const double v = xty[(size_t)(j + (int64_t)k * p)] * (y_scale[k] * ystd[(size_t)k]);
      s += v * v;
    }
    best = std::max(best, sqrt(s));
  }
  return best / (double)n;
}

// Where the sample order comes from (include/sgdnet_hip.h "sample order").
struct DrawSource {
  const sgdnet_control* ctl;
  sgdnet_rng rng;
  int64_t pos = 0;
  explicit DrawSource(const sgdnet_control* c) : ctl(c) {
    if (c->rng_state) rng = *c->rng_state;
    else sgdnet_rng_seed(&rng, c->seed);
  }
  void finish() const {
    if (internal() && ctl->rng_state) *ctl->rng_state = rng;
  }
  bool internal() const { return !ctl->sample_stream && !ctl->unif; }
  // an explicit stream names samples of the whole data set: it cannot be laid out per shard
  bool shardable() const { return !ctl->sample_stream; }
  double next_unif() {
    if (!ctl->unif) return sgdnet_rng_unif(&rng);
    double u;
    do {
      u = ctl->unif(ctl->unif_ctx);
    } while (u <= 0.0 || u >= 1.0);
    return u;
  }
  // One epoch of `count` draws over n samples.  shards > 1: the layout the virtual-shard kernels
  // read (include/sgdnet_hip.h: sgdnet_solver_set_virtual_shards) -- count / shards entries per
  // shard, shard after shard, entry t of shard v drawn uniformly from shard v's sample range; the
  // generator is still advanced by exactly `count` uniforms, like the reference's epoch.
  int fill(uint32_t n, uint32_t* out, int64_t count, int shards = 1) {
    if (ctl->sample_stream) {
      if (pos + count > ctl->sample_stream_len) {
        set_error("explicit sample stream exhausted: %lld draws requested, %lld supplied",
                  (long long)(pos + count), (long long)ctl->sample_stream_len);
        return SGDNET_ESTREAM;
      }
      for (int64_t i = 0; i < count; ++i) {
        const uint32_t v = ctl->sample_stream[pos + i];
        if (v >= n) {
          set_error("sample_stream[%lld] = %u is not a sample index (n_samples = %u)", (long long)(pos + i), v, n);
          return SGDNET_EINVAL;
        }
        out[i] = v;
      }
    } else if (shards > 1) {
      const int64_t dps = count / shards, base = (int64_t)n / shards, rem = (int64_t)n % shards;
      int64_t i = 0;
      for (int v = 0; v < shards; ++v) {
        const int64_t lo = (int64_t)v * base + std::min<int64_t>(v, rem);
        const double size = (double)(base + (v < rem ? 1 : 0));
        for (int64_t t = 0; t < dps; ++t) out[i++] = (uint32_t)(lo + (int64_t)floor(size * next_unif()));
      }
      for (; i < count; ++i) {       // positions no shard consumes
        (void)next_unif();
        out[i] = 0;
      }
    } else if (ctl->unif) {
      const double nd = (double)n;
      for (int64_t i = 0; i < count; ++i) out[i] = (uint32_t)floor(nd * next_unif());
    } else {
      sgdnet_rng_fill(&rng, n, out, count);
    }
    pos += count;
    return SGDNET_OK;
  }
};

// largest eigenvalue of Xs'Xs / m for an m x p column-major sample of the rows (power iteration)
double sample_gram_lmax(const double* xs, size_t m, size_t p) {
  std::vector<double> v(p, 1.0 / std::sqrt((double)p)), w(p), u(m);
  double lmax = 0.0;
  for (int it = 0; it < 30; ++it) {
    std::fill(u.begin(), u.end(), 0.0);
    for (size_t j = 0; j < p; ++j) {
      const double* col = xs + j * m;
      const double vj = v[j];
      for (size_t r = 0; r < m; ++r) u[r] += col[r] * vj;
    }
    double nrm = 0.0;
    for (size_t j = 0; j < p; ++j) {
      const double* col = xs + j * m;
      double sacc = 0.0;
      for (size_t r = 0; r < m; ++r) sacc += col[r] * u[r];
      w[j] = sacc / (double)m;
      nrm += w[j] * w[j];
    }
    nrm = std::sqrt(nrm);
    if (!(nrm > 0.0)) break;
    const double prev = lmax;
    lmax = nrm;
    for (size_t j = 0; j < p; ++j) v[j] = w[j] / nrm;
    if (it >= 3 && std::fabs(lmax - prev) <= 2e-3 * lmax) break;
  }
  return lmax;
}

int64_t auto_batch(const Features& X, double max_sample_sqnorm, double* l_f = nullptr) {
  // largest mean squared feature value = largest diagonal entry of X'X/n
  double diag = 0.0;
  for (int64_t j = 0; j < X.p; ++j) {
    double s = 0.0;
    if (X.sparse) {
      for (int64_t q = X.colptr[j]; q < X.colptr[j + 1]; ++q) s += X.val[(size_t)q] * X.val[(size_t)q];
    } else {
      const double* col = X.xd.data() + (size_t)j * (size_t)X.n;
      for (int64_t i = 0; i < X.n; ++i) s += col[i] * col[i];
    }
    diag = std::max(diag, s / (double)X.n);
  }
  // The diagonal only bounds the largest eigenvalue of X'X/n from below; strongly collinear
  // columns (abalone: 8 size measurements of one animal) have an L_F several times their
  // diagonal and the window comes out that many times too long.  For small dense x the Gram
  // matrix is cheap: power iteration gives the eigenvalue itself.
  if (!X.sparse && (double)X.n * (double)X.p * (double)X.p <= 2e8 && X.p > 1) {
    const size_t p = (size_t)X.p, n = (size_t)X.n;
    std::vector<double> G(p * p, 0.0), v(p, 1.0), u(p);
    for (size_t j = 0; j < p; ++j)
      for (size_t k = 0; k <= j; ++k) {
        const double *a = X.xd.data() + j * n, *b = X.xd.data() + k * n;
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
        G[j * p + k] = G[k * p + j] = s / (double)n;
      }
    double lmax = 0.0;
    for (int it = 0; it < 200; ++it) {
      double nu = 0.0;
      for (size_t j = 0; j < p; ++j) {
        double s = 0.0;
        for (size_t k = 0; k < p; ++k) s += G[j * p + k] * v[k];
        u[j] = s;
        nu += s * s;
      }
      nu = std::sqrt(nu);
      if (!(nu > 0.0)) break;
      const double prev = lmax;
      lmax = nu;                                   // |G v| with |v| = 1
      for (size_t j = 0; j < p; ++j) v[j] = u[j] / nu;
      if (it > 5 && std::fabs(lmax - prev) <= 1e-6 * lmax) break;
      if (it == 0) lmax = 0.0;                     // v was not normalised yet
    }
    diag = std::max(diag, lmax);
  } else if (!X.sparse && X.p > 1) {
    // larger dense x: the same power iteration through X itself, over evenly spaced rows
    // (rows x features <= 2e6 per step)
    const size_t p = (size_t)X.p, n = (size_t)X.n;
    // (2e6 elements per step, at least 1000 rows -- but never more than 16M elements: 1000 rows of 10^6 features
    //  would be 8 GB)
    const size_t m_max = std::max<size_t>(8, std::max<size_t>(std::min<size_t>(1000, (size_t)16000000 / p), (size_t)2000000 / p));
    const size_t stride = (n + m_max - 1) / m_max, m = (n + stride - 1) / stride;
    std::vector<double> v(p, 1.0 / std::sqrt((double)p)), w(p), u(m);
    double lmax = 0.0;
    for (int it = 0; it < 30; ++it) {
      std::fill(u.begin(), u.end(), 0.0);
      for (size_t j = 0; j < p; ++j) {
        const double* col = X.xd.data() + j * n;
        const double vj = v[j];
        for (size_t r = 0; r < m; ++r) u[r] += col[r * stride] * vj;
      }
      double nrm = 0.0;
      for (size_t j = 0; j < p; ++j) {
        const double* col = X.xd.data() + j * n;
        double s = 0.0;
        for (size_t r = 0; r < m; ++r) s += col[r * stride] * u[r];
        w[j] = s / (double)m;
        nrm += w[j] * w[j];
      }
      nrm = std::sqrt(nrm);
      if (!(nrm > 0.0)) break;
      const double prev = lmax;
      lmax = nrm;
      for (size_t j = 0; j < p; ++j) v[j] = w[j] / nrm;
      if (it >= 3 && std::fabs(lmax - prev) <= 2e-3 * lmax) break;
    }
    diag = std::max(diag, lmax);
  }
  if (l_f) *l_f = diag;
  return sgdnet_auto_batch(max_sample_sqnorm, diag);
}

// 2 L_max / L_F without the exported rule's floor of 64 (kWindowFloor instead); *raw = the unclamped value
int64_t window_rule(double max_sample_sqnorm, double l_f, double* raw) {
  *raw = (max_sample_sqnorm > 0.0 && l_f > 0.0) ? 2.0 * max_sample_sqnorm / l_f : 64.0;
  if (!(*raw < 131072.0)) return 131072;
  return *raw < (double)kWindowFloor ? kWindowFloor : (int64_t)*raw;
}

int validate(const sgdnet_control* c, const sgdnet_result* out, int y_cols) {
  if (!c || !out || !out->a0 || !out->beta || !out->lambda || !out->dev_ratio || !out->return_codes) {
    set_error("null control/result pointer");
    return SGDNET_EINVAL;
  }
  if (c->family < SGDNET_GAUSSIAN || c->family > SGDNET_MGAUSSIAN) {
    set_error("unknown family %d", c->family);
    return SGDNET_EINVAL;
  }
  if (c->n_lambda <= 0 || c->n_classes <= 0 || c->max_iter == 0 || c->tol < 0.0 ||
      c->elasticnet_mix < 0.0 || c->elasticnet_mix > 1.0) {
    set_error("invalid control field (n_lambda, n_classes, max_iter, tol or elasticnet_mix)");
    return SGDNET_EINVAL;
  }
  if (c->n_lambda_user > 0 && (c->n_lambda_user != c->n_lambda || !c->lambda)) {
    set_error("control.lambda must hold n_lambda values");
    return SGDNET_EINVAL;
  }
  if (c->family == SGDNET_MGAUSSIAN ? (y_cols != c->n_classes) : (y_cols != 1)) {
    set_error("response has %d columns, family expects %d", y_cols,
              c->family == SGDNET_MGAUSSIAN ? c->n_classes : 1);
    return SGDNET_EINVAL;
  }
  if (c->debug && !c->losses_sink && (!out->losses || !out->losses_len)) {
    set_error("control.debug needs control.losses_sink, or result.losses and result.losses_len");
    return SGDNET_EINVAL;
  }
  return SGDNET_OK;
}

// class codes are used as array indices (FitNullModel, LambdaMax, the gradient kernels)
int validate_response(const sgdnet_control* c, const double* y, int64_t n) {
  if (c->family != SGDNET_BINOMIAL && c->family != SGDNET_MULTINOMIAL) return SGDNET_OK;
  const double top = c->family == SGDNET_BINOMIAL ? 1.0 : (double)(c->n_classes - 1);
  for (int64_t i = 0; i < n; ++i)
    if (!(y[i] >= 0.0 && y[i] <= top && y[i] == floor(y[i]))) {
      set_error("response[%lld] = %g is not a class code in 0..%d", (long long)i, y[i], (int)top);
      return SGDNET_EINVAL;
    }
  return SGDNET_OK;
}

int fit_common(Features& X, const double* y_in, int Ky, const sgdnet_control* ctl, sgdnet_result* out) {
  const int family = ctl->family, K = ctl->n_classes;
  const int64_t n = X.n, p = X.p;
  const int n_lambda = ctl->n_lambda;
  const bool fit_intercept = ctl->intercept != 0;
  const double mix = ctl->elasticnet_mix;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: the SAGA backend has no CPU fallback");
    return SGDNET_ENODEVICE;
  }
  PhaseTimer pt;

  std::vector<double> y(y_in, y_in + n * Ky);
  std::vector<double> yt((size_t)(n * Ky));
  std::vector<double> y_center((size_t)K, 0.0), y_scale((size_t)K, 1.0);

  auto transpose_y = [&]() {
    for (int64_t i = 0; i < n; ++i)
      for (int k = 0; k < Ky; ++k) yt[(size_t)(k + i * Ky)] = y[(size_t)(i + (int64_t)k * n)];
  };

  transpose_y();
  out->nulldev = null_deviance(family, K, yt.data(), Ky, n, fit_intercept);      // sgdnet.cpp:154

  if (family == SGDNET_GAUSSIAN) {                                               // families.h:68-79
    col_mean_sd(y.data(), n, 1, y_center.data(), y_scale.data());
    for (int64_t i = 0; i < n; ++i) y[(size_t)i] = (y[(size_t)i] - y_center[0]) / y_scale[0];
  } else if (family == SGDNET_MGAUSSIAN && ctl->standardize_response) {          // families.h:337-348
    std::vector<double> m((size_t)Ky), s((size_t)Ky);
    col_mean_sd(y.data(), n, Ky, m.data(), s.data());
    standardize_cols(y.data(), n, Ky, m.data(), s.data());
  }

  // RegularizationPath: utils.h:142-181
  std::vector<double> lambda((size_t)n_lambda), alpha((size_t)n_lambda), beta((size_t)n_lambda);
  if (ctl->n_lambda_user == 0) {
    const double lmax = lambda_max(family, K, X, y.data(), Ky, y_scale.data()) / std::max(mix, 0.001);
    if (std::isnan(lmax)) return SGDNET_EHIP;   // device pass failed (sgdnet_last_error says why)
    if (lmax != 0.0) {
      const double log_from = log(lmax);
      const double step = (log(lmax * ctl->lambda_min_ratio) - log_from) / (double)(n_lambda - 1);
      for (int i = 0; i < n_lambda; ++i) lambda[(size_t)i] = exp(log_from + i * step);
    } else {
      std::fill(lambda.begin(), lambda.end(), 0.0);
    }
  } else {
    std::copy(ctl->lambda, ctl->lambda + n_lambda, lambda.begin());
  }
  const double max_scale = *std::max_element(y_scale.begin(), y_scale.end());
  for (int i = 0; i < n_lambda; ++i) {
    alpha[(size_t)i] = (1.0 - mix) * lambda[(size_t)i] / max_scale;
    beta[(size_t)i] = mix * lambda[(size_t)i] / max_scale;
  }

  transpose_y();                                                                 // sgdnet.cpp:178

  // ColNormsMax: utils.h:60-85
  double norm_max = 0.0;
  if (X.dev && X.dense_dev) {
    // L_F for the automatic window from a strided sample of the standardised rows (<= 2e6 elements), while
    // the column-major copy is still there; then transpose + row norms on the device
    if (ctl->mode != SGDNET_MODE_EXACT && ctl->batch <= 0 && p > 1) {
      const int64_t m_max = std::max<int64_t>(8, std::max<int64_t>(std::min<int64_t>(1000, 16000000 / p), 2000000 / p));   // <= 16M elements
      const int64_t stride = (n + m_max - 1) / m_max, m = (n + stride - 1) / stride;
      std::vector<double> xs((size_t)(m * p));
      int rcd = dense_sample_rows(*X.dev, stride, m, xs.data(), X.st);
      if (rcd) return rcd;
      X.dev_max_mean_sq = std::max(X.dev_max_mean_sq, sample_gram_lmax(xs.data(), (size_t)m, (size_t)p));
    }
    int rcd = dense_setup_finish(*X.dev, X.st, &norm_max);
    if (rcd) return rcd;
  } else if (X.dev) {
    // transpose, row norms and record packing on the device; y rides inside the records
    static const int align = exp_env_int("SGDNET_REC_ALIGN", 128);
    int rcd = device_setup_finish(*X.dev, yt.data(), Ky, ctl->standardize ? 1 : 0, align, X.st, &norm_max);
    if (rcd) return rcd;
    if (ctl->mode != SGDNET_MODE_EXACT && ctl->batch <= 0 && option(kOptWindowEigenvalue)) {   // the automatic window needs L_F itself
      double lmax = 0.0;
      rcd = device_gram_lmax(*X.dev, ctl->standardize ? 1 : 0, X.st, &lmax);
      if (rcd) return rcd;
      if (getenv("SGDNET_TRACE"))
        fprintf(stderr, "[sgdnet]   L_F: largest eigenvalue of X'X/n %.4g, its diagonal bound %.4g\n", lmax,
                X.dev_max_mean_sq);
      X.dev_max_mean_sq = std::max(X.dev_max_mean_sq, lmax);
    }
  } else if (X.sparse) {    double csq = 0.0;
    if (ctl->standardize)
      for (int64_t j = 0; j < p; ++j) csq += X.x_center_scaled[(size_t)j] * X.x_center_scaled[(size_t)j];
    for (int64_t i = 0; i < n; ++i) {
      double nrm = 0.0, cnz = 0.0;
      for (int64_t q = X.sptr[(size_t)i]; q < X.sptr[(size_t)i + 1]; ++q) {
        if (ctl->standardize) {
          const double cj = X.x_center_scaled[(size_t)X.sidx[(size_t)q]];
          const double dlt = X.sval[(size_t)q] - cj;
          nrm += dlt * dlt;
          cnz += cj * cj;
        } else {
          nrm += X.sval[(size_t)q] * X.sval[(size_t)q];
        }
      }
      if (ctl->standardize) nrm += csq - cnz;
      norm_max = std::max(norm_max, nrm);
    }
  } else {
    for (int64_t i = 0; i < n; ++i) {
      double nrm = 0.0;
      for (int64_t j = 0; j < p; ++j) nrm += X.xt[(size_t)(j + i * p)] * X.xt[(size_t)(j + i * p)];
      norm_max = std::max(norm_max, nrm);
    }
  }
  const double L_scaling = (family == SGDNET_GAUSSIAN || family == SGDNET_MGAUSSIAN) ? 1.0 : 0.25;

  std::vector<double> b0((size_t)K, 0.0);
  fit_null_model(family, K, yt.data(), Ky, n, fit_intercept, b0.data());          // sgdnet.cpp:210
  const double null_dev_scaled = null_deviance(family, K, yt.data(), Ky, n, fit_intercept);  // :211

  // penalty functor: sgdnet.cpp:80-98
  int penalty = SGDNET_ELASTICNET;
  if (mix == 0.0) penalty = SGDNET_RIDGE;
  else if (family == SGDNET_MGAUSSIAN || (family == SGDNET_MULTINOMIAL && ctl->type_multinomial == 1))
    penalty = SGDNET_GROUPLASSO;

  int mode = ctl->mode;
  int64_t batch = ctl->batch;
  // SGDNET_MODE_BATCHED means "batched where it is implemented": more than 64 classes run the exact
  // iteration instead (a global options(sgdnet.mode = "batched") in R must not make such fits fail);
  // (dense x with 17..64 classes: the class-lane form of round 4; until then sgdnet_fit_dense handed it to the sparse entry point)
  if (mode == SGDNET_MODE_AUTO) mode = SGDNET_MODE_BATCHED;
  if (mode == SGDNET_MODE_BATCHED && K > 64) mode = SGDNET_MODE_EXACT;
  if (mode == SGDNET_MODE_BATCHED) {
    if (batch <= 0) {
      double l_f = X.dev_max_mean_sq, raw = 0.0;
      if (!X.dev) (void)auto_batch(X, norm_max, &l_f);
      // dense x takes the dense intercept step (no 0.01 decay): the constant feature is part of the curvature the
      // stale sum has to respect (its mean square is 1; + 1 bounds the largest eigenvalue of the augmented Gram)
      // (standardised dense features are centred: the constant direction is orthogonal to them and the largest
      //  eigenvalue is max(L_F, 1); otherwise the coupling through the column means is bounded by + 1 after the
      //  step-size normalisation by the largest row)
      if (!X.sparse && fit_intercept) l_f = ctl->standardize ? std::max(l_f, 1.0) : l_f + 1.0;
      batch = window_rule(norm_max, l_f, &raw);
      // ... and so does an epoch of more than kMaxBatchesPerEpoch batches (a short window on many samples): the
      // captured epoch would be a graph of several 10^4 launches, each a few microseconds of fixed cost
      if (ctl->mode == SGDNET_MODE_AUTO && (raw < (double)kWindowFloor || n / batch > kMaxBatchesPerEpoch)) {
        if (getenv("SGDNET_TRACE"))
          fprintf(stderr, "[sgdnet]   mode = auto: window rule gives %.1f draws (%lld batches per epoch): exact iteration\n", raw,
                  (long long)(n / batch));
        mode = SGDNET_MODE_EXACT;
        batch = 0;
      }
    }
  } else if (mode != SGDNET_MODE_EXACT) {
    set_error("unknown mode %d", mode);
    return SGDNET_EINVAL;
  }

  sgdnet_problem pb{};
  pb.family = family;
  pb.n_classes = K;
  pb.n_samples = n;
  pb.n_total = n;
  pb.n_features = p;
  pb.fit_intercept = fit_intercept ? 1 : 0;
  pb.standardize = (X.sparse && ctl->standardize) ? 1 : 0;
  if (X.dev) {
    // matrix (and for sparse x the centring vector and records) are adopted from the device setup
  } else if (X.sparse) {
    pb.rowptr = X.sptr.data();
    pb.colidx = X.sidx.data();
    pb.values = X.sval.data();
    pb.x_center_scaled = pb.standardize ? X.x_center_scaled.data() : nullptr;
  } else {
    pb.x_dense = X.xt.data();
  }
  pb.y = yt.data();
  pb.y_rows = Ky;
  pb.device = ctl->device;

  // ---- the fit sharded over several GPUs of the node (control.n_gpus, ABI 4) ----
  // Rank q holds the samples [q n / N, (q + 1) n / N) on its own GPU with its own virtual shards; the ranks' epoch
  // kernels average ALL replicas among themselves (sgdnet_solver_link_peers).  SS[0] == S leads: every rank holds
  // the same coefficients after an epoch, so the path's decisions are taken from S and applied to all.
  const int NG = ctl->n_gpus > 1 ? ctl->n_gpus : 1;
  std::vector<int> rank_dev((size_t)NG, ctl->device);
  std::vector<int64_t> rank_lo((size_t)NG + 1, 0);
  if (NG > 1) {
    if (NG > 8 || !X.sparse || X.dev || K != 1 || mode != SGDNET_MODE_BATCHED || (p & 1) || ctl->debug ||
        ctl->sample_stream || ctl->unif) {
      set_error("control.n_gpus = %d: a fit is sharded over GPUs in batched mode (mode = batched / auto with a window the "
                "rule accepts), for sparse x with one response and an even number of features, with the built-in "
                "generator and without debug losses (at most 8 GPUs)", NG);
      return SGDNET_EUNSUPPORTED;
    }
    for (int q = 0; q < NG; ++q) {
      rank_dev[(size_t)q] = ctl->devices ? ctl->devices[q] : ctl->device + q;
      if (rank_dev[(size_t)q] < 0 || rank_dev[(size_t)q] >= ndev) {
        set_error("control.n_gpus = %d: device %d out of range (%d devices)", NG, rank_dev[(size_t)q], ndev);
        return SGDNET_EINVAL;
      }
      rank_lo[(size_t)q + 1] = n / NG * (q + 1) + std::min<int64_t>(q + 1, n % NG);     // sgdnet_amd/parallel.py: shard_bounds
    }
  }
  rank_lo[(size_t)NG] = n;

  sgdnet_solver* S = nullptr;
  std::vector<sgdnet_solver*> SS;
  pt.mark("response, path, step sizes");
  struct Guard {
    std::vector<sgdnet_solver*>* ss;
    ~Guard() {
      for (sgdnet_solver* q : *ss) sgdnet_solver_destroy(q);
    }
  } guard{&SS};
  int rc = SGDNET_OK;
  if (NG == 1) {
    rc = X.dev ? solver_create_adopting(&pb, *X.dev, &S) : sgdnet_solver_create(&pb, &S);
    if (rc) return rc;
    SS.push_back(S);
  } else {
    std::vector<int64_t> ptr_q;
    for (int q = 0; q < NG && !rc; ++q) {
      const int64_t lo = rank_lo[(size_t)q], hi = rank_lo[(size_t)q + 1];
      sgdnet_problem pq = pb;
      pq.n_samples = hi - lo;
      pq.n_total = hi - lo;                       // local normalisation: a rank's shards average their own samples
      ptr_q.assign(X.sptr.begin() + lo, X.sptr.begin() + hi + 1);
      const int64_t off = ptr_q[0];
      for (int64_t& v : ptr_q) v -= off;
      pq.rowptr = ptr_q.data();
      pq.colidx = X.sidx.data() + off;
      pq.values = X.sval.data() + off;
      pq.y = yt.data() + lo * Ky;
      pq.device = rank_dev[(size_t)q];
      sgdnet_solver* sq = nullptr;
      rc = sgdnet_solver_create(&pq, &sq);
      if (!rc) SS.push_back(sq);
    }
    if (rc) return rc;
    S = SS[0];
  }
  pt.mark("solver create (pack + H2D)");
  auto for_all = [&](auto f) -> int {
    for (sgdnet_solver* q : SS) {
      const int r = f(q);
      if (r) return r;
    }
    return SGDNET_OK;
  };

  rc = for_all([&](sgdnet_solver* q) { return sgdnet_solver_set_state(q, 1, b0.data()); });
  if (rc) return rc;
  if (mode == SGDNET_MODE_BATCHED && K > 16 && !solver_batched_available(S, batch)) {
    // 17..64 classes have one batched form, the binned one, and it needs feature ranges (at most 2048 of them)
    if (ctl->mode == SGDNET_MODE_BATCHED && getenv("SGDNET_TRACE"))
      fprintf(stderr, "[sgdnet]   %d classes on %lld features: no batched form, exact iteration\n", K, (long long)p);
    mode = SGDNET_MODE_EXACT;
    batch = 0;
  }

  DrawSource draws(ctl);
  int vshards = 0;
  // Virtual shards (include/sgdnet_hip.h): with enough samples per feature the batched fit of
  // one response (round 3: or of 2..16 classes of sparse x; round 4: of dense x too) runs as up to 8 locally normalised replicas over sample ranges, averaged on the
  // device every n / 32 draws -- same optimum, same epochs to tolerance, 2x the epochs per second
  // at the benchmark shapes.  sgdnet_set_option("virtual_shards", 0) switches it off, V forces V.
  // The shard kernels read a per-shard layout of the sample order: the built-in generator and
  // the unif callback produce it (DrawSource::fill), an explicit sample_stream cannot.
  if (mode == SGDNET_MODE_BATCHED && K <= 16 && draws.shardable()) {
    int V = 1;
    // at least 100 samples per feature in every shard, and a problem large enough for the
    // per-launch cost to matter (small correlated data, e.g. abalone 4177 x 9, converges slower
    // or not at all when its replicas are averaged)
    if (n >= 200000)
      while (V < 8 && (int64_t)(2 * V) * 100 * p <= n) V *= 2;
    // dense x with several classes (round 4): two replicas.  Its windows are a few hundred draws, an epoch is launch-bound
    // and V shards make it V times shorter, but on such well-conditioned data the averaged replicas need more epochs
    // (125 / 213 / 417 at V = 1 / 2 / 4 on 1M x 100, K = 4; profiles/r04_dense_multiclass_vshards.txt): 2 is what pays
    if (!X.sparse && K > 1 && V > 2) V = 2;
    if (option(kOptVirtualShards) >= 0) V = option(kOptVirtualShards);
    if (NG > 1) {
      // the job stays a V-way average (8 at most: what the averaging tolerates at these sizes), cut over
      // the ranks -- at least two shards per rank (what the epoch kernel carries)
      V = std::max(2, std::min(8, V) / NG);
      std::vector<int> on_dev((size_t)ndev, 0);
      for (int q = 0; q < NG; ++q) ++on_dev[(size_t)rank_dev[(size_t)q]];
      for (int q = 0; q < NG && !rc; ++q) {
        if (on_dev[(size_t)rank_dev[(size_t)q]] > 1)          // ranks that share a GPU (rehearsals) share its CUs
          rc = sgdnet_solver_set_cu_budget(SS[(size_t)q], 256 / on_dev[(size_t)rank_dev[(size_t)q]]);
        if (!rc) rc = sgdnet_solver_set_virtual_shards(SS[(size_t)q], V);
        // a quarter of a shard's own epoch between two averages, the same draw count on every rank
        if (!rc) rc = sgdnet_solver_set_merge_period(SS[(size_t)q], std::max<int64_t>(1, (n / NG / V) / 4));
      }
      if (!rc) rc = sgdnet_solver_link_peers(SS.data(), NG);
      if (rc) return rc;
      vshards = V;
    } else if (V >= 2 && V <= 8) {
      rc = sgdnet_solver_set_virtual_shards(S, V);
      if (rc && rc != SGDNET_EUNSUPPORTED) return rc;
      if (!rc) vshards = V;
    }
  }
  if (NG > 1 && vshards < 2) {
    set_error("control.n_gpus = %d: the sample order cannot be laid out per shard (explicit sample_stream?)", NG);
    return SGDNET_EUNSUPPORTED;
  }
  if (vshards > 1 && ctl->batch <= 0 && batch > 0) {
    // at most 1/8 beyond the rule's window when that saves the short last round of every shard's epoch
    // (include/sgdnet_hip.h: sgdnet_shard_window; the rule keeps a factor 3 to the unstable regime, profiles/NOTES.md)
    batch = sgdnet_shard_window(batch, (n / NG) / vshards);
  }

  // built-in generator: the draws are produced in HBM (r_rng_device.hip), one epoch ahead of
  // the epoch that consumes them, on a side stream (solver.cpp: solver_rng_*)
  // Exact mode with the built-in generator: the draws come in BLOCKS of several epochs (generated on
  // the device, ~1M draws at a time) and one launch runs as many epochs as the block still holds,
  // with the convergence test in the kernel -- a small problem (iris: 150 draws per epoch) is then no
  // longer one launch + one host round trip per epoch.  The stream is consumed contiguously across
  // epochs and lambdas, so the generator ends exactly where the reference's would: the final state is
  // the block's start state stepped by the draws that were used.
  const bool exact_blocks = draws.internal() && mode == SGDNET_MODE_EXACT && !ctl->debug && option(kOptExactEpochBlocks);
  const bool pipe = draws.internal() && !exact_blocks;
  struct {
    sgdnet_rng start;
    int64_t cap = 0, used = 0;
    bool have = false;
  } blk;
  const int64_t blk_epochs = std::max<int64_t>(1, std::min<int64_t>(64, (1 << 20) / std::max<int64_t>(1, n)));
  struct PipeGuard {
    std::vector<sgdnet_solver*>* ss;
    sgdnet_rng* rng;
    bool on;
    ~PipeGuard() {
      if (!on) return;
      sgdnet_rng scratch;
      for (size_t q = 0; q < ss->size(); ++q) (void)solver_rng_close((*ss)[q], q == 0 ? rng : &scratch);
    }
  } pipe_guard{&SS, &draws.rng, false};
  if (pipe) {
    // batched mode with epochs of 200 000 draws or more: 8-32 generators side by side ON THE ONE
    // R STREAM (one makes 10M draws in 5.3 ms, six epochs of the batched kernels at C4): generator g
    // starts g * ceil(n / G) draws into the epoch and all of them jump n draws per epoch
    // (mt_jump.cpp), so the sample order is set.seed()'s whatever G is.  Smaller problems and exact
    // mode keep a single generator.  SGDNET_RNG_GENERATORS overrides.
    int gens = 1;
    if (mode == SGDNET_MODE_BATCHED) {
      const int forced = option(kOptRngGenerators);
      // a generator's workgroup cannot share a CU with a gather workgroup (LDS and registers are
      // taken), so a long-running generator costs every overlapping gather launch a second round:
      // C4 epochs 1.07 / 0.93 / 0.86 ms with 8 / 16 / 32 generators (0.85 with the stream resident)
      gens = forced > 0 ? forced : (n >= 200000 ? (int)std::min<int64_t>(32, std::max<int64_t>(8, n / 300000)) : 1);
    }
    if (NG == 1) {
      rc = solver_rng_open(S, &draws.rng, n, gens);
      if (rc) return rc;
      pipe_guard.on = true;
      rc = solver_rng_prefetch(S);
      if (rc) return rc;
    } else {
      // ONE R stream over all ranks: an epoch is n consecutive draws of set.seed()'s generator, rank q takes the
      // n_q of them that start lo_q draws in -- its generators start there (the caller's state jumped lo_q draws,
      // mt_jump.cpp) and move n draws per epoch like everybody's.  Rank 0's state after the fit is R's after
      // epochs * n draws: what one GPU returns.
      pipe_guard.on = true;
      for (int q = 0; q < NG; ++q) {
        sgdnet_rng start = draws.rng;
        if (rank_lo[(size_t)q] > 0) {
          std::vector<uint32_t> poly(624);
          if (!mt_jump_poly((uint64_t)rank_lo[(size_t)q], poly.data())) {
            set_error("control.n_gpus: the jump polynomial of the generator could not be formed");
            return SGDNET_EHIP;
          }
          mt_jump_host(&draws.rng, poly.data(), &start);
        }
        const int64_t nq = rank_lo[(size_t)q + 1] - rank_lo[(size_t)q];
        const int gq = std::max(2, gens / NG);
        rc = solver_rng_open(SS[(size_t)q], &start, nq, gq, n);
        if (!rc) rc = solver_rng_prefetch(SS[(size_t)q]);
        if (rc) return rc;
      }
    }
  }
  std::vector<uint32_t> chunk((size_t)n);
  std::vector<double> w((size_t)(K * p)), b((size_t)K), xbb((size_t)K);
  std::vector<double> losses;        // debug: grows with the epochs run, like the reference's vector (saga-sparse.h:364)
  double n_iter = 0.0;
  int64_t auto_window = batch;      // shrinks for good when a run blew up or a fit got worse
  double prev_dev = HUGE_VAL;
  int retries = 0;
  double t_rng = 0.0, t_run = 0.0, t_chk = 0.0, t_dev = 0.0;   // SGDNET_TRACE: where the path's time goes
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(now() - t0).count();
  };

  for (int li = 0; li < n_lambda; ++li) {                                        // sgdnet.cpp:217-273
    // StepSize: utils.h:31-51
    const double L = (norm_max + (fit_intercept ? 1.0 : 0.0)) * L_scaling + alpha[(size_t)li];
    const double mu_n = 2.0 * (double)n * alpha[(size_t)li];
    const double gamma = 1.0 / (2.0 * L + std::min(L, mu_n));
    rc = for_all([&](sgdnet_solver* q) { return sgdnet_solver_set_penalty(q, penalty, gamma, alpha[(size_t)li], beta[(size_t)li]); });
    if (rc) return rc;

    unsigned epochs = 0;
    int converged = 0;
    if (mode == SGDNET_MODE_BATCHED && ctl->batch <= 0) batch = auto_window;   // a new lambda: a new step size
    if (li == 0 && getenv("SGDNET_TRACE")) fprintf(stderr, "[sgdnet]   window %lld draws\n", (long long)batch);
    double best_ratio = HUGE_VAL;
    int worse = 0;
    // one epoch per launch: exactly the draws the reference would consume are taken
    // from the source (R's RNG state after the call matches, SURVEY.md 8b "RNG")
    while (epochs < ctl->max_iter && !converged) {
      int64_t stream_off = 0;
      auto t0 = now();
      unsigned want_epochs = 1;
      const bool multi = pipe && NG > 1;
      if (multi) {
        // every rank's epoch is enqueued before any of them is waited for: the ranks' kernels wait for each other
        int64_t offs[8] = {0};
        for (int q = 0; q < NG && !rc; ++q) {
          rc = solver_rng_prefetch(SS[(size_t)q]);
          if (!rc) rc = solver_rng_acquire(SS[(size_t)q], &offs[q]);
        }
        draws.pos += n;
        t_rng += since(t0);
        t0 = now();
        for (int q = 0; q < NG && !rc; ++q)
          rc = sgdnet_solver_enqueue_epochs(SS[(size_t)q], batch, offs[q], rank_lo[(size_t)q + 1] - rank_lo[(size_t)q], 1);
        for (int q = 0; q < NG && !rc; ++q) rc = solver_rng_release(SS[(size_t)q]);
        // ConvergenceCheck on every rank: the same coefficients everywhere, each rank keeps its own w_prev
        for (int q = 0; q < NG && !rc; ++q) {
          int cq = 0;
          rc = sgdnet_solver_convergence(SS[(size_t)q], ctl->tol, &cq);
          if (q == 0) converged = cq;
          if (!rc && solver_fused_aborted(SS[(size_t)q])) {
            set_error("control.n_gpus = %d: rank %d's epoch kernel could not run (its GPU is shared with other work, or "
                      "the ranks' kernels did not get to run side by side)", NG, q);
            rc = SGDNET_EHIP;
          }
        }
        if (rc) return rc;
        t_run += since(t0);
        t0 = now();
        epochs += 1;
      } else {
      if (exact_blocks) {
        if (!blk.have || blk.cap - blk.used < n) {
          blk.start = draws.rng;
          blk.cap = blk_epochs * n;
          blk.used = 0;
          blk.have = true;
          rc = sgdnet_solver_generate_stream(S, &draws.rng, blk.cap);   // draws.rng <- state after the block
          if (rc) return rc;
        }
        stream_off = blk.used;
        want_epochs = (unsigned)std::min<int64_t>((blk.cap - blk.used) / n, (int64_t)(ctl->max_iter - epochs));
      } else if (pipe) {
        rc = solver_rng_prefetch(S);               // next epoch's draws, concurrently
        if (rc) return rc;
        rc = solver_rng_acquire(S, &stream_off);   // this epoch's
        draws.pos += n;
      } else {
        rc = draws.fill((uint32_t)n, chunk.data(), n, vshards > 1 ? vshards : 1);
        if (rc) return rc;
        rc = sgdnet_solver_upload_stream(S, chunk.data(), n);
      }
      if (rc) return rc;
      t_rng += since(t0);
      t0 = now();
      unsigned ran = 0;
      if (ctl->debug && losses.size() < (size_t)epochs + 1) losses.resize(std::max<size_t>(64, 2 * losses.size()));
      rc = sgdnet_solver_run(S, mode, batch, stream_off, n, want_epochs, ctl->tol, &ran, &converged,
                             ctl->debug ? losses.data() + epochs : nullptr);
      if (rc == SGDNET_EUNSUPPORTED && solver_bin_overflowed(S)) {
        // binned form: a feature range got more entries in one batch than its bin holds, the epoch is void.
        // More room (or, in the end, the atomic form) and this lambda again from the null model; with more
        // than 16 classes and no room left there is no batched form: mode = auto then fits in exact mode.
        if (pipe) {
          int rcr = solver_rng_release(S);
          if (rcr) return rcr;
        }
        rc = solver_grow_bins(S);
        if (rc) {
          t_batched_diverged = true;
          return rc;
        }
        if (getenv("SGDNET_TRACE")) fprintf(stderr, "[sgdnet]   lambda %d: a bin overflowed -> more room, again\n", li);
        // (a cold restart: the warm start of the previous lambda goes too, since the void epoch has been applied to
        //  it; the lambda gets its full max_iter again -- the draws of the void epochs stay consumed, as R's generator
        //  would have it, and are counted in draws_used)
        rc = solver_reset_state(S, b0.data());
        if (rc) return rc;
        epochs = 0;
        worse = 0;
        best_ratio = HUGE_VAL;
        converged = 0;
        continue;
      }
      if (rc) return rc;
      if (exact_blocks) {
        blk.used += (int64_t)ran * n;
        draws.pos += (int64_t)ran * n;
      }
      if (pipe) {
        rc = solver_rng_release(S);
        if (rc) return rc;
      }
      t_run += since(t0);
      t0 = now();
      epochs += ran;
      }   // one rank
      if (mode == SGDNET_MODE_BATCHED) {
        // guard of the automatic window: the stale-sum step is only stable below ~L_max/L_F
        // draws, and the bound used by sgdnet_auto_batch is optimistic for correlated
        // features.  A change ratio that keeps growing (or stops being finite) halves it.
        double ch = 0.0, sz = 0.0;
        sgdnet_solver_last_change(S, &ch, &sz);
        const double ratio = sz > 0.0 ? ch / sz : 0.0;
        // the soft threshold maps a NaN coefficient to 0, so a blown-up run can look converged
        // (max|w| = 0): the intercept keeps the evidence
        bool finite = std::isfinite(ratio) && std::isfinite(sz);
        if (finite && fit_intercept) {
          rc = sgdnet_solver_get_state(S, 1, b.data());
          if (rc) return rc;
          for (int k = 0; k < K; ++k) finite = finite && std::isfinite(b[(size_t)k]);
        }
        if (!finite && NG > 1) {
          set_error("control.n_gpus = %d: the batched iteration diverged (non-finite coefficients); fit on one GPU, or pass a "
                    "smaller control.batch", NG);
          return SGDNET_EUNSUPPORTED;
        }
        if (!finite) {
          // restart this lambda from the null model: first without virtual shards (their
          // averaging assumes shards that look alike), then with a quarter of the window
          if (ctl->batch > 0 && vshards <= 1) {
            set_error("batched mode diverged (non-finite coefficients); pass a smaller control.batch");
            return SGDNET_EUNSUPPORTED;
          }
          if (vshards > 1) {
            vshards = 0;
            rc = sgdnet_solver_set_virtual_shards(S, 0);
          } else if (batch > kRetryWindowMin) {
            // the rule's own floor is 64 draws; a fit that blows up there (few, strongly scaled dense
            // features) gets a shorter window before batched mode is given up
            batch = std::max<int64_t>(kRetryWindowMin, batch / 4);
            auto_window = batch;
            if (getenv("SGDNET_TRACE"))
              fprintf(stderr, "[sgdnet]   lambda %d: non-finite coefficients -> window %lld, again\n", li, (long long)batch);
          } else {
            set_error("batched mode diverged (non-finite coefficients) at the smallest window; use mode = exact");
            t_batched_diverged = true;
            return SGDNET_EUNSUPPORTED;
          }
          if (!rc) rc = solver_reset_state(S, b0.data());
          if (rc) return rc;
          worse = 0;
          best_ratio = HUGE_VAL;
          converged = 0;
          continue;
        }
        // a run on its way out changes the coefficients by a growing, LARGE fraction of their size
        // per epoch; near convergence the ratio is noise around the tolerance and means nothing
        // (without the second condition a 100-lambda path halved its way down to 64 draws)
        // ... and at lambda_max, where the solution is exactly 0, max|w| is rounding noise and the
        // ratio means nothing either (a C3 path spent 54 epochs there halving down to 78 draws)
        const bool at_lambda_max = li == 0 && ctl->n_lambda_user == 0;   // solution exactly 0: no signal
        if (ratio > 4.0 * best_ratio && ratio > 0.05 && sz > 1e-9 && epochs > 2 && !at_lambda_max) ++worse;
        else worse = 0;
        if (ratio > 0.0) best_ratio = std::min(best_ratio, ratio);
        if (worse >= 2 && ctl->batch <= 0 && batch > kWindowFloor) {
          if (getenv("SGDNET_TRACE"))
            fprintf(stderr, "[sgdnet]   lambda %d epoch %u: change ratio %.3g after best %.3g -> window %lld halved\n", li,
                    epochs, ratio, best_ratio, (long long)batch);
          batch = std::max<int64_t>(kWindowFloor, batch / 2);
          // what made the window too long (correlated features) does not depend on lambda: keep
          // it -- except at lambda_max, where a handful of coefficients flicker around zero
          if (li > 0) auto_window = batch;
          worse = 0;
          best_ratio = ratio;
        }
      }
      t_chk += since(t0);
    }
    auto t1 = now();
    out->return_codes[li] = (epochs == ctl->max_iter) ? 1.0 : 0.0;               // saga-sparse.h:376-382
    auto report_losses = [&]() {
      if (!ctl->debug) return;
      if (ctl->losses_sink) ctl->losses_sink(ctl->losses_ctx, li, losses.data(), (int)epochs);
      if (out->losses && out->losses_len) {
        memcpy(out->losses + (size_t)li * ctl->max_iter, losses.data(), sizeof(double) * epochs);
        out->losses_len[li] = (int32_t)epochs;
      }
    };

    double dev = 0.0;
    rc = for_all([&](sgdnet_solver* q) {                                         // sgdnet.cpp:246-256 (every rank: its samples)
      double dq = 0.0;
      const int r = sgdnet_solver_deviance(q, &dq);
      dev += dq;
      return r;
    });
    if (rc) return rc;
    // Safety net of the automatic window: along a decreasing lambda path the deviance of the
    // training data can only fall.  A window that is too long for the data does not have to blow
    // up -- it can settle into a bounded oscillation that the change-ratio guard never sees and
    // that returns a useless fit (deviance above the null model's).  Then: a quarter of the window
    // for the rest of the path, and this lambda again from the null model.
    // ... and whatever the order of a user-supplied lambda sequence: a fit whose deviance is above the
    // null model's (w = 0, intercept only -- the point every lambda can reach) is not a fit.
    const bool worse_than_previous = li > 0 && lambda[(size_t)li] < lambda[(size_t)li - 1] && dev > prev_dev * (1.0 + 1e-3);
    // (a null deviance of exactly 0 -- a constant response -- leaves nothing to compare with: every dev > 0 would
    //  burn the whole ladder)
    const bool worse_than_null = (null_dev_scaled > 0.0 && dev > null_dev_scaled * (1.0 + 1e-3)) || !std::isfinite(dev);
    if (mode == SGDNET_MODE_BATCHED && ctl->batch <= 0 && (worse_than_previous || worse_than_null)) {
      if (batch <= kWindowFloor || retries >= 8) {
        // the ladder is exhausted and the fit is still worse than a point every lambda can reach: not a fit.
        // mode = auto reruns the whole fit with the exact iteration (sgdnet_fit_*), explicit batched reports it
        set_error("batched mode: the fit at lambda[%d] is worse than %s (deviance %.6g) after %d restarts down to a window of "
                  "%lld draws; use mode = exact", li, worse_than_null ? "the null model" : "the previous lambda's", dev, retries,
                  (long long)batch);
        t_batched_diverged = true;
        return SGDNET_EUNSUPPORTED;
      }
      if (getenv("SGDNET_TRACE"))
        fprintf(stderr, "[sgdnet]   lambda %d: deviance %.6g (previous lambda %.6g, null model %.6g) -> window %lld / 4, again\n",
                li, dev, prev_dev, null_dev_scaled, (long long)batch);
      if (vshards > 1 && NG == 1) {
        vshards = 0;
        rc = sgdnet_solver_set_virtual_shards(S, 0);
        if (rc) return rc;
      }
      auto_window = std::max<int64_t>(kWindowFloor, batch / 4);
      rc = for_all([&](sgdnet_solver* q) { return solver_reset_state(q, b0.data()); });
      if (rc) return rc;
      ++retries;
      --li;
      continue;
    }
    retries = 0;
    prev_dev = dev;
    n_iter += (double)epochs;          // epochs of the accepted run of this lambda only
    report_losses();
    out->dev_ratio[li] = 1.0 - dev / null_dev_scaled;                            // :258
    out->lambda[li] = lambda[(size_t)li];

    // Rescale: utils.h:352-378
    rc = sgdnet_solver_get_state(S, 0, w.data());
    if (rc) return rc;
    rc = sgdnet_solver_get_state(S, 1, b.data());
    if (rc) return rc;
    double* bo = out->beta + (size_t)li * (size_t)(K * p);
    double* ao = out->a0 + (size_t)li * (size_t)K;
    std::fill(xbb.begin(), xbb.end(), 0.0);
    for (int64_t j = 0; j < p; ++j)
      for (int k = 0; k < K; ++k) {
        const double v = w[(size_t)(k + j * K)] * (y_scale[(size_t)k] / X.x_scale[(size_t)j]);
        bo[k + j * K] = v;
        xbb[(size_t)k] += X.x_center[(size_t)j] * v;
      }
    for (int k = 0; k < K; ++k)
      ao[k] = fit_intercept ? b[(size_t)k] * y_scale[(size_t)k] + y_center[(size_t)k] - xbb[(size_t)k]
                            : b[(size_t)k];
    t_dev += since(t1);
  }
  if (getenv("SGDNET_TRACE"))
    fprintf(stderr, "[sgdnet]   of which: sample order %.3f s, epochs %.3f s, per-epoch checks %.3f s, per-lambda deviance + rescale %.3f s\n",
            t_rng, t_run, t_chk, t_dev);
  pt.mark("lambda path (SAGA + deviance)");
  out->npasses = n_iter;
  out->draws_used = draws.pos;
  if (pipe) {
    pipe_guard.on = false;
    for (size_t q = 0; q < SS.size(); ++q) {
      sgdnet_rng other;
      rc = solver_rng_close(SS[q], q == 0 ? &draws.rng : &other);   // (rank 0:) state after exactly the epochs that ran
      if (rc) return rc;
    }
  }
  if (exact_blocks && blk.have) {                    // state after exactly the draws that were used
    draws.rng = blk.start;
    std::vector<uint32_t> scratch((size_t)std::max<int64_t>(1, blk.used));
    if (blk.used > 0) sgdnet_rng_fill(&draws.rng, (uint32_t)n, scratch.data(), blk.used);
  }
  draws.finish();
  return SGDNET_OK;
}

}  // namespace

// mode = auto promises a fit: when the batched iteration gives up (non-finite coefficients even at the
// shortest window) the whole fit is run again in exact mode
template <class F>
static int with_exact_fallback(const sgdnet_control* ctl, F fit) {
  t_batched_diverged = false;
  int rc = fit(ctl);
  if (rc == SGDNET_EUNSUPPORTED && t_batched_diverged && ctl && ctl->mode == SGDNET_MODE_AUTO) {
    if (getenv("SGDNET_TRACE")) fprintf(stderr, "[sgdnet] mode = auto: batched iteration gave up, fitting again in exact mode\n");
    sgdnet_control exact = *ctl;
    exact.mode = SGDNET_MODE_EXACT;
    exact.batch = 0;
    t_batched_diverged = false;
    rc = fit(&exact);
  }
  return rc;
}

extern "C" {

static int fit_sparse_impl(const sgdnet_csc* x, const double* y, int y_cols, const sgdnet_control* ctl,
                           sgdnet_result* out);
static int fit_dense_impl(const double* x, int64_t n, int64_t p, const double* y, int y_cols,
                          const sgdnet_control* ctl, sgdnet_result* out);

int sgdnet_fit_sparse(const sgdnet_csc* x, const double* y, int y_cols, const sgdnet_control* ctl,
                      sgdnet_result* out) {
  return with_exact_fallback(ctl, [&](const sgdnet_control* c) { return fit_sparse_impl(x, y, y_cols, c, out); });
}

int sgdnet_fit_dense(const double* x, int64_t n, int64_t p, const double* y, int y_cols,
                     const sgdnet_control* ctl, sgdnet_result* out) {
  return with_exact_fallback(ctl, [&](const sgdnet_control* c) { return fit_dense_impl(x, n, p, y, y_cols, c, out); });
}

static int fit_sparse_impl(const sgdnet_csc* x, const double* y, int y_cols, const sgdnet_control* ctl,
                           sgdnet_result* out) {
  int rc = validate(ctl, out, y_cols);
  if (rc) return rc;
  if (!x || !y || x->n_rows <= 0 || x->n_cols <= 0 || !x->colptr || !x->rowidx || !x->values) {
    set_error("sgdnet_fit_sparse: invalid matrix");
    return SGDNET_EINVAL;
  }
  Features X;
  X.sparse = true;
  X.n = x->n_rows;
  X.p = x->n_cols;
  X.colptr = x->colptr;
  X.rowidx = x->rowidx;
  const int64_t n = X.n, p = X.p, nnz = x->colptr[p];
  if (x->colptr[0] != 0) {
    set_error("colptr[0] must be 0");
    return SGDNET_EINVAL;
  }
  for (int64_t j = 0; j < p; ++j)
    if (x->colptr[j + 1] < x->colptr[j]) {
      set_error("colptr is not non-decreasing at column %lld", (long long)j);
      return SGDNET_EINVAL;
    }
  rc = validate_response(ctl, y, n);
  if (rc) return rc;
  for (int64_t q = 0; q < nnz; ++q) {
    const int32_t r = x->rowidx[q];
    if (r < 0 || r >= n) {
      set_error("row index %d out of range at position %lld", r, (long long)q);
      return SGDNET_EINVAL;
    }
  }
  if (!option(kOptHostSetup) && !(ctl->n_gpus > 1)) {   // (a fit sharded over several GPUs cuts the host copy into the ranks' ranges)
    // default: the per-fit O(nnz) passes run on the device (setup_device.hip)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      set_error("no HIP device available: the SAGA backend has no CPU fallback");
      return SGDNET_ENODEVICE;
    }
    if (ctl->device < 0 || ctl->device >= ndev) {
      set_error("device %d out of range (%d devices)", ctl->device, ndev);
      return SGDNET_EINVAL;
    }
    SGD_HIP_TRY(hipSetDevice(ctl->device));
    DeviceSetup dev;
    hipStream_t st = nullptr;
    SGD_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    X.dev = &dev;
    X.st = st;
    rc = device_setup_begin(dev, x, ctl->standardize ? 1 : 0, st, X.x_center, X.x_scale, &X.dev_max_mean_sq);
    if (!rc) rc = fit_common(X, y, y_cols, ctl, out);
    dev.release();
    (void)hipStreamDestroy(st);
    return rc;
  }
  X.val.assign(x->values, x->values + nnz);
  X.x_center.assign((size_t)p, 0.0);
  X.x_scale.assign((size_t)p, 1.0);
  X.x_center_scaled.assign((size_t)p, 0.0);
  if (ctl->standardize) {                                     // utils.h:110-121, math.h:66-79,89-112
    for (int64_t j = 0; j < p; ++j) {
      const int64_t q0 = x->colptr[j], q1 = x->colptr[j + 1];
      double s = 0.0;
      for (int64_t q = q0; q < q1; ++q) s += X.val[(size_t)q];
      const double mean = s / (double)n;
      double var = 0.0;
      for (int64_t q = q0; q < q1; ++q) var += pow(X.val[(size_t)q] - mean, 2) / (double)n;
      var += (double)(n - (q1 - q0)) * mean * mean / (double)n;
      const double sd = (var == 0.0) ? 1.0 : sqrt(var);
      for (int64_t q = q0; q < q1; ++q) X.val[(size_t)q] /= sd;
      X.x_center[(size_t)j] = mean;
      X.x_scale[(size_t)j] = sd;
      X.x_center_scaled[(size_t)j] = mean / sd;             // sgdnet.cpp:150
    }
  }
  // AdaptiveTranspose (utils.h:276-281): counting sort into sample-major order
  X.sptr.assign((size_t)n + 1, 0);
  for (int64_t q = 0; q < nnz; ++q) X.sptr[(size_t)x->rowidx[q] + 1]++;
  for (int64_t i = 0; i < n; ++i) X.sptr[(size_t)i + 1] += X.sptr[(size_t)i];
  X.sidx.resize((size_t)nnz);
  X.sval.resize((size_t)nnz);
  {
    std::vector<int64_t> fill(X.sptr.begin(), X.sptr.end() - 1);
    for (int64_t j = 0; j < p; ++j)
      for (int64_t q = x->colptr[j]; q < x->colptr[j + 1]; ++q) {
        const int64_t dst = fill[(size_t)x->rowidx[q]]++;
        X.sidx[(size_t)dst] = (int32_t)j;
        X.sval[(size_t)dst] = X.val[(size_t)q];
      }
  }
  if (getenv("SGDNET_TRACE")) fprintf(stderr, "[sgdnet] features: standardize + transpose done\n");
  return fit_common(X, y, y_cols, ctl, out);
}

static int fit_dense_impl(const double* x, int64_t n, int64_t p, const double* y, int y_cols,
                          const sgdnet_control* ctl, sgdnet_result* out) {
  int rc = validate(ctl, out, y_cols);
  if (rc) return rc;
  if (!x || !y || n <= 0 || p <= 0) {
    set_error("sgdnet_fit_dense: invalid matrix");
    return SGDNET_EINVAL;
  }
  rc = validate_response(ctl, y, n);
  if (rc) return rc;
  Features X;
  X.sparse = false;
  X.n = n;
  X.p = p;
  if (n * p >= kDenseDeviceSetupElems && !option(kOptHostSetup)) {
    // large dense x: statistics, standardisation, lambda_max products, transpose and row norms on the
    // device (dense_setup_*), no host pass over the n * p doubles beyond the one upload
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      set_error("no HIP device available: the SAGA backend has no CPU fallback");
      return SGDNET_ENODEVICE;
    }
    if (ctl->device < 0 || ctl->device >= ndev) {
      set_error("device %d out of range (%d devices)", ctl->device, ndev);
      return SGDNET_EINVAL;
    }
    SGD_HIP_TRY(hipSetDevice(ctl->device));
    DeviceSetup dev;
    hipStream_t st = nullptr;
    SGD_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    X.dev = &dev;
    X.st = st;
    X.dense_dev = true;
    X.x_center_scaled.assign((size_t)p, 0.0);
    rc = dense_setup_begin(dev, x, n, p, ctl->standardize ? 1 : 0, st, X.x_center, X.x_scale, &X.dev_max_mean_sq);
    if (!rc) rc = fit_common(X, y, y_cols, ctl, out);
    dev.release();
    (void)hipStreamDestroy(st);
    return rc;
  }
  X.xd.assign(x, x + n * p);
  X.x_center.assign((size_t)p, 0.0);
  X.x_scale.assign((size_t)p, 1.0);
  X.x_center_scaled.assign((size_t)p, 0.0);                   // sgdnet.cpp:151
  if (ctl->standardize) {                                     // utils.h:99-108
    col_mean_sd(X.xd.data(), n, p, X.x_center.data(), X.x_scale.data());
    standardize_cols(X.xd.data(), n, p, X.x_center.data(), X.x_scale.data());
  }
  X.xt.resize((size_t)(n * p));                               // utils.h:283-288
  transpose_to_sample_major(X.xd.data(), n, p, X.xt.data());
  return fit_common(X, y, y_cols, ctl, out);
}

}  // extern "C"
