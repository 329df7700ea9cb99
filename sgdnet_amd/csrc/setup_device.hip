// Per-fit setup passes on the device for sparse x.
//
// What SetupSgdnet does once per fit before the lambda loop (reference src/sgdnet.cpp:143-184),
// on the feature-major matrix R passes in, without bringing the O(nnz) work back to the host:
//   col_stats_kernel     Mean / StandardDeviation / PreprocessFeatures   math.h:66-112, utils.h:110-121
//   xt_times_kernel      x^T * y_map for Family::LambdaMax               families.h:119-126,203-220,300-325,387-406
//   transpose            AdaptiveTranspose (feature-major -> sample-major) utils.h:276-281
//                        as a STABLE radix sort of the entries by sample id, so feature ids stay
//                        ascending inside a sample (the order the exact kernel's dot product uses)
//   row_norm_kernel      ColNormsMax                                      utils.h:60-77
//   pack_records_kernel  the batched gather's packed records (saga_batched.hip)
// All of it is streaming / segmented-reduction work bound by HBM bandwidth; none of it is on
// the per-epoch path.
#include <cmath>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "device_math.hpp"
#include "setup_device.hpp"

namespace sgdnet {

namespace {

constexpr int kTB = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < kTB / 64; ++i) t += red[i];
  __syncthreads();
  return t;
}

// one block per feature column: mean, population sd (0 -> 1), in-place scaling
__global__ __launch_bounds__(kTB) void col_stats_kernel(const int32_t* colptr, double* val, int64_t n,
                                                        int standardize, double* center, double* scale,
                                                        double* center_scaled, double* mean_sq) {
  __shared__ double red[kTB / 64];
  const int64_t j = blockIdx.x;
  const int64_t q0 = colptr[j], q1 = colptr[j + 1];
  double mean = 0.0, sd = 1.0;
  if (standardize) {
    double s = 0.0;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += kTB) s += val[q];
    mean = block_sum(s, red) / (double)n;
    double v = 0.0;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += kTB) {
      const double dlt = val[q] - mean;
      v += dlt * dlt / (double)n;                       // math.h:103
    }
    double var = block_sum(v, red);
    var += (double)(n - (q1 - q0)) * mean * mean / (double)n;   // implicit zeros, math.h:105-106
    sd = var == 0.0 ? 1.0 : sqrt(var);
    for (int64_t q = q0 + threadIdx.x; q < q1; q += kTB) val[q] /= sd;   // utils.h:118-120
    __syncthreads();
  }
  double sq = 0.0;
  for (int64_t q = q0 + threadIdx.x; q < q1; q += kTB) sq += val[q] * val[q];
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    center[j] = mean;
    scale[j] = sd;
    center_scaled[j] = mean / sd;                       // sgdnet.cpp:150
    mean_sq[j] = sq / (double)n;                        // diagonal of X'X/n (auto batch)
  }
}

// out[j + c*p] = sum_q val[q] * ymap[rowidx[q] + c*n]
__global__ __launch_bounds__(kTB) void xt_times_kernel(const int32_t* colptr, const int32_t* rowidx,
                                                       const double* val, const double* ymap, int64_t n,
                                                       int64_t p, int cols, double* out) {
  __shared__ double red[kTB / 64];
  const int64_t j = blockIdx.x;
  const int64_t q0 = colptr[j], q1 = colptr[j + 1];
  for (int c = 0; c < cols; ++c) {
    const double* yc = ymap + (int64_t)c * n;
    double s = 0.0;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += kTB) s += val[q] * yc[rowidx[q]];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[j + (int64_t)c * p] = s;
  }
}

// colof[q] = j for q in column j; entry[q] = q
__global__ __launch_bounds__(kTB) void expand_cols_kernel(const int32_t* colptr, int32_t* colof,
                                                          int32_t* entry) {
  const int64_t j = blockIdx.x;
  for (int64_t q = colptr[j] + threadIdx.x; q < colptr[j + 1]; q += kTB) {
    colof[q] = (int32_t)j;
    entry[q] = (int32_t)q;
  }
}

__global__ __launch_bounds__(kTB) void count_rows_kernel(const int32_t* rowidx, int64_t nnz,
                                                         unsigned long long* counts) {
  for (int64_t q = (int64_t)blockIdx.x * kTB + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * kTB)
    atomicAdd(counts + rowidx[q], 1ull);
}

__global__ __launch_bounds__(kTB) void gather_sorted_kernel(const int32_t* perm, const int32_t* colof,
                                                            const double* val, int64_t nnz, int32_t* sidx,
                                                            double* sval) {
  for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kTB) {
    const int32_t q = perm[i];
    sidx[i] = colof[q];
    sval[i] = val[q];
  }
}

// per sample: squared norm (centred when standardize) -> max; row length histogram (65 bins)
__global__ __launch_bounds__(kTB) void row_norm_kernel(const int64_t* sptr, const int32_t* sidx,
                                                       const double* sval, const double* c, double csq,
                                                       int64_t n, unsigned long long* max_bits,
                                                       unsigned long long* hist, unsigned long long* zmax) {
  double best = 0.0;
  unsigned long long zm = 0;
  for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTB) {
    const int64_t q0 = sptr[i], q1 = sptr[i + 1];
    double nrm = 0.0, cnz = 0.0;
    for (int64_t q = q0; q < q1; ++q) {
      if (c) {
        const double cj = c[sidx[q]];
        const double dlt = sval[q] - cj;
        nrm += dlt * dlt;
        cnz += cj * cj;
      } else {
        nrm += sval[q] * sval[q];
      }
    }
    if (c) nrm += csq - cnz;          // ||x_i - c||^2 = sum_nz (x-c)^2 + sum_{not nz} c^2
    best = fmax(best, nrm);
    const unsigned long long z = (unsigned long long)(q1 - q0);
    zm = z > zm ? z : zm;
    atomicAdd(hist + (z < 64 ? z : 64), 1ull);
  }
  best = wave_max(best);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(max_bits, (unsigned long long)__double_as_longlong(best));
    atomicMax(zmax, zm);
  }
}

__global__ __launch_bounds__(kTB) void ovf_count_kernel(const int64_t* sptr, int64_t n, int cap, int ovf_cap,
                                                        long long* cnt) {
  for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTB) {
    const int64_t z = sptr[i + 1] - sptr[i];
    cnt[i] = z > cap ? (z - cap + ovf_cap - 1) / ovf_cap : 0;
  }
}

__global__ __launch_bounds__(kTB) void pack_records_kernel(const int64_t* sptr, const int32_t* sidx,
                                                           const double* sval, const double* y, int y_in_rec,
                                                           const long long* ovf_off, int64_t n, int stride,
                                                           int cap, int val_off, int ovf_stride, int ovf_cap,
                                                           char* rec, char* ovf) {
  for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTB) {
    char* base = rec + (size_t)i * stride;
    const int64_t q0 = sptr[i];
    const int nnz = (int)(sptr[i + 1] - q0);
    *reinterpret_cast<double*>(base) = y_in_rec ? y[i] : 0.0;
    *reinterpret_cast<int*>(base + 8) = nnz;
    const int c0 = nnz < cap ? nnz : cap;
    int* ridx = reinterpret_cast<int*>(base + 16);
    double* rval = reinterpret_cast<double*>(base + val_off);
    for (int e = 0; e < cap; ++e) {
      ridx[e] = e < c0 ? sidx[q0 + e] : 0;
      rval[e] = e < c0 ? sval[q0 + e] : 0.0;
    }
    int done = c0;
    long long id = ovf_off[i];
    *reinterpret_cast<int*>(base + 12) = (int)id;
    while (done < nnz) {
      const int c = (nnz - done) < ovf_cap ? (nnz - done) : ovf_cap;
      char* ob = ovf + (size_t)id * ovf_stride;
      *reinterpret_cast<int*>(ob) = (int)(id + 1);
      *reinterpret_cast<int*>(ob + 4) = c;
      int* oi = reinterpret_cast<int*>(ob + 8);
      double* ov = reinterpret_cast<double*>(ob + 8 + 4 * ovf_cap);
      for (int e = 0; e < ovf_cap; ++e) {
        oi[e] = e < c ? sidx[q0 + done + e] : 0;
        ov[e] = e < c ? sval[q0 + done + e] : 0.0;
      }
      done += c;
      ++id;
    }
  }
}

template <typename T>
int dmalloc(T** p, size_t count) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (count ? count : 1));
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", sizeof(T) * count, hipGetErrorString(e));
    return SGDNET_ENOMEM;
  }
  return SGDNET_OK;
}

int grid_for(int64_t items) {
  int64_t g = (items + kTB - 1) / kTB;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (int)g;
}

}  // namespace

void DeviceSetup::release() {
  for (void* p : {(void*)colptr, (void*)rowidx, (void*)val, (void*)sptr, (void*)sidx, (void*)sval,
                  (void*)center_scaled, (void*)rec, (void*)ovf, (void*)xd_cm, (void*)xd_t})
    if (p) (void)hipFree(p);
  xd_cm = xd_t = nullptr;
  colptr = rowidx = nullptr;
  val = sval = center_scaled = nullptr;
  sptr = nullptr;
  sidx = nullptr;
  rec = ovf = nullptr;
}

// Upload the dgCMatrix slots and run column statistics (+ scaling when standardize).
int device_setup_begin(DeviceSetup& S, const sgdnet_csc* x, int standardize, hipStream_t st,
                       std::vector<double>& x_center, std::vector<double>& x_scale, double* max_mean_sq) {
  S.n = x->n_rows;
  S.p = x->n_cols;
  S.nnz = x->colptr[S.p];
  int rc;
  if ((rc = dmalloc(&S.colptr, (size_t)S.p + 1)) || (rc = dmalloc(&S.rowidx, (size_t)S.nnz)) ||
      (rc = dmalloc(&S.val, (size_t)S.nnz)) || (rc = dmalloc(&S.center_scaled, (size_t)S.p)))
    return rc;
  SGD_HIP_TRY(hipMemcpyAsync(S.colptr, x->colptr, sizeof(int32_t) * ((size_t)S.p + 1), hipMemcpyHostToDevice, st));
  SGD_HIP_TRY(hipMemcpyAsync(S.rowidx, x->rowidx, sizeof(int32_t) * (size_t)S.nnz, hipMemcpyHostToDevice, st));
  SGD_HIP_TRY(hipMemcpyAsync(S.val, x->values, sizeof(double) * (size_t)S.nnz, hipMemcpyHostToDevice, st));
  double *center = nullptr, *scale = nullptr, *msq = nullptr;
  if ((rc = dmalloc(&center, (size_t)S.p)) || (rc = dmalloc(&scale, (size_t)S.p)) || (rc = dmalloc(&msq, (size_t)S.p)))
    return rc;
  hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)S.p), dim3(kTB), 0, st, S.colptr, S.val, S.n, standardize,
                     center, scale, S.center_scaled, msq);
  SGD_HIP_TRY(hipGetLastError());
  x_center.resize((size_t)S.p);
  x_scale.resize((size_t)S.p);
  std::vector<double> m((size_t)S.p);
  SGD_HIP_TRY(hipMemcpyAsync(x_center.data(), center, sizeof(double) * (size_t)S.p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipMemcpyAsync(x_scale.data(), scale, sizeof(double) * (size_t)S.p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipMemcpyAsync(m.data(), msq, sizeof(double) * (size_t)S.p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  double best = 0.0;
  for (double v : m) best = v > best ? v : best;
  *max_mean_sq = best;
  (void)hipFree(center);
  (void)hipFree(scale);
  (void)hipFree(msq);
  return SGDNET_OK;
}

// xty (p x cols, host) = x^T ymap; ymap is n x cols on the host.
int device_xt_times(const DeviceSetup& S, const double* ymap_host, int cols, double* xty_host, hipStream_t st) {
  double *ymap = nullptr, *out = nullptr;
  int rc;
  if ((rc = dmalloc(&ymap, (size_t)S.n * cols)) || (rc = dmalloc(&out, (size_t)S.p * cols))) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(ymap, ymap_host, sizeof(double) * (size_t)S.n * cols, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(xt_times_kernel, dim3((unsigned)S.p), dim3(kTB), 0, st, S.colptr, S.rowidx, S.val, ymap, S.n,
                     S.p, cols, out);
  SGD_HIP_TRY(hipGetLastError());
  SGD_HIP_TRY(hipMemcpyAsync(xty_host, out, sizeof(double) * (size_t)S.p * cols, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(ymap);
  (void)hipFree(out);
  return SGDNET_OK;
}

// out += sum over the sampled rows i of x_i (x_i . v): one application of X'X restricted to every
// stride-th sample (thread per row; rows are short)
__global__ __launch_bounds__(kTB) void gram_apply_kernel(const int64_t* sptr, const int32_t* sidx,
                                                        const double* sval, int64_t n, int64_t stride,
                                                        const double* v, double* out) {
  for (int64_t r = (int64_t)blockIdx.x * kTB + threadIdx.x; r * stride < n; r += (int64_t)gridDim.x * kTB) {
    const int64_t i = r * stride;
    const int64_t q0 = sptr[i], q1 = sptr[i + 1];
    double u = 0.0;
    for (int64_t q = q0; q < q1; ++q) u += sval[q] * v[sidx[q]];
    if (u != 0.0)
      for (int64_t q = q0; q < q1; ++q) atomicAdd(out + sidx[q], sval[q] * u);
  }
}

// Largest eigenvalue of X'X / n (of the centred features when standardize) by power iteration
// over at most ~2M evenly spaced samples.  The batched mode's window is 2 L_max / L_F and L_F is
// this eigenvalue; its diagonal lower bound is only good for features without a common
// component -- non-negative sparse data (counts, tf-idf, uniform(0,1) values) has one, worth
// (p - 1) * density^2 * mean^2 on top of the diagonal, and a window 8x too long then oscillates
// into a useless fit at small lambda without ever producing a non-finite number.
int device_gram_lmax(const DeviceSetup& S, int standardize, hipStream_t st, double* lmax) {
  const int64_t n = S.n, p = S.p;
  const int64_t stride = (n + 2000000 - 1) / 2000000;
  const int64_t m = (n + stride - 1) / stride;
  double *v_dev = nullptr, *out_dev = nullptr;
  int rc;
  if ((rc = dmalloc(&v_dev, (size_t)p)) || (rc = dmalloc(&out_dev, (size_t)p))) return rc;
  std::vector<double> v((size_t)p, 1.0 / std::sqrt((double)p)), w((size_t)p), c;
  if (standardize) {
    c.resize((size_t)p);
    SGD_HIP_TRY(hipMemcpy(c.data(), S.center_scaled, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost));
  }
  double lam = 0.0;
  for (int it = 0; it < 30; ++it) {
    SGD_HIP_TRY(hipMemcpyAsync(v_dev, v.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, st));
    SGD_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(double) * (size_t)p, st));
    hipLaunchKernelGGL(gram_apply_kernel, dim3(grid_for(m)), dim3(kTB), 0, st, S.sptr, S.sidx, S.sval, n, stride,
                       v_dev, out_dev);
    SGD_HIP_TRY(hipGetLastError());
    SGD_HIP_TRY(hipMemcpyAsync(w.data(), out_dev, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
    SGD_HIP_TRY(hipStreamSynchronize(st));
    double cv = 0.0;
    if (standardize)
      for (int64_t j = 0; j < p; ++j) cv += c[(size_t)j] * v[(size_t)j];
    double nrm = 0.0;
    for (int64_t j = 0; j < p; ++j) {
      w[(size_t)j] = w[(size_t)j] / (double)m - (standardize ? c[(size_t)j] * cv : 0.0);
      nrm += w[(size_t)j] * w[(size_t)j];
    }
    nrm = std::sqrt(nrm);
    if (!(nrm > 0.0)) break;
    const double prev = lam;
    lam = nrm;                                     // |A v| with |v| = 1: a lower bound that grows to lambda_max
    for (int64_t j = 0; j < p; ++j) v[(size_t)j] = w[(size_t)j] / nrm;
    if (it >= 3 && std::fabs(lam - prev) <= 2e-3 * lam) break;
  }
  (void)hipFree(v_dev);
  (void)hipFree(out_dev);
  *lmax = lam;
  return SGDNET_OK;
}

// Feature-major -> sample-major, row norms, packed records.  Frees the feature-major copy.
int device_setup_finish(DeviceSetup& S, const double* y_host, int y_rows, int standardize, int rec_align,
                        hipStream_t st, double* max_sqnorm) {
  const int64_t n = S.n, p = S.p, nnz = S.nnz;
  int rc;
  // ---- transpose: stable sort of (row, entry) pairs by row ----
  int32_t *colof = nullptr, *entry = nullptr, *rows_sorted = nullptr, *perm = nullptr;
  unsigned long long* counts = nullptr;
  if ((rc = dmalloc(&colof, (size_t)nnz)) || (rc = dmalloc(&entry, (size_t)nnz)) ||
      (rc = dmalloc(&rows_sorted, (size_t)nnz)) || (rc = dmalloc(&perm, (size_t)nnz)) ||
      (rc = dmalloc(&counts, (size_t)n + 1)) || (rc = dmalloc(&S.sptr, (size_t)n + 1)) ||
      (rc = dmalloc(&S.sidx, (size_t)nnz)) || (rc = dmalloc(&S.sval, (size_t)nnz)))
    return rc;
  hipLaunchKernelGGL(expand_cols_kernel, dim3((unsigned)p), dim3(kTB), 0, st, S.colptr, colof, entry);
  SGD_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * ((size_t)n + 1), st));
  hipLaunchKernelGGL(count_rows_kernel, dim3(grid_for(nnz)), dim3(kTB), 0, st, S.rowidx, nnz, counts);
  SGD_HIP_TRY(hipGetLastError());
  int end_bit = 1;
  while (end_bit < 32 && (1ll << end_bit) < n) ++end_bit;
  size_t tmp_bytes = 0, scan_bytes = 0;
  SGD_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, S.rowidx, rows_sorted, entry, perm, (int)nnz, 0,
                                                 end_bit, st));
  SGD_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, reinterpret_cast<long long*>(counts),
                                               reinterpret_cast<long long*>(S.sptr), (int)(n + 1), st));
  void* tmp = nullptr;
  const size_t tb = tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes;
  SGD_HIP_TRY(hipMalloc(&tmp, tb ? tb : 1));
  SGD_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, S.rowidx, rows_sorted, entry, perm, (int)nnz, 0,
                                                 end_bit, st));
  SGD_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, reinterpret_cast<long long*>(counts),
                                               reinterpret_cast<long long*>(S.sptr), (int)(n + 1), st));
  hipLaunchKernelGGL(gather_sorted_kernel, dim3(grid_for(nnz)), dim3(kTB), 0, st, perm, colof, S.val, nnz, S.sidx,
                     S.sval);
  SGD_HIP_TRY(hipGetLastError());
  SGD_HIP_TRY(hipStreamSynchronize(st));
  for (void* q : {(void*)colof, (void*)entry, (void*)rows_sorted, (void*)perm, (void*)S.colptr, (void*)S.rowidx,
                  (void*)S.val})
    (void)hipFree(q);
  S.colptr = S.rowidx = nullptr;
  S.val = nullptr;

  // ---- ColNormsMax + row length statistics ----
  double csq = 0.0;
  if (standardize) {
    std::vector<double> c((size_t)p);
    SGD_HIP_TRY(hipMemcpy(c.data(), S.center_scaled, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost));
    for (double v : c) csq += v * v;
  }
  unsigned long long* stats = nullptr;   // [0] max bits, [1] zmax, [2..66] histogram
  if ((rc = dmalloc(&stats, 67))) return rc;
  SGD_HIP_TRY(hipMemsetAsync(stats, 0, sizeof(unsigned long long) * 67, st));
  hipLaunchKernelGGL(row_norm_kernel, dim3(grid_for(n)), dim3(kTB), 0, st, S.sptr, S.sidx, S.sval,
                     standardize ? S.center_scaled : nullptr, csq, n, stats, stats + 2, stats + 1);
  SGD_HIP_TRY(hipGetLastError());
  unsigned long long hs[67];
  SGD_HIP_TRY(hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  memcpy(max_sqnorm, &hs[0], 8);
  S.avg_nnz = (float)((double)nnz / (double)n);

  // ---- record geometry: 90th percentile row, 128-B-aligned stride (solver.cpp: build_records) ----
  int cap = 1;
  {
    unsigned long long acc = 0;
    const unsigned long long want = (unsigned long long)(0.9 * (double)n);
    for (int z = 0; z <= 64; ++z) {
      acc += hs[2 + z];
      cap = z < 1 ? 1 : z;
      if (acc >= want) break;
    }
    if (cap >= 64) cap = (int)(hs[1] < 512 ? hs[1] : 512);
  }
  auto rec_bytes = [](int c) { return 16 + ((4 * c + 7) & ~7) + 8 * c; };
  const int stride = (rec_bytes(cap) + rec_align - 1) / rec_align * rec_align;
  while (rec_bytes(cap + 1) <= stride) ++cap;
  S.rec_stride = stride;
  S.rec_cap = cap;
  S.rec_val_off = 16 + ((4 * cap + 7) & ~7);
  constexpr int kOvfStride = 256, kOvfCap = 20;
  long long *ocnt = nullptr, *ooff = nullptr;
  if ((rc = dmalloc(&ocnt, (size_t)n + 1)) || (rc = dmalloc(&ooff, (size_t)n + 1))) return rc;
  SGD_HIP_TRY(hipMemsetAsync(ocnt, 0, sizeof(long long) * ((size_t)n + 1), st));
  hipLaunchKernelGGL(ovf_count_kernel, dim3(grid_for(n)), dim3(kTB), 0, st, S.sptr, n, cap, kOvfCap, ocnt);
  SGD_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, ocnt, ooff, (int)(n + 1), st));
  long long n_ovf = 0;
  SGD_HIP_TRY(hipMemcpyAsync(&n_ovf, ooff + n, sizeof(long long), hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  if ((rc = dmalloc(&S.rec, (size_t)n * stride)) || (rc = dmalloc(&S.ovf, (size_t)(n_ovf ? n_ovf : 1) * kOvfStride)))
    return rc;
  double* y_dev = nullptr;
  if (y_rows == 1) {
    if ((rc = dmalloc(&y_dev, (size_t)n))) return rc;
    SGD_HIP_TRY(hipMemcpyAsync(y_dev, y_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(pack_records_kernel, dim3(grid_for(n)), dim3(kTB), 0, st, S.sptr, S.sidx, S.sval, y_dev,
                     y_rows == 1 ? 1 : 0, ooff, n, stride, cap, S.rec_val_off, kOvfStride, kOvfCap, S.rec, S.ovf);
  SGD_HIP_TRY(hipGetLastError());
  SGD_HIP_TRY(hipStreamSynchronize(st));
  for (void* q : {(void*)tmp, (void*)counts, (void*)stats, (void*)ocnt, (void*)ooff, (void*)y_dev})
    if (q) (void)hipFree(q);
  return SGDNET_OK;
}


// ---------------------------------------------------------------------------------------------
// Dense x (SURVEY.md 8 row f1 for SgdnetDense): the same once-per-fit passes for a column-major
// n x p matrix.  All streaming: a block per column for the statistics and products, LDS tiles for
// the transpose.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTB) void dense_col_stats_kernel(double* x, int64_t n, int standardize, double* center,
                                                              double* scale, double* mean_sq) {
  __shared__ double red[kTB / 64];
  double* col = x + (int64_t)blockIdx.x * n;
  double mean = 0.0, sd = 1.0;
  if (standardize) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kTB) s += col[i];
    mean = block_sum(s, red) / (double)n;                          // math.h:66-79
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kTB) {
      const double dlt = col[i] - mean;
      v += dlt * dlt;
    }
    const double var = block_sum(v, red) / (double)n;              // math.h:114-130 (population sd, 0 -> 1)
    sd = var == 0.0 ? 1.0 : sqrt(var);
    for (int64_t i = threadIdx.x; i < n; i += kTB) col[i] = (col[i] - mean) / sd;   // utils.h:99-108
    __syncthreads();
  }
  double sq = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += kTB) sq += col[i] * col[i];
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    center[blockIdx.x] = mean;
    scale[blockIdx.x] = sd;
    mean_sq[blockIdx.x] = sq / (double)n;
  }
}

__global__ __launch_bounds__(kTB) void dense_xt_times_kernel(const double* x, const double* ymap, int64_t n, int64_t p,
                                                             int cols, double* out) {
  __shared__ double red[kTB / 64];
  const double* col = x + (int64_t)blockIdx.x * n;
  for (int c = 0; c < cols; ++c) {
    const double* yc = ymap + (int64_t)c * n;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kTB) s += col[i] * yc[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x + (int64_t)c * p] = s;
  }
}

// xt[j + i*p] = x[i + j*n]: 32 x 32 tiles through LDS, both sides coalesced
// (tiles in a one-dimensional grid: grid.y stops at 65 535 = 2.09M features or samples)
__global__ __launch_bounds__(256) void dense_transpose_kernel(const double* x, int64_t n, int64_t p, double* xt) {
  __shared__ double tile[32][33];
  const int64_t tiles_i = (n + 31) / 32;
  const int64_t tj = (int64_t)blockIdx.x / tiles_i, ti = (int64_t)blockIdx.x - tj * tiles_i;
  const int64_t i0 = ti * 32, j0 = tj * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = i0 + tx, j = j0 + r;
    tile[r][tx] = (i < n && j < p) ? x[i + j * n] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = i0 + r, j = j0 + tx;
    if (i < n && j < p) xt[j + i * p] = tile[tx][r];
  }
}

// max over samples of |x_i|^2 on the sample-major matrix: a wavefront per sample
__global__ __launch_bounds__(kTB) void dense_row_norm_kernel(const double* xt, int64_t n, int64_t p,
                                                             unsigned long long* max_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kTB + threadIdx.x) >> 6, nwaves = (int64_t)gridDim.x * (kTB / 64);
  double best = 0.0;
  for (int64_t i = wave; i < n; i += nwaves) {
    const double* row = xt + i * p;
    double s = 0.0;
    for (int64_t j = lane; j < p; j += 64) s += row[j] * row[j];
    s = wave_sum(s);
    best = s > best ? s : best;
  }
  if (lane == 0 && best > 0.0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(best));
}

__global__ __launch_bounds__(kTB) void dense_sample_rows_kernel(const double* x, int64_t n, int64_t p, int64_t stride,
                                                                int64_t m, double* out) {
  for (int64_t t = (int64_t)blockIdx.x * kTB + threadIdx.x; t < m * p; t += (int64_t)gridDim.x * kTB) {
    const int64_t r = t % m, j = t / m;
    out[t] = x[r * stride + j * n];
  }
}

int dense_setup_begin(DeviceSetup& S, const double* x_host, int64_t n, int64_t p, int standardize, hipStream_t st,
                      std::vector<double>& x_center, std::vector<double>& x_scale, double* max_mean_sq) {
  S.n = n;
  S.p = p;
  int rc;
  if ((rc = dmalloc(&S.xd_cm, (size_t)n * (size_t)p))) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(S.xd_cm, x_host, sizeof(double) * (size_t)n * (size_t)p, hipMemcpyHostToDevice, st));
  double *center = nullptr, *scale = nullptr, *msq = nullptr;
  if ((rc = dmalloc(&center, (size_t)p)) || (rc = dmalloc(&scale, (size_t)p)) || (rc = dmalloc(&msq, (size_t)p))) return rc;
  hipLaunchKernelGGL(dense_col_stats_kernel, dim3((unsigned)p), dim3(kTB), 0, st, S.xd_cm, n, standardize, center, scale,
                     msq);
  SGD_HIP_TRY(hipGetLastError());
  x_center.resize((size_t)p);
  x_scale.resize((size_t)p);
  std::vector<double> m((size_t)p);
  SGD_HIP_TRY(hipMemcpyAsync(x_center.data(), center, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipMemcpyAsync(x_scale.data(), scale, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipMemcpyAsync(m.data(), msq, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  double best = 0.0;
  for (double v : m) best = v > best ? v : best;
  *max_mean_sq = best;
  (void)hipFree(center);
  (void)hipFree(scale);
  (void)hipFree(msq);
  return SGDNET_OK;
}

int dense_xt_times(const DeviceSetup& S, const double* ymap_host, int cols, double* xty_host, hipStream_t st) {
  double *ymap = nullptr, *out = nullptr;
  int rc;
  if ((rc = dmalloc(&ymap, (size_t)S.n * cols)) || (rc = dmalloc(&out, (size_t)S.p * cols))) return rc;
  SGD_HIP_TRY(hipMemcpyAsync(ymap, ymap_host, sizeof(double) * (size_t)S.n * cols, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(dense_xt_times_kernel, dim3((unsigned)S.p), dim3(kTB), 0, st, S.xd_cm, ymap, S.n, S.p, cols, out);
  SGD_HIP_TRY(hipGetLastError());
  SGD_HIP_TRY(hipMemcpyAsync(xty_host, out, sizeof(double) * (size_t)S.p * cols, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(ymap);
  (void)hipFree(out);
  return SGDNET_OK;
}

int dense_sample_rows(const DeviceSetup& S, int64_t stride, int64_t m, double* out_host, hipStream_t st) {
  double* out = nullptr;
  int rc;
  if ((rc = dmalloc(&out, (size_t)m * (size_t)S.p))) return rc;
  int64_t grid = (m * S.p + kTB - 1) / kTB;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(dense_sample_rows_kernel, dim3((unsigned)grid), dim3(kTB), 0, st, S.xd_cm, S.n, S.p, stride, m, out);
  SGD_HIP_TRY(hipGetLastError());
  SGD_HIP_TRY(hipMemcpyAsync(out_host, out, sizeof(double) * (size_t)m * (size_t)S.p, hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(out);
  return SGDNET_OK;
}

// transpose to sample-major, row norms; the column-major copy is released
int dense_setup_finish(DeviceSetup& S, hipStream_t st, double* max_sqnorm) {
  int rc;
  if ((rc = dmalloc(&S.xd_t, (size_t)S.n * (size_t)S.p))) return rc;
  const int64_t tiles = ((S.n + 31) / 32) * ((S.p + 31) / 32);
  if (tiles > 0x7fffffffll) {
    set_error("dense x of %lld x %lld is beyond the device setup's transpose", (long long)S.n, (long long)S.p);
    return SGDNET_EUNSUPPORTED;
  }
  hipLaunchKernelGGL(dense_transpose_kernel, dim3((unsigned)tiles), dim3(256), 0, st, S.xd_cm, S.n, S.p, S.xd_t);
  SGD_HIP_TRY(hipGetLastError());
  unsigned long long* mb = nullptr;
  if ((rc = dmalloc(&mb, 1))) return rc;
  SGD_HIP_TRY(hipMemsetAsync(mb, 0, sizeof(unsigned long long), st));
  int64_t grid = (S.n * 64 + kTB - 1) / kTB;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(dense_row_norm_kernel, dim3((unsigned)grid), dim3(kTB), 0, st, S.xd_t, S.n, S.p, mb);
  SGD_HIP_TRY(hipGetLastError());
  unsigned long long bits = 0;
  SGD_HIP_TRY(hipMemcpyAsync(&bits, mb, sizeof(bits), hipMemcpyDeviceToHost, st));
  SGD_HIP_TRY(hipStreamSynchronize(st));
  memcpy(max_sqnorm, &bits, sizeof(double));
  (void)hipFree(mb);
  (void)hipFree(S.xd_cm);
  S.xd_cm = nullptr;
  return SGDNET_OK;
}

}  // namespace sgdnet
