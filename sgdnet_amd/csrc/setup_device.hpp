// Device-resident setup state of one sparse fit (setup_device.hip).
#pragma once

#include <vector>

#include "common.hpp"

namespace sgdnet {

struct DeviceSetup {
  int64_t n = 0, p = 0, nnz = 0;
  // feature-major copy of x (freed by device_setup_finish)
  int32_t* colptr = nullptr;
  int32_t* rowidx = nullptr;
  double* val = nullptr;
  // sample-major matrix, centring vector and packed records: adopted by the solver
  int64_t* sptr = nullptr;
  int32_t* sidx = nullptr;
  double* sval = nullptr;
  double* center_scaled = nullptr;
  char* rec = nullptr;
  char* ovf = nullptr;
  int rec_stride = 0, rec_cap = 0, rec_val_off = 0;
  float avg_nnz = 0.f;
  // dense x (large problems only, dense_setup_*): column-major copy (freed after the transpose) and
  // the sample-major p x n matrix the solver adopts
  double* xd_cm = nullptr;
  double* xd_t = nullptr;
  void release();
};

// Dense x on the device: column statistics + standardisation (utils.h:99-108), lambda_max products,
// the transpose (utils.h:283-288) and ColNormsMax without the host passes over n*p doubles.
// Used above kDenseDeviceSetupElems elements; smaller matrices (the reference's own data sets) keep
// the host loops, whose sums run in the reference's order.
constexpr int64_t kDenseDeviceSetupElems = 4000000;
int dense_setup_begin(DeviceSetup& S, const double* x_host, int64_t n, int64_t p, int standardize, hipStream_t st,
                      std::vector<double>& x_center, std::vector<double>& x_scale, double* max_mean_sq);
int dense_xt_times(const DeviceSetup& S, const double* ymap_host, int cols, double* xty_host, hipStream_t st);
int dense_setup_finish(DeviceSetup& S, hipStream_t st, double* max_sqnorm);
// rows r * stride (r < m) of the standardised matrix, m x p column-major, for the host power iteration
int dense_sample_rows(const DeviceSetup& S, int64_t stride, int64_t m, double* out_host, hipStream_t st);

int device_setup_begin(DeviceSetup& S, const sgdnet_csc* x, int standardize, hipStream_t st,
                       std::vector<double>& x_center, std::vector<double>& x_scale, double* max_mean_sq);
int device_xt_times(const DeviceSetup& S, const double* ymap_host, int cols, double* xty_host, hipStream_t st);
int device_gram_lmax(const DeviceSetup& S, int standardize, hipStream_t st, double* lmax);
int device_setup_finish(DeviceSetup& S, const double* y_host, int y_rows, int standardize, int rec_align,
                        hipStream_t st, double* max_sqnorm);

// solver.cpp: builds a solver around buffers that device_setup_* left on the device
// (ownership moves to the solver).  pb carries the scalar fields and the HOST response.
int solver_create_adopting(const sgdnet_problem* pb, DeviceSetup& S, sgdnet_solver** out);

}  // namespace sgdnet

// solver.cpp: sample-order pipeline of the fit driver (the next epoch's draws are generated on a
// side stream while the current epoch runs)
int solver_rng_open(sgdnet_solver* s, sgdnet_rng* rng, int64_t n, int generators = 1, int64_t jump_draws = 0);
int solver_reset_state(sgdnet_solver* s, const double* b0);
bool solver_bin_overflowed(const sgdnet_solver* s);
bool solver_fused_aborted(const sgdnet_solver* s);   // the last ConvergenceCheck found a fused epoch launch that gave up
int solver_grow_bins(sgdnet_solver* s);
bool solver_batched_available(sgdnet_solver* s, int64_t batch);
int solver_rng_prefetch(sgdnet_solver* s);
int solver_rng_acquire(sgdnet_solver* s, int64_t* offset);
int solver_rng_release(sgdnet_solver* s);
int solver_rng_close(sgdnet_solver* s, sgdnet_rng* rng);
