// R-compatible Mersenne-Twister on the device: the sample order of a fit without a
// round trip through the host.
//
// The reference draws floor(R::runif(0, n)) once per inner iteration
// (src/saga-sparse.h:261, src/saga-dense.h:152).  At 10M draws per epoch the host
// generator (r_rng.cpp, ~50 ms per epoch plus a 40 MB upload) would bound a batched fit whose
// epoch takes ~3 ms, so the same stream is produced in HBM.  MT19937's recurrence
//   x[k+624] = x[k+397] ^ twist(x[k], x[k+1])
// has dependency distance 227 (= 624 - 397): a block of 624 words is regenerated in three
// phases of 227 / 227 / 170 independent words (double-buffered in LDS), then tempered, scaled
// exactly like unif_rand() and floored.  One wavefront carries the sequence (it is inherently
// serial across blocks); the other CUs keep running SAGA.
#include "common.hpp"
#include "r_rng_word.hpp"
#include "r_rng_bodies.hpp"

namespace sgdnet {


__global__ __launch_bounds__(kRngBlock* kGenPerWg) void r_mt_state_kernel(const uint32_t* st_in, uint32_t* st_out,
                                                                          uint32_t* out, int64_t count, int64_t seg,
                                                                          int gens) {
  __shared__ uint32_t bufs[kGenPerWg][2][kMtN + 1];
  mt_state_body((int)blockIdx.x, (int)gridDim.x, bufs, st_in, st_out, out, count, seg, gens);
}

__global__ __launch_bounds__(kJumpBlock) void r_mt_jump_kernel(const uint32_t* st_in_all, uint32_t* st_out_all,
                                                               const uint32_t* poly, int gens) {
  extern __shared__ uint32_t xs[];
  mt_jump_body((int)blockIdx.x, (int)gridDim.x, xs, st_in_all, st_out_all, poly, gens);
}

// gens generators, at most `max_wgs` workgroups (each takes its generators in turn)
int launch_rng_jump(const uint32_t* state_in, uint32_t* state_out, const uint32_t* poly_dev, int gens,
                    hipStream_t st, int max_wgs) {
  static bool attr_done_dev[64] = {};
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (!attr_done_dev[cur & 63]) {
    SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(r_mt_jump_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kJumpLds));
    attr_done_dev[cur & 63] = true;
  }
  int grid = gens;
  if (max_wgs > 0 && grid > max_wgs) grid = max_wgs;
  hipLaunchKernelGGL(r_mt_jump_kernel, dim3(grid), dim3(kJumpBlock), kJumpLds, st, state_in, state_out, poly_dev, gens);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int rng_generators_per_workgroup() { return kGenPerWg; }

// shards.V > 1 (virtual shards, common.hpp): the positions are laid out run by run (a run = what the shards do
// between two merges across GPUs; one run = the whole epoch on a single GPU), and inside a run shard after
// shard: V regions of run_len / V draws, region v drawing from shard v's sample range -- lo_v +
// floor(size_v * u).  The last run of an epoch may be shorter.  Positions past V * (run_len / V) of a run keep
// the plain floor(n * u) and are not consumed.
struct RngShards {
  int V;
  int64_t run;       // draws per run (the last one: what is left)
  int64_t dps;       // draws per shard in a full run = run / V
  double inv_run;    // 1 / run: the run of a position without a 64-bit division
  double lo[8], size[8];
};

// (64-bit integer divisions are software routines of ~100 instructions on this part, and this kernel runs on
// whatever CUs are free between two gather launches, i.e. in front of the next one: position -> (run, shard) is a
// multiplication with one correction step and a handful of compares.)
// one position of the stream: the raw word -> the draw (virtual shards: of the shard that owns the position)
__device__ __forceinline__ uint32_t convert_at(int64_t i, uint32_t word, int64_t count, double n, const RngShards& sh) {
  if (sh.V > 1) {
    int64_t r = 0, j = i;
    if (sh.run < count) {
      r = (int64_t)((double)i * sh.inv_run);
      j = i - r * sh.run;
      if (j < 0) {
        --r;
        j += sh.run;
      } else if (j >= sh.run) {
        ++r;
        j -= sh.run;
      }
    }
    const int64_t left = count - r * sh.run;
    const int64_t dps = left >= sh.run ? sh.dps : left / sh.V;      // the division: in the last, shorter run only
    if (dps > 0 && j < dps * sh.V) {
      int v = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) v += (q < sh.V && j >= (int64_t)q * dps) ? 1 : 0;
      return (uint32_t)sh.lo[v] + word_to_draw(word, sh.size[v]);
    }
  }
  return word_to_draw(word, n);
}

__global__ __launch_bounds__(256) void r_mt_convert_kernel(uint32_t* out, int64_t count, uint32_t n_samples,
                                                           RngShards sh) {
  const double n = (double)n_samples;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    out[i] = convert_at(i, out[i], count, n, sh);
}

// The same on a FEW CUs (round 4): while the fused epoch kernel of the virtual shards holds every CU but the
// generators' own for a whole epoch, 2048 small workgroups with one 4-byte load in flight per thread have nowhere to
// run but those few CUs (0.8 ms per 10M draws there).  Here a launch is two 1024-thread workgroups per reserved CU
// and a thread keeps four 16-byte loads in flight.
constexpr int kConvU = 4;
__global__ __launch_bounds__(1024) void r_mt_convert_narrow_kernel(uint32_t* out, int64_t count, uint32_t n_samples,
                                                                   RngShards sh) {
  const double n = (double)n_samples;
  const int64_t n4 = count >> 2;
  uint4* out4 = reinterpret_cast<uint4*>(out);
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; i0 < n4; i0 += stride * kConvU) {
    uint4 x[kConvU];
#pragma unroll
    for (int u = 0; u < kConvU; ++u) {
      const int64_t q = i0 + u * stride;
      x[u] = q < n4 ? out4[q] : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < kConvU; ++u) {
      const int64_t q = i0 + u * stride;
      if (q < n4) {
        uint4 y;
        y.x = convert_at(4 * q, x[u].x, count, n, sh);
        y.y = convert_at(4 * q + 1, x[u].y, count, n, sh);
        y.z = convert_at(4 * q + 2, x[u].z, count, n, sh);
        y.w = convert_at(4 * q + 3, x[u].w, count, n, sh);
        out4[q] = y;
      }
    }
  }
  if (blockIdx.x == 0) {
    const int64_t i = 4 * n4 + threadIdx.x;
    if (i < count) out[i] = convert_at(i, out[i], count, n, sh);
  }
}

// unif_rand() itself (tempering, scaling to [0, 1), fixup into (0, 1)): the draws stats::runif() hands score()'s AUC
// for its tie breakers (R/score.R:221)
__global__ __launch_bounds__(256) void r_mt_unif_kernel(const uint32_t* raw, double* out, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    uint32_t y = raw[i];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    const double i2_32m1 = 2.328306437080797e-10;
    double u = (double)y * 2.3283064365386963e-10;
    if (u <= 0.0) u = 0.5 * i2_32m1;
    if (1.0 - u <= 0.0) u = 1.0 - 0.5 * i2_32m1;
    out[i] = u;
  }
}

// `count` consecutive unif_rand() of the generator in state_in (one sgdnet_rng on the device) into out; raw: scratch of
// `count` words; state_out (may alias state_in): the generator after them
int launch_rng_unif(const uint32_t* state_in, uint32_t* state_out, uint32_t* raw, double* out, int64_t count,
                    hipStream_t st) {
  hipLaunchKernelGGL(r_mt_state_kernel, dim3(1), dim3(kRngBlock * kGenPerWg), 0, st, state_in, state_out, raw, count,
                     count, 1);
  int grid = (int)((count + 256 * 8 - 1) / (256 * 8));
  if (grid < 1) grid = 1;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(r_mt_unif_kernel, dim3(grid), dim3(256), 0, st, raw, out, count);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

static RngShards make_shards(int n_shards, const double* shard_size, int64_t count, int64_t run_len) {
  RngShards sh{};
  sh.V = n_shards;
  if (n_shards > 1) {
    sh.run = run_len > 0 && run_len < count ? run_len : count;
    sh.dps = sh.run / n_shards;
    sh.inv_run = 1.0 / (double)sh.run;
    double lo = 0.0;
    for (int v = 0; v < n_shards; ++v) {
      sh.lo[v] = lo;
      sh.size[v] = shard_size[v];
      lo += shard_size[v];
    }
  }
  return sh;
}

// raw words -> draws, in place (the second half of launch_rng_fill; also run on its own on a slot of the sample-order
// pipeline that was left raw for the fused epoch kernel and is consumed by other kernels after all)
int launch_rng_convert(uint32_t* out, int64_t count, uint32_t n_samples, hipStream_t st, int n_shards,
                       const double* shard_size, int64_t run_len, int narrow_cus) {
  const RngShards sh = make_shards(n_shards, shard_size, count, run_len);
  if (narrow_cus > 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(r_mt_convert_narrow_kernel, dim3(2 * narrow_cus), dim3(1024), 0, st, out, count, n_samples, sh);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  int grid = (int)((count + 256 * 8 - 1) / (256 * 8));
  if (grid < 1) grid = 1;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(r_mt_convert_kernel, dim3(grid), dim3(256), 0, st, out, count, n_samples, sh);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// state_in -> state_out (may alias); raw words then draws into out[0, count)
// convert: 1 the wide conversion kernel, 2 the narrow one (narrow_cus CUs), 0 none: the slot keeps the raw words (the
// fused epoch kernel of the virtual shards converts its own shares, saga_batched.hip)
int launch_rng_fill(const uint32_t* state_in, uint32_t* state_out, uint32_t n_samples, uint32_t* out,
                    int64_t count, hipStream_t st, int n_shards, const double* shard_size, int gens, int64_t run_len,
                    int convert, int narrow_cus, int wgs) {
  if (gens < 1) gens = 1;
  const int64_t seg = (count + gens - 1) / gens;
  int grid = (gens + kGenPerWg - 1) / kGenPerWg;
  if (wgs > grid) grid = wgs < gens ? wgs : gens;          // the generators spread over the workgroups they were given
  hipLaunchKernelGGL(r_mt_state_kernel, dim3(grid), dim3(kRngBlock * kGenPerWg), 0, st,
                     state_in, state_out, out, count, seg, gens);
  SGD_HIP_TRY(hipGetLastError());
  if (!convert) return SGDNET_OK;
  return launch_rng_convert(out, count, n_samples, st, n_shards, shard_size, run_len, convert == 2 ? narrow_cus : 0);
}

}  // namespace sgdnet
