// R-compatible Mersenne-Twister on the device: the sample order of a fit without a
// round trip through the host.
//
// The reference draws floor(R::runif(0, n)) once per inner iteration
// (src/saga-sparse.h:261, src/saga-dense.h:152).  At 10M draws per epoch the host
// generator (r_rng.cpp, ~50 ms per epoch plus a 40 MB upload) would bound a batched fit whose
// epoch takes ~3 ms, so the same stream is produced in HBM.  MT19937's recurrence
//   x[k+624] = x[k+397] ^ twist(x[k], x[k+1])
// has dependency distance 227 (= 624 - 397): a block of 624 words is regenerated in three
// barrier-separated phases of 227 / 227 / 170 independent words (double-buffered in LDS), then
// tempered, scaled exactly like unif_rand() and written out coalesced.  One workgroup carries
// the sequence (it is inherently serial across blocks); the other 255 CUs keep running SAGA.
#include "common.hpp"

namespace sgdnet {

namespace {

constexpr int kN = 624, kM = 397, kRngBlock = 256;

__device__ __forceinline__ uint32_t twist(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// tempering + unif_rand() scaling/fixup + floor(n * u), as r_rng.cpp does on the host
__device__ __forceinline__ uint32_t word_to_draw(uint32_t y, double n) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  const double i2_32m1 = 2.328306437080797e-10;
  double u = (double)y * 2.3283064365386963e-10;
  if (u <= 0.0) u = 0.5 * i2_32m1;
  if (1.0 - u <= 0.0) u = 1.0 - 0.5 * i2_32m1;
  return (uint32_t)floor(0.0 + (n - 0.0) * u);
}

}  // namespace

// st: [0] = mti, [1..624] = mt (the layout of sgdnet_rng); updated in place.
__global__ __launch_bounds__(kRngBlock) void r_mt_fill_kernel(uint32_t* st, uint32_t n_samples,
                                                              uint32_t* out, int64_t count) {
  __shared__ uint32_t bufA[kN], bufB[kN];
  const int tid = threadIdx.x;
  const double n = (double)n_samples;
  for (int i = tid; i < kN; i += kRngBlock) bufA[i] = st[1 + i];
  uint32_t mti = st[0];
  __syncthreads();
  uint32_t* cur = bufA;
  uint32_t* nxt = bufB;

  int64_t produced = 0;
  {  // words left in the current block
    const int64_t left = mti < (uint32_t)kN ? (int64_t)(kN - mti) : 0;
    const int64_t take = left < count ? left : count;
    for (int64_t i = tid; i < take; i += kRngBlock) out[i] = word_to_draw(cur[mti + i], n);
    produced = take;
    mti += (uint32_t)take;
  }
  while (produced < count) {
    if (tid < kN - kM) nxt[tid] = cur[tid + kM] ^ twist(cur[tid], cur[tid + 1]);
    __syncthreads();
    {
      const int k = (kN - kM) + tid;
      if (k < 2 * (kN - kM)) nxt[k] = nxt[k - (kN - kM)] ^ twist(cur[k], cur[k + 1]);
    }
    __syncthreads();
    {
      const int k = 2 * (kN - kM) + tid;
      if (k < kN) nxt[k] = nxt[k - (kN - kM)] ^ twist(cur[k], k == kN - 1 ? nxt[0] : cur[k + 1]);
    }
    __syncthreads();
    uint32_t* t = cur;
    cur = nxt;
    nxt = t;
    const int64_t rest = count - produced;
    const int64_t take = rest < kN ? rest : kN;
    for (int64_t i = tid; i < take; i += kRngBlock) out[produced + i] = word_to_draw(cur[i], n);
    produced += take;
    mti = (uint32_t)take;
  }
  __syncthreads();
  for (int i = tid; i < kN; i += kRngBlock) st[1 + i] = cur[i];
  if (tid == 0) st[0] = mti;
}

int launch_rng_fill(uint32_t* state_dev, uint32_t n_samples, uint32_t* out, int64_t count, hipStream_t st) {
  hipLaunchKernelGGL(r_mt_fill_kernel, dim3(1), dim3(kRngBlock), 0, st, state_dev, n_samples, out, count);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

}  // namespace sgdnet
