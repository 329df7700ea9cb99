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

namespace sgdnet {

namespace {

constexpr int kN = 624, kM = 397;

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

// st_in / st_out: [0] = mti, [1..624] = mt (the layout of sgdnet_rng).
//
// One 256-thread workgroup carries the sequence (it is inherently serial across 624-word
// blocks).  Thread t makes the words t, 227 + t and 454 + t of the next block: the second needs
// the first and the third the second -- the thread's own registers -- plus words of the OLD
// block, so the three phases need no barrier between them (the one exception, word 623, needs
// new word 0, which its thread recomputes from the old block).  One barrier per block then
// publishes the new block.  The kernel writes raw state words; tempering, the unif_rand() scaling
// and floor(n * u) are embarrassingly parallel and run as a second, wide kernel in place.
// (Three barriers per block and the conversion inside the loop: 10.3 ms per 10M draws.)
constexpr int kRngBlock = 256;

// Several independent generators (batched mode, solver.cpp: solver_rng_open): workgroup g carries
// generator g (state g of st_in / st_out) and fills the g-th segment of `seg` words.
__global__ __launch_bounds__(kRngBlock) void r_mt_state_kernel(const uint32_t* st_in, uint32_t* st_out,
                                                               uint32_t* out, int64_t count, int64_t seg) {
  __shared__ uint32_t buf[2][kN + 1];
  const int t = threadIdx.x;
  {
    const int64_t g = blockIdx.x;
    st_in += g * (kN + 1);
    st_out += g * (kN + 1);
    out += g * seg;
    const int64_t left = count - g * seg;
    count = left < 0 ? 0 : (left < seg ? left : seg);
  }
  for (int i = t; i < kN; i += kRngBlock) buf[0][i] = st_in[1 + i];
  uint32_t mti = st_in[0];
  __syncthreads();
  int64_t produced = 0;
  {  // words left in the current block
    const int64_t left = mti < (uint32_t)kN ? (int64_t)(kN - mti) : 0;
    const int64_t take = left < count ? left : count;
    for (int64_t i = t; i < take; i += kRngBlock) out[i] = buf[0][mti + i];
    produced = take;
    mti += (uint32_t)take;
  }
  constexpr int kD = kN - kM;   // 227
  int c = 0;
  while (produced < count) {
    const uint32_t* cur = buf[c];
    uint32_t* nxt = buf[c ^ 1];
    const int64_t rest = count - produced;
    const int take = rest < kN ? (int)rest : kN;
    uint32_t v = 0;
    if (t < kD) {
      v = cur[t + kM] ^ twist(cur[t], cur[t + 1]);                       // word t
      nxt[t] = v;
      if (t < take) out[produced + t] = v;
      const int k2 = kD + t;
      v ^= twist(cur[k2], cur[k2 + 1]);                                  // word 227 + t
      nxt[k2] = v;
      if (k2 < take) out[produced + k2] = v;
      const int k3 = 2 * kD + t;
      if (k3 < kN) {                                                      // word 454 + t
        const uint32_t nb = k3 == kN - 1 ? (cur[kM] ^ twist(cur[0], cur[1])) : cur[k3 + 1];
        v ^= twist(cur[k3], nb);
        nxt[k3] = v;
        if (k3 < take) out[produced + k3] = v;
      }
    }
    // publish the new block: only the LDS writes have to be complete -- __syncthreads() would
    // also wait for the global stores of the raw words above (vmcnt(0)), which nobody in this
    // kernel reads, and one store round trip per 624 words is what bounded the generator
    // (0.33 us per block on an idle chip, ~2.5 us next to a running epoch)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    c ^= 1;
    produced += take;
    mti = (uint32_t)take;
  }
  for (int i = t; i < kN; i += kRngBlock) st_out[1 + i] = buf[c][i];
  if (t == 0) st_out[0] = mti;
}

// Jump-ahead on the device (mt_jump.cpp has the mathematics and the host form): workgroup g moves
// generator g's state window J words down its stream, J given by poly = x^J mod phi:
//   out[j] = XOR over { i : poly_i = 1 } of x[i + j],  x = the raw word sequence that starts with the window.
// The 19937 + 624 words of x live in LDS (82 KB); they are produced block by block with the same
// three-phase step as r_mt_state_kernel, then thread j accumulates its word.  mti is kept.
constexpr int kJumpSeq = 19937 + kN;         // words of x the convolution reads
constexpr int kJumpBlock = 640;              // >= 624 threads: one per word of the window

__global__ __launch_bounds__(kJumpBlock) void r_mt_jump_kernel(const uint32_t* st_in, uint32_t* st_out,
                                                               const uint32_t* poly) {
  extern __shared__ uint32_t xs[];           // kJumpSeq + kN words (the last block may run over)
  const int t = threadIdx.x;
  st_in += (int64_t)blockIdx.x * (kN + 1);
  st_out += (int64_t)blockIdx.x * (kN + 1);
  if (t < kN) {
    xs[t] = st_in[1 + t];
    xs[kJumpSeq + kN + t] = poly[t];
  }
  __syncthreads();
  constexpr int kD = kN - kM;
  for (int base = 0; base + kN < kJumpSeq; base += kN) {
    const uint32_t* cur = xs + base;
    uint32_t* nxt = xs + base + kN;
    if (t < kD) {
      uint32_t v = cur[t + kM] ^ twist(cur[t], cur[t + 1]);
      nxt[t] = v;
      const int k2 = kD + t;
      v ^= twist(cur[k2], cur[k2 + 1]);
      nxt[k2] = v;
      const int k3 = 2 * kD + t;
      if (k3 < kN) {
        const uint32_t nb = k3 == kN - 1 ? (cur[kM] ^ twist(cur[0], cur[1])) : cur[k3 + 1];
        v ^= twist(cur[k3], nb);
        nxt[k3] = v;
      }
    }
    __syncthreads();
  }
  if (t < kN) {
    // The polynomial sits in LDS behind the sequence (every thread reads the same word: a broadcast);
    // a term is one LDS read + one XOR behind a scalar bit scan.  ~0.4 ms per launch: the ~10 000 terms
    // per thread cost ~10 issued instructions each (requesting eight reads at a time measured slower).
    uint32_t acc = 0u;
    const uint32_t* pl = xs + kJumpSeq + kN;
    uint32_t bits_next = __builtin_amdgcn_readfirstlane(pl[0]);
    for (int wi = 0; wi < kN; ++wi) {
      uint32_t bits = bits_next;
      bits_next = __builtin_amdgcn_readfirstlane(pl[wi + 1 < kN ? wi + 1 : wi]);
      const uint32_t* xb = xs + 32 * wi + t;
      while (bits) {
        const int b = __ffs((int)bits) - 1;
        acc ^= xb[b];
        bits &= bits - 1u;
      }
    }
    st_out[1 + t] = acc;
  }
  if (t == 0) st_out[0] = st_in[0];
}

int launch_rng_jump(const uint32_t* state_in, uint32_t* state_out, const uint32_t* poly_dev, int gens,
                    hipStream_t st) {
  static bool attr_done_dev[64] = {};
  int cur = 0;
  (void)hipGetDevice(&cur);
  const size_t lds = sizeof(uint32_t) * (size_t)(kJumpSeq + 2 * kN);
  if (!attr_done_dev[cur & 63]) {
    SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(r_mt_jump_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done_dev[cur & 63] = true;
  }
  hipLaunchKernelGGL(r_mt_jump_kernel, dim3(gens), dim3(kJumpBlock), lds, st, state_in, state_out, poly_dev);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// shards.V > 1 (virtual shards, common.hpp): the epoch's positions are split into V regions of
// dps draws and region v draws from shard v's sample range -- lo_v + floor(size_v * u); positions
// past V * dps (n not a multiple of V) keep the plain floor(n * u) and are not consumed
struct RngShards {
  int V;
  int64_t dps;
  double lo[8], size[8];
};

__global__ __launch_bounds__(256) void r_mt_convert_kernel(uint32_t* out, int64_t count, uint32_t n_samples,
                                                           RngShards sh) {
  const double n = (double)n_samples;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    if (sh.V > 1 && i < sh.dps * sh.V) {
      const int v = (int)(i / sh.dps);
      out[i] = (uint32_t)sh.lo[v] + word_to_draw(out[i], sh.size[v]);
    } else {
      out[i] = word_to_draw(out[i], n);
    }
  }
}

// state_in -> state_out (may alias); raw words then draws into out[0, count)
int launch_rng_fill(const uint32_t* state_in, uint32_t* state_out, uint32_t n_samples, uint32_t* out,
                    int64_t count, hipStream_t st, int n_shards, const double* shard_size, int gens) {
  RngShards sh{};
  sh.V = n_shards;
  if (n_shards > 1) {
    sh.dps = count / n_shards;
    double lo = 0.0;
    for (int v = 0; v < n_shards; ++v) {
      sh.lo[v] = lo;
      sh.size[v] = shard_size[v];
      lo += shard_size[v];
    }
  }
  if (gens < 1) gens = 1;
  const int64_t seg = (count + gens - 1) / gens;
  hipLaunchKernelGGL(r_mt_state_kernel, dim3(gens), dim3(kRngBlock), 0, st, state_in, state_out, out, count, seg);
  int grid = (int)((count + 256 * 8 - 1) / (256 * 8));
  if (grid < 1) grid = 1;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(r_mt_convert_kernel, dim3(grid), dim3(256), 0, st, out, count, n_samples, sh);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

}  // namespace sgdnet
