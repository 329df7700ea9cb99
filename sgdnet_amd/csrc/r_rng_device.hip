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

namespace sgdnet {

namespace {

constexpr int kN = 624, kM = 397;

__device__ __forceinline__ uint32_t twist(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
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
constexpr int kRngBlock = 256;      // threads that carry one generator
constexpr int kGenPerWg = 4;        // generators per workgroup (they step in lock-step: one barrier per block serves all)

// Several generators on ONE R stream (batched mode, solver.cpp: solver_rng_open): generator g (state g of
// st_in / st_out) fills the g-th segment of `seg` words.  A generator is latency-bound (one barrier per 624
// words, 227 busy lanes), so kGenPerWg of them share a workgroup -- and with it a CU: the sample order of a C4
// epoch then holds 8 CUs instead of 32 for the same ~0.18 ms (the LDS gather forms give their workgroups'
// CUs up to the generators, lds_target_grid).
__global__ __launch_bounds__(kRngBlock* kGenPerWg) void r_mt_state_kernel(const uint32_t* st_in, uint32_t* st_out,
                                                                          uint32_t* out, int64_t count, int64_t seg,
                                                                          int gens) {
  __shared__ uint32_t bufs[kGenPerWg][2][kN + 1];
  const int t = threadIdx.x & (kRngBlock - 1);
  const int sub = threadIdx.x / kRngBlock;
  const int64_t g = (int64_t)blockIdx.x * kGenPerWg + sub;
  const bool live = g < gens;
  uint32_t(*buf)[kN + 1] = bufs[sub];
  int64_t max_count = 0;                      // the longest segment of this workgroup: its generators loop together
  {
    const int64_t g0 = (int64_t)blockIdx.x * kGenPerWg;
    const int64_t left0 = count - g0 * seg;
    max_count = left0 < 0 ? 0 : (left0 < seg ? left0 : seg);
  }
  if (live) {
    st_in += g * (kN + 1);
    st_out += g * (kN + 1);
    out += g * seg;
    const int64_t left = count - g * seg;
    count = left < 0 ? 0 : (left < seg ? left : seg);
  } else {
    count = 0;
  }
  uint32_t mti = 0;
  if (live) {
    for (int i = t; i < kN; i += kRngBlock) buf[0][i] = st_in[1 + i];
    mti = st_in[0];
  }
  __syncthreads();
  int64_t produced = 0;
  if (live) {  // words left in the current block
    const int64_t left = mti < (uint32_t)kN ? (int64_t)(kN - mti) : 0;
    const int64_t take = left < count ? left : count;
    for (int64_t i = t; i < take; i += kRngBlock) out[i] = buf[0][mti + i];
    produced = take;
    mti += (uint32_t)take;
  }
  constexpr int kD = kN - kM;   // 227
  int c = 0;
  // every generator of the workgroup makes the same number of trips (the barrier is the workgroup's): the
  // first generator's segment is the longest, and all of them start with at most kN words in hand
  const int64_t trips = (max_count + kN - 1) / kN + 1;
  for (int64_t trip = 0; trip < trips; ++trip) {
    const bool on = produced < count;
    if (on) {
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
      produced += take;
      mti = (uint32_t)take;
    }
    // publish the new block: only the LDS writes have to be complete -- __syncthreads() would
    // also wait for the global stores of the raw words above (vmcnt(0)), which nobody in this
    // kernel reads, and one store round trip per 624 words is what bounded the generator
    // (0.33 us per block on an idle chip, ~2.5 us next to a running epoch)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (on) c ^= 1;
  }
  if (live) {
    for (int i = t; i < kN; i += kRngBlock) st_out[1 + i] = buf[c][i];
    if (t == 0) st_out[0] = mti;
  }
}

// Jump-ahead on the device (mt_jump.cpp has the mathematics and the host form): a generator's state window
// moves J words down its stream, J given by poly = x^J mod phi:
//   out[j] = XOR over { i : poly_i = 1 } of x[i + j],  x = the raw word sequence that starts with the window.
// The 19937 + 624 words of x live in LDS (82 KB); they are produced block by block with the same three-phase
// step as r_mt_state_kernel.  mti is kept.
//
// The sum is ~10 000 terms x 624 words per generator.  Round 2 gave thread j the word j and walked the
// polynomial with a scalar bit scan: one LDS read, one XOR and ~8 bookkeeping instructions per term on ten
// wavefronts -- 0.4 ms per launch, 41 % of all kernel time of a C4 epoch.  Now the 624 polynomial words are cut
// into one slice per wavefront (16 x 39 words) and a lane owns kJR = 11 CONSECUTIVE output words: walking its
// slice bit by bit it keeps x[i + 11 l .. i + 11 l + 10] in registers, so a step is ONE new LDS word per lane
// (stride 11: conflict-free) behind a wave-uniform branch on the polynomial bit, and a set bit costs 11 XORs
// with no memory access at all; the register window rotates through an 11-step unrolled body.  The 16 partial
// windows are XORed through LDS.  ~25 us per generator instead of 400; a workgroup handles its generators one
// after the other so that the launch holds no more CUs than the generators' own (lds_target_grid).
constexpr int kJumpBlock = 1024;
constexpr int kJumpWaves = kJumpBlock / 64;  // 16 polynomial slices
constexpr int kJR = 11;                      // consecutive output words per lane (57 lanes x 11 >= 624)
constexpr int kJLanes = (kN + kJR - 1) / kJR;
constexpr int kJOut = kJLanes * kJR;         // 627
constexpr int kPolyPerWave = kN / kJumpWaves;   // 39 polynomial words per wavefront
static_assert(kPolyPerWave * kJumpWaves == kN, "the polynomial splits evenly over the wavefronts");
constexpr int kJumpXs = 33 * kN;             // 20 592 words: the windows read x[0 .. 19 936 + 11 + 10 + 11 * 56]
static_assert(kJumpXs >= 19937 + 2 * kJR + kJR * (kJLanes - 1), "the sequence covers every window");
constexpr size_t kJumpLds = sizeof(uint32_t) * (size_t)(kJumpXs + kN + kJumpWaves * kJOut);

__global__ __launch_bounds__(kJumpBlock) void r_mt_jump_kernel(const uint32_t* st_in_all, uint32_t* st_out_all,
                                                               const uint32_t* poly, int gens) {
  extern __shared__ uint32_t xs[];           // [kJumpXs] sequence | [kN] polynomial | [16][kJOut] partial windows
  uint32_t* pl = xs + kJumpXs;
  uint32_t* part = pl + kN;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  if (t < kN) pl[t] = poly[t];
  for (int g = blockIdx.x; g < gens; g += gridDim.x) {
    const uint32_t* st_in = st_in_all + (int64_t)g * (kN + 1);
    uint32_t* st_out = st_out_all + (int64_t)g * (kN + 1);
    __syncthreads();                           // the previous generator's partial windows have been read
    if (t < kN) xs[t] = st_in[1 + t];
    __syncthreads();
    constexpr int kD = kN - kM;
    for (int base = 0; base + 2 * kN <= kJumpXs; base += kN) {
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
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // wavefront `wave`: polynomial bits [lo, hi); lane l: output words 11 l .. 11 l + 10
    const int lo = wave * kPolyPerWave * 32;
    const int hi_raw = lo + kPolyPerWave * 32;
    const int hi = hi_raw < 19937 ? hi_raw : 19937;
    const bool lane_on = lane < kJLanes;
    const uint32_t* xl = xs + (lane_on ? kJR * lane : 0);
    uint32_t acc[kJR], win[kJR];
#pragma unroll
    for (int r = 0; r < kJR; ++r) {
      acc[r] = 0u;
      win[r] = xl[lo + r];
    }
    for (int i0 = lo; i0 < hi; i0 += kJR) {
      // the next kJR polynomial bits as one wave-uniform field
      const int wi = i0 >> 5, sh = i0 & 31;
      const uint32_t w0 = __builtin_amdgcn_readfirstlane(pl[wi]);
      const uint32_t w1 = __builtin_amdgcn_readfirstlane(pl[wi + 1 < kN ? wi + 1 : wi]);
      uint32_t field = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh);
      const int left = hi - i0;
      field &= left >= kJR ? ((1u << kJR) - 1u) : ((1u << left) - 1u);
      const uint32_t* xn = xl + i0 + kJR;      // word entering the window after step k: xn[k]
#pragma unroll
      for (int k = 0; k < kJR; ++k) {
        // logical window word r of step k sits in win[(r + k) % kJR]
        if (field & (1u << k)) {
#pragma unroll
          for (int r = 0; r < kJR; ++r) acc[r] ^= win[(r + k) % kJR];
        }
        win[k] = xn[k];
      }
    }
    if (lane_on) {
#pragma unroll
      for (int r = 0; r < kJR; ++r) part[wave * kJOut + kJR * lane + r] = acc[r];
    }
    __syncthreads();
    if (t < kN) {
      uint32_t v = 0u;
#pragma unroll
      for (int wv = 0; wv < kJumpWaves; ++wv) v ^= part[wv * kJOut + t];
      st_out[1 + t] = v;
    }
    if (t == 0) st_out[0] = st_in[0];
  }
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
                    int convert, int narrow_cus) {
  if (gens < 1) gens = 1;
  const int64_t seg = (count + gens - 1) / gens;
  hipLaunchKernelGGL(r_mt_state_kernel, dim3((gens + kGenPerWg - 1) / kGenPerWg), dim3(kRngBlock * kGenPerWg), 0, st,
                     state_in, state_out, out, count, seg, gens);
  SGD_HIP_TRY(hipGetLastError());
  if (!convert) return SGDNET_OK;
  return launch_rng_convert(out, count, n_samples, st, n_shards, shard_size, run_len, convert == 2 ? narrow_cus : 0);
}

}  // namespace sgdnet
