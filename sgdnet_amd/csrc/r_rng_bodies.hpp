// R's Mersenne-Twister on the device: the bodies of the state and jump-ahead kernels (r_rng_device.hip), shared
// with the fused epoch kernel of the virtual shards (saga_batched.hip), whose spare workgroups produce the NEXT
// epoch's sample order while the epoch runs (round 4).
#pragma once

#include <stdint.h>

#include <hip/hip_runtime.h>

namespace sgdnet {

constexpr int kMtN = 624, kMtM = 397;

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

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
// blk / nblk: this workgroup's index among the generators' workgroups and their number -- workgroup b carries
// generators b, b + nblk, b + 2 nblk, ... (at most kGenPerWg of them; nblk * kGenPerWg >= gens), so that a launch with
// more workgroups than ceil(gens / kGenPerWg) spreads the generators (short epochs: the generators' workgroups must
// not outlast the epoch they run beside); bufs: kGenPerWg x 2 x (kN + 1) words of LDS
__device__ __forceinline__ void mt_state_body(int blk, int nblk, uint32_t (*bufs)[2][kMtN + 1], const uint32_t* st_in,
                                              uint32_t* st_out, uint32_t* out, int64_t count, int64_t seg, int gens) {
  constexpr int kN = kMtN, kM = kMtM;
  const int t = threadIdx.x & (kRngBlock - 1);
  const int sub = threadIdx.x / kRngBlock;
  const int64_t g = (int64_t)sub * nblk + blk;
  const bool live = g < gens;
  uint32_t(*buf)[kN + 1] = bufs[sub];
  int64_t max_count = 0;                      // the longest segment of this workgroup: its generators loop together
  {
    const int64_t g0 = (int64_t)blk;                // the workgroup's first generator: the longest segment among its own
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
        v = cur[t + kM] ^ mt_twist(cur[t], cur[t + 1]);                       // word t
        nxt[t] = v;
        if (t < take) out[produced + t] = v;
        const int k2 = kD + t;
        v ^= mt_twist(cur[k2], cur[k2 + 1]);                                  // word 227 + t
        nxt[k2] = v;
        if (k2 < take) out[produced + k2] = v;
        const int k3 = 2 * kD + t;
        if (k3 < kN) {                                                      // word 454 + t
          const uint32_t nb = k3 == kN - 1 ? (cur[kM] ^ mt_twist(cur[0], cur[1])) : cur[k3 + 1];
          v ^= mt_twist(cur[k3], nb);
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
constexpr int kJLanes = (kMtN + kJR - 1) / kJR;
constexpr int kJOut = kJLanes * kJR;         // 627
constexpr int kPolyPerWave = kMtN / kJumpWaves;   // 39 polynomial words per wavefront
static_assert(kPolyPerWave * kJumpWaves == kMtN, "the polynomial splits evenly over the wavefronts");
constexpr int kJumpXs = 33 * kMtN;             // 20 592 words: the windows read x[0 .. 19 936 + 11 + 10 + 11 * 56]
static_assert(kJumpXs >= 19937 + 2 * kJR + kJR * (kJLanes - 1), "the sequence covers every window");
constexpr size_t kJumpLds = sizeof(uint32_t) * (size_t)(kJumpXs + kMtN + kJumpWaves * kJOut);

// blk / nblk: this workgroup's index among the generators' workgroups and their number (a workgroup takes its
// generators in turn); xs: kJumpLds bytes of LDS -- [kJumpXs] sequence | [kN] polynomial | [16][kJOut] partial windows
__device__ __forceinline__ void mt_jump_body(int blk, int nblk, uint32_t* xs, const uint32_t* st_in_all,
                                             uint32_t* st_out_all, const uint32_t* poly, int gens) {
  constexpr int kN = kMtN, kM = kMtM;
  uint32_t* pl = xs + kJumpXs;
  uint32_t* part = pl + kN;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  if (t < kN) pl[t] = poly[t];
  for (int g = blk; g < gens; g += nblk) {
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
        uint32_t v = cur[t + kM] ^ mt_twist(cur[t], cur[t + 1]);
        nxt[t] = v;
        const int k2 = kD + t;
        v ^= mt_twist(cur[k2], cur[k2 + 1]);
        nxt[k2] = v;
        const int k3 = 2 * kD + t;
        if (k3 < kN) {
          const uint32_t nb = k3 == kN - 1 ? (cur[kM] ^ mt_twist(cur[0], cur[1])) : cur[k3 + 1];
          v ^= mt_twist(cur[k3], nb);
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

}  // namespace sgdnet
