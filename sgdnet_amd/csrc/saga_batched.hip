// Batched ("B-stale") sparse SAGA kernels for gfx950: the throughput path.
//
// A batch is `m` consecutive draws of the sample stream evaluated against one
// snapshot of (w, intercept).  In unscaled coordinates (true w = wscale * w of
// the reference) the reference iteration src/saga-sparse.h:258-337 applied to
// the batch becomes, per feature j (DESIGN.md "Batched mode"):
//
//   gather : per draw i   lp = w . x_s + b ; g = Gradient(lp, y_s)            (:274, :279)
//                         gc = g - g_memory[s] ; g_memory[s] = g              (:281-282)
//                         D[:, j] += x_sj * gc   for j in nz(x_s)             (:306-313, :328-335)
//   sweep  : per feature  w_j = r^m w_j - gamma LS_m G_j - gamma D_j ; prox   (:316-325 + penalties.h)
//                         G_j += D_j / n ;  D_j = 0
//            intercept    gb += d0/n ; b -= gamma (0.01 m gb + d0/n)          (:300-304; dense x: m gb, saga-dense.h:170-173)
//
// with r = 1 - alpha*gamma and LS_m = sum_{k<m} r^k (= lag_scaling[m], :229-240).
// m == 1 is the reference iteration itself.  A sample drawn twice inside one
// batch sees the same snapshot, so its second draw has gc == 0: the first draw
// claims the sample (atomic exchange) and later ones contribute nothing.
//
// Memory behaviour: the gather kernel is the HBM-bound one -- per draw it pulls
// one stream entry, the sample's packed record (y, z indices, z values) and the
// gradient memory (algorithmic 16 + 12 z + 16 K bytes) at random
// sample positions; 16-lane groups own one draw so that a wavefront has 4
// independent gathers in flight and reduces x.w with intra-row shuffles.  w, D
// and G are K*p doubles (80 KB at 10k features) and stay L2 / Infinity-Cache
// resident.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include "device_math.hpp"
#include "r_rng_word.hpp"
#include "r_rng_bodies.hpp"

#ifndef SGDNET_BIN_BLOCK
#define SGDNET_BIN_BLOCK 1024
#endif
#ifndef SGDNET_BIN_W
#define SGDNET_BIN_W 8
#endif
#ifndef SGDNET_RANGE_BLOCK
#define SGDNET_RANGE_BLOCK 512
#endif

namespace sgdnet {

namespace {

constexpr int kGroup = 16;          // lanes per draw
constexpr int kBlock = 256;

__device__ __forceinline__ double group_sum(double v) {
  v += __shfl_xor(v, 8, kGroup);
  v += __shfl_xor(v, 4, kGroup);
  v += __shfl_xor(v, 2, kGroup);
  v += __shfl_xor(v, 1, kGroup);
  return v;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  // no-return global_atomic_add_f64
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, double v) {
  // v >= 0: the IEEE bit pattern is monotone in v
  __hip_atomic_fetch_max(p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
}

// Implicit centring of sparse x (standardize = TRUE; reference saga-sparse.h:127-128,
// 276-277 does it with dense O(p) work per iteration).  Against a snapshot of w it folds into
// two per-batch scalars per class: lp -= c.w and D_j -= c_j * sum_i gc_i.  c.w lives in two
// sets of 16 accumulation slots: the sweep of batch B adds its blocks' partial sums of
// c_j * w_new into set (B+1)&1 (zeroed by gather B), gather B+1 reads that set.
constexpr int kCwSlots = 16;

// Intercept accumulator d0[k] = sum_i gc_ik: two sets (batch parity) of kD0Slots slots.  A
// gather with at most kD0Slots workgroups stores one partial per workgroup; a larger grid adds
// atomically into slot (workgroup % kD0Slots) of a set the previous sweep left zeroed.  Either
// way the sweep sums at most kD0Slots values per class (thousands of same-address atomics, or
// thousands of partials summed by one block, would each cost tens of microseconds).
constexpr int kD0Slots = 256;

__device__ __forceinline__ double* d0_set(const SagaDev& d, int batch_id) {
  return d.d0_part + (size_t)(batch_id & 1) * kD0Slots * d.K;
}

__device__ __forceinline__ void d0_publish(const SagaDev& d, int batch_id, int k, double tot) {
  double* set = d0_set(d, batch_id);
  if (gridDim.x <= (unsigned)kD0Slots)
    set[(size_t)blockIdx.x * d.K + k] = tot;
  else if (tot != 0.0)
    __hip_atomic_fetch_add(set + (size_t)(blockIdx.x % kD0Slots) * d.K + k, tot, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// The epoch's bookkeeping on the device (captured epoch graphs replay without the host touching LamParams):
// the next epoch's draws follow this one's -- in the two-epoch buffer of the sample-order pipeline, in the other half.
__device__ __forceinline__ void end_epoch(LamParams* lamp, int batches) {
  int64_t sb = lamp->stream_base + lamp->draws_per_epoch;
  if (lamp->stream_wrap > 0 && sb >= lamp->stream_wrap) sb -= lamp->stream_wrap;
  lamp->stream_base = sb;
  lamp->batch_seq += batches;
}

__device__ __forceinline__ double cw_sum(const SagaDev& d, int batch_id, int k) {
  const double* set = d.cw + (size_t)(batch_id & 1) * kCwSlots * d.K;
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < kCwSlots; ++i) t += set[i * d.K + k];
  return t;
}

__device__ __forceinline__ void cw_clear_next(const SagaDev& d, int batch_id) {
  if (blockIdx.x == 0 && (int)threadIdx.x < kCwSlots * d.K)
    d.cw[(size_t)((batch_id + 1) & 1) * kCwSlots * d.K + threadIdx.x] = 0.0;
}

}  // namespace

// --------------------------------------------------------------------------
// Packed sample records (built once per solver, solver.cpp: build_records).
// Random full 128-B lines stream at the HBM rate on MI355X (26 G random 256-B records/s,
// scripts/microbench/gather_rate.hip), partial lines waste it, and every dependent hop is a
// 1-2.5 us round trip, so a draw should touch as few lines as possible, whole, with no
// pointer hop:
//
//   record s at rec + s*stride (stride = 128-B multiple sized for the 90th
//   percentile row; requests are served in 128-B units):  [f64 y][i32 nnz][i32 ovf][i32 idx[cap]] pad8 [f64 val[cap]]
//   rows longer than cap continue in 256-B overflow records:
//                     [i32 next][i32 cnt][i32 idx[20]][f64 val[20]]
//
// At z = 10 a draw is one 256-B record = 2 requests (was: 2 row pointers + y + idx + val ~ 6).
// --------------------------------------------------------------------------
// compact records (below): 128 B per sample, the gradient memory of a one-response fit in the last 8
constexpr int kCStride = 128;
constexpr int kCMOff = 120;

__device__ __forceinline__ double* m_slot(const SagaDev& d, int64_t s) {
  // gradient memory of sample s for the one-response sparse kernels: inside the compact record while the
  // solver keeps it there (solver.cpp: m_to_record / m_to_array), else the K x n array
  // (base and stride are kept by the host: a select between the two addresses in front of the atomic
  //  exchange sends this compiler's instcombine into a segmentation fault)
  return reinterpret_cast<double*>(d.m_base + (size_t)s * (size_t)d.m_stride);
}

constexpr int kOvfStride = 256;
constexpr int kOvfCap = 20;

template <class F>
__device__ __forceinline__ void row_tail_for_each(const SagaDev& d, const char* base, int nnz, int ovf,
                                                  int gl, F f) {
  const int cap = d.rec_cap;
  const int cnt0 = nnz < cap ? nnz : cap;
  const int* ridx = reinterpret_cast<const int*>(base + 16);
  const double* rval = reinterpret_cast<const double*>(base + d.rec_val_off);
  for (int e = kGroup + gl; e < cnt0; e += kGroup) f((int64_t)ridx[e], rval[e]);
  int rem = nnz - cnt0;
  while (rem > 0) {
    const char* ob = d.ovf + (size_t)ovf * kOvfStride;
    const int next = reinterpret_cast<const int*>(ob)[0];
    const int c = reinterpret_cast<const int*>(ob)[1];
    const int* oi = reinterpret_cast<const int*>(ob + 8);
    const double* ov = reinterpret_cast<const double*>(ob + 8 + 4 * kOvfCap);
    for (int e = gl; e < c; e += kGroup) f((int64_t)oi[e], ov[e]);
    rem -= c;
    ovf = next;
  }
}

// --------------------------------------------------------------------------
// One draw, executed by a 16-lane group: gather the record, x.w, gradient,
// gradient-memory update, scatter of x*gc into Dt.  K == 1: the gradient memory
// is claimed, read and updated by ONE atomic exchange (a repeated draw reads back
// the value just stored, so its gc is exactly 0).  K > 1: an int claim per
// sample, then plain loads/stores.  kLds: Dt is a workgroup-private LDS copy of D
// (ds_add_f64), else the global D (global_atomic_add_f64).
// gc[] returns the draw's gradient change on lane 0 of the group, 0 elsewhere.
// --------------------------------------------------------------------------
template <bool kLds>
__device__ __forceinline__ void scatter_add(double* p, double v) {
  if (kLds)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wsrc: the coefficients the draw is evaluated against (d.w, or a virtual shard's replica)
template <int KMAX, bool kLds>
__device__ __forceinline__ void saga_draw(const SagaDev& d, const uint32_t s, const int gl,
                                          const int batch_id, const double (&bk)[KMAX], double* Dt,
                                          double (&gc)[KMAX], const double* wsrc) {
  const int K = KMAX == 1 ? 1 : d.K;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) gc[k] = 0.0;
  const char* base = d.rec + (size_t)s * d.rec_stride;

  // independent of the row (K > 1): claim and old gradient memory
  int prev = batch_id;
  double mold[KMAX];
  if (KMAX > 1) {
    if (gl == 0)
      prev = __hip_atomic_exchange(d.claim + s, batch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) mold[k] = (k < K) ? d.M[k + (int64_t)s * K] : 0.0;
  }

  const double y0 = *reinterpret_cast<const double*>(base);
  const int nnz = *reinterpret_cast<const int*>(base + 8);
  const int ovf = *reinterpret_cast<const int*>(base + 12);
  const int cnt0 = nnz < d.rec_cap ? nnz : d.rec_cap;
  const bool has_tail = nnz > cnt0 || cnt0 > kGroup;   // overflow records or a wide main record

  double acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.0;

  // first (usually only) chunk of the row stays in registers for the scatter
  int64_t jf = -1;
  double vf = 0.0;
  if (gl < cnt0) {
    jf = reinterpret_cast<const int*>(base + 16)[gl];
    vf = reinterpret_cast<const double*>(base + d.rec_val_off)[gl];
    const double* wj = wsrc + jf * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) acc[k] += vf * wj[k];
  }
  if (has_tail) {
    row_tail_for_each(d, base, nnz, ovf, gl, [&](int64_t j, double v) {
      const double* wj = wsrc + j * K;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) acc[k] += v * wj[k];
    });
  }

  double lp[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) lp[k] = group_sum(acc[k]) + bk[k];

  int first;
  if (KMAX == 1) {
    double g0;
    if (d.family == SGDNET_BINOMIAL)
      g0 = 1.0 - y0 - 1.0 / (1.0 + exp(lp[0]));
    else
      g0 = lp[0] - y0;
    double gcv = 0.0;
    if (gl == 0) {
      const double old = __hip_atomic_exchange(m_slot(d, s), g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gcv = g0 - old;
    }
    gc[0] = __shfl(gcv, 0, kGroup);
    first = gc[0] != 0.0;
  } else {
    first = __shfl(prev != batch_id ? 1 : 0, 0, kGroup);
    if (first) {
      // gradient: every lane of the group computes the same K values
      double g[KMAX];
      if (d.family == SGDNET_MULTINOMIAL) {
        const double lse = log_sum_exp(lp, K);
        const unsigned cls = (unsigned)(y0 + 0.5);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          g[k] = 0.0;
          if (k < K) {
            g[k] = exp(lp[k] - lse);
            if ((unsigned)k == cls) g[k] -= 1.0;
          }
        }
      } else {
        const double* ys = d.y + (int64_t)s * d.Ky;   // mgaussian: Ky == K responses
#pragma unroll
        for (int k = 0; k < KMAX; ++k) g[k] = (k < K) ? lp[k] - ys[k] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          gc[k] = g[k] - mold[k];
          if (gl == (k & (kGroup - 1))) d.M[k + (int64_t)s * K] = g[k];
        }
      }
    }
  }

  if (first) {
    if (jf >= 0) {
      double* dj = Dt + jf * K;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K && gc[k] != 0.0) scatter_add<kLds>(dj + k, vf * gc[k]);
    }
    if (has_tail) {
      row_tail_for_each(d, base, nnz, ovf, gl, [&](int64_t j, double v) {
        double* dj = Dt + j * K;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (k < K && gc[k] != 0.0) scatter_add<kLds>(dj + k, v * gc[k]);
      });
    }
  }
  if (gl != 0) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) gc[k] = 0.0;   // count each draw once in the intercept sum
  }
}

// Intercept accumulator: one partial per block and class, summed by the sweep in a
// fixed order (thousands of same-address atomics would serialise at ~12 ns each).
template <int KMAX, int kThreads>
__device__ __forceinline__ void store_d0_partial(const SagaDev& d, int K, int batch_id,
                                                 const double (&gc)[KMAX]) {
  __shared__ double part[kThreads / 64][KMAX];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      const double tot = wave_sum(gc[k]);
      if ((threadIdx.x & 63) == 0) part[wave][k] = tot;
    }
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double tot = 0.0;
#pragma unroll
    for (int wv = 0; wv < kThreads / 64; ++wv) tot += part[wv][threadIdx.x];
    d0_publish(d, batch_id, threadIdx.x, tot);
  }
}

// --------------------------------------------------------------------------
// gather, global-scatter form: one draw per 16-lane group, every draw of the
// batch in flight at once (the kernel is a chain of dependent loads stream ->
// record -> w, so parallelism, not per-thread work, hides the HBM latency).
// Scattered fp64 atomics run at ~23 G requests/s chip-wide: 10 per draw at z = 10.
// --------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(kBlock) void saga_batch_gather_kernel(SagaDev d, const LamParams* lamp,
                                                                   int64_t t0_in_epoch, int m,
                                                                   int batch_id_offset) {
  const int K = KMAX == 1 ? 1 : d.K;
  const int gl = threadIdx.x & (kGroup - 1);
  const int i = (blockIdx.x * kBlock + threadIdx.x) / kGroup;
  const int64_t t0 = lamp->stream_base + t0_in_epoch;
  const int batch_id = lamp->batch_seq + batch_id_offset;
  double gc[KMAX], bk[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    gc[k] = 0.0;
    bk[k] = k < K ? d.b[k] - (d.standardize ? cw_sum(d, batch_id, k) : 0.0) : 0.0;
  }
  if (d.standardize) cw_clear_next(d, batch_id);
  if (i < m) saga_draw<KMAX, false>(d, d.stream[t0 + i], gl, batch_id, bk, d.D, gc, d.w);
  if (d.fit_intercept || d.standardize) store_d0_partial<KMAX, kBlock>(d, K, batch_id, gc);
}

// --------------------------------------------------------------------------
// K == 1 software pipeline for the LDS-privatised gather: a 16-lane group keeps 2*U
// draws in flight.  The source is ordered in phases (stream -> records -> x.w ->
// gradient -> gradient-memory exchange -> LDS scatter) with no atomic between the
// loads of a phase, so the round trips of every phase overlap.  A record's first
// `cap` slots are always readable (zero padded), so the idx/val loads do not wait
// for the header.
// --------------------------------------------------------------------------
#ifdef SGDNET_PHASE_TIMING
// development aid (never in the product build): shader-clock stamps after all outstanding
// memory operations of the wave have returned
__device__ __forceinline__ unsigned long long phase_stamp() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define PHASE(slot)                                                                      \
  do {                                                                                    \
    if (d.dbg && threadIdx.x == 0) d.dbg[(size_t)blockIdx.x * 16 + (slot)] = phase_stamp(); \
  } while (0)
#define PHASE_FIRST(slot)                                                                 \
  do {                                                                                    \
    if (d.dbg && threadIdx.x == 0 && stamp) d.dbg[(size_t)blockIdx.x * 16 + (slot)] = phase_stamp(); \
  } while (0)
#else
#define PHASE(slot) ((void)0)
#define PHASE_FIRST(slot) ((void)0)
#endif

// One pipeline stage set for U draws of a 16-lane group (K == 1).
// --------------------------------------------------------------------------
// K == 1, 8-lane form (records with rec_cap >= 16): an 8-lane group owns FOUR draws per pass
// and every lane holds TWO entries of each (one 8-byte index load, one 16-byte value load), so
// a wavefront carries 32 draws per pass -- twice the records in flight of the 16-lane form for
// the same number of load, exp and exchange instructions.  Lane 2q of the group owns draw q:
// it loads the draw's sample id and the record header (response, nnz, overflow link),
// evaluates the gradient and issues the gradient-memory exchange.  Entries 16.. of a row
// (2.7 % of the rows at 10 non-zeros per row) take the tail path below.
// Unused slots of a record are zero (both packers clear the records), so no per-entry count
// is needed: a zero value contributes nothing and is skipped by the scatter.
// --------------------------------------------------------------------------
constexpr int kLanes8 = 8;
constexpr int kInReg8 = 16;          // entries of a row held in registers

template <class F>
__device__ __forceinline__ void row_tail8(const SagaDev& d, const char* base, int nnz, int ovf, int gl, F f) {
  const int cap = d.rec_cap;
  const int cnt0 = nnz < cap ? nnz : cap;
  const int* ridx = reinterpret_cast<const int*>(base + 16);
  const double* rval = reinterpret_cast<const double*>(base + d.rec_val_off);
  for (int e = kInReg8 + gl; e < cnt0; e += kLanes8) f((int64_t)ridx[e], rval[e]);
  int rem = nnz - cnt0;
  while (rem > 0) {
    const char* ob = d.ovf + (size_t)ovf * kOvfStride;
    const int next = reinterpret_cast<const int*>(ob)[0];
    const int c = reinterpret_cast<const int*>(ob)[1];
    const int* oi = reinterpret_cast<const int*>(ob + 8);
    const double* ov = reinterpret_cast<const double*>(ob + 8 + 4 * kOvfCap);
    for (int e = gl; e < c; e += kLanes8) f((int64_t)oi[e], ov[e]);
    rem -= c;
    ovf = next;
  }
}

struct RecHeader {
  double y;
  int nnz;
  int ovf;
};

// All draws lo + g8 + k * (groups * 4) .. of one group; returns the sum of the gradient changes
// of the draws this lane owns.
template <int kThreads>
__device__ __forceinline__ double k1_lanes8_draws(const SagaDev& d, const uint32_t* sp, int lo, int hi, double b0,
                                                  const double* wv, double* Dl) {
  typedef int ipair_t __attribute__((ext_vector_type(2)));
  typedef double dpair_t __attribute__((ext_vector_type(2)));
  constexpr int U = 4;
  constexpr int kG = kThreads / kLanes8;        // groups per workgroup
  constexpr int kStep = kG * U;                 // draws per workgroup pass
  const int gl = threadIdx.x & (kLanes8 - 1);
  const int g8 = threadIdx.x / kLanes8;
  const int q = gl >> 1;                        // the draw of the pass this lane owns / holds x.w of
  const bool is_owner = (gl & 1) == 0;
  const size_t stride = (size_t)d.rec_stride;
  const int val_off = d.rec_val_off;
  double gct = 0.0;
  int i = lo + g8;
  if (i >= hi) return 0.0;
  // sample id of this lane's own draw; draws past the end stand in with draw i (discarded)
  uint32_t s_own = sp[i + q * kG < hi ? i + q * kG : i];
  for (; i < hi; i += kStep) {
    const bool v_own = i + q * kG < hi;
    uint32_t su[U];
#pragma unroll
    for (int u = 0; u < U; ++u) su[u] = (uint32_t)__shfl((int)s_own, 2 * u, kLanes8);
    const RecHeader hd = *reinterpret_cast<const RecHeader*>(d.rec + (size_t)s_own * stride);
    ipair_t jf[U];
    dpair_t vf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const char* base = d.rec + (size_t)su[u] * stride;
      jf[u] = *reinterpret_cast<const ipair_t*>(base + 16 + 8 * gl);
      vf[u] = *reinterpret_cast<const dpair_t*>(base + val_off + 16 * gl);
    }
    // ids of the next pass: requested before this pass's records are waited for
    const uint32_t s_this = s_own;
    {
      const int in = i + kStep;
      if (in < hi) s_own = sp[in + q * kG < hi ? in + q * kG : in];
    }
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = vf[u].x * wv[jf[u].x] + vf[u].y * wv[jf[u].y];
    const bool own_tail = v_own && hd.nnz > kInReg8;
    const bool any_tail = __ballot(own_tail) != 0;        // wave-uniform
    if (any_tail) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int nz = __shfl(own_tail ? hd.nnz : 0, 2 * u, kLanes8);
        if (nz > kInReg8) {
          const int ov = __shfl(hd.ovf, 2 * u, kLanes8);
          double a = 0.0;
          row_tail8(d, d.rec + (size_t)su[u] * stride, nz, ov, gl, [&](int64_t j, double v) { a += v * wv[j]; });
          acc[u] += a;
        }
      }
    }
    // merged butterfly: xor 4 halves four values to two, xor 2 to one, xor 1 finishes; draw q's
    // x.w ends up in lanes 2q, 2q+1 of the group
    const bool hi4 = (gl & 4) != 0, hi2 = (gl & 2) != 0;
    const double r0 = (hi4 ? acc[2] : acc[0]) + __shfl_xor(hi4 ? acc[0] : acc[2], 4, kLanes8);
    const double r1 = (hi4 ? acc[3] : acc[1]) + __shfl_xor(hi4 ? acc[1] : acc[3], 4, kLanes8);
    double t = (hi2 ? r1 : r0) + __shfl_xor(hi2 ? r0 : r1, 2, kLanes8);
    t += __shfl_xor(t, 1, kLanes8);
    const double lp = t + b0;
    const double g0 = d.family == SGDNET_BINOMIAL ? 1.0 - hd.y - 1.0 / (1.0 + exp(lp)) : lp - hd.y;
    double gcp = 0.0;
    if (is_owner && v_own) {
      // claim, read and update in ONE returning atomic: a repeated draw of the batch reads back
      // the value just stored (same snapshot, same g0), so its gc is exactly 0
      const double old = __hip_atomic_exchange(m_slot(d, s_this), g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gcp = g0 - old;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double gc = __shfl(gcp, 2 * u, kLanes8);
      if (gc != 0.0) {
        if (vf[u].x != 0.0) scatter_add<true>(Dl + jf[u].x, vf[u].x * gc);
        if (vf[u].y != 0.0) scatter_add<true>(Dl + jf[u].y, vf[u].y * gc);
        if (any_tail) {
          const int nz = __shfl(own_tail ? hd.nnz : 0, 2 * u, kLanes8);
          if (nz > kInReg8) {
            const int ov = __shfl(hd.ovf, 2 * u, kLanes8);
            row_tail8(d, d.rec + (size_t)su[u] * stride, nz, ov, gl,
                      [&](int64_t j, double v) { scatter_add<true>(Dl + j, v * gc); });
          }
        }
      }
    }
    gct += gcp;
  }
  return gct;
}

// --------------------------------------------------------------------------
// Compact records (K == 1, p <= 65536) with the gradient memory inside.
//
// Round 2's gather read a draw's record and then claimed / read / updated the sample's gradient memory with
// one returning device-scope exchange on a separate 80 MB table.  What the memory system charges for the
// candidates was measured without any compute (scripts/microbench/gather_patterns.hip, profiles/r03b_*,
// r03j_*; a fresh stream segment per repetition): random 128-B lines 24 us per 2^20 draws; line + dependent
// exchange on the separate table 54 us; line + an independent 8-byte load from a second table 45 us (ANY second
// random access costs what the line costs: the fabric serves ~45 G requests/s whatever their size); line + the
// same exchange aimed INTO the line just read 51.6 us; line + a plain 8-byte store into it 43-47 us.  So the
// gradient memory of sample s lives in the last 8 bytes of the sample's own line, and the 0/1 response of a
// binomial fit in a bit beside the long-row bit, which frees the room for a twelfth entry.
//
// The plain store needs to know beforehand which of a batch's repeated draws of a sample carries the change
// (the exchange decides it on the fly: a repeat reads back the value just stored).  That was built and
// measured -- stream_tag_kernel, a bitmap of the shard's samples in LDS, tagged draws `sample | first | long
// | y`: the gather came down to 62-63.5 us per launch alone (from 66), but marking first occurrences is
// 10M LDS atomics + 10M gathers per epoch on CUs that retire about one lane per clock of either: 185 us per
// epoch as one workgroup per shard and batch, 117 + 51 us as eight sub-range workgroups per batch with dense
// bit planes and a combine kernel, 139 us with three sub-ranges storing their words directly (partial-line
// writes), and hidden on the sample-order side stream it took CUs from the gather (62 -> 68 us per launch).
// The exchange INTO the record needs none of it and gives up ~2 us per launch: DESIGN.md 5 "Round 3".
//
//   plane P, 128 B per sample:  [ val[E] : 8 E | id[E] : 2 E (16-bit) | pad | y : 8 at 112 (E = 11) | M : 8 at 120 ]
//       E = 12 for binomial fits (the 0/1 response is a bit of cmeta), 11 otherwise
//   plane Q, 128 B per sample, touched only for rows with more than E entries (21 % / 30 % at 10 per row):
//       [ nnz : 4 | - : 4 | id[12] : 24 | val[12] : 96 ]   entries E .. E + 11
//   entries E + 12 .. of a row are read from the sample-major CSR arrays
//   cmeta, 2 bits per sample: bit 0 = the row has more than E entries, bit 1 = y != 0 (binomial); looked up
//       one pass ahead for sample ids requested two passes ahead, and carried in the id's spare high bits
//
// Lane mapping as before: lanes 0..5 of the 8-lane group hold P's entries 2 slot, 2 slot + 1, lanes 6..7 the
// first four of Q; entries E + 4 .. take the tail path.
// --------------------------------------------------------------------------
constexpr int kCQ = 12;               // entries in plane Q
constexpr int kCYOff = 112;           // response inside plane P (E = 11)
constexpr uint32_t kLongBit = 0x80000000u;
constexpr uint32_t kYBit = 0x20000000u;
constexpr uint32_t kIdMask = 0x1fffffffu;

__global__ __launch_bounds__(256) void pack_compact_kernel(const int64_t* ptr, const int32_t* idx,
                                                           const double* val, const double* y, int64_t n, int E,
                                                           int y_in_tag, char* P, char* Q, uint32_t* meta) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t q0 = ptr[i];
    const int nnz = (int)(ptr[i + 1] - q0);
    char* pb = P + (size_t)i * kCStride;
    double* pv = reinterpret_cast<double*>(pb);
    uint16_t* pid = reinterpret_cast<uint16_t*>(pb + 8 * E);
    for (int e = 0; e < E; ++e) pv[e] = e < nnz ? val[q0 + e] : 0.0;
    for (int e = 0; e < (kCMOff - 8 * E) / 2; ++e) pid[e] = e < E && e < nnz ? (uint16_t)idx[q0 + e] : (uint16_t)0;
    if (!y_in_tag) *reinterpret_cast<double*>(pb + kCYOff) = y[i];
    *reinterpret_cast<double*>(pb + kCMOff) = 0.0;
    uint32_t bits = 0u;
    if (nnz > E) {
      char* qb = Q + (size_t)i * kCStride;
      reinterpret_cast<int*>(qb)[0] = nnz;
      reinterpret_cast<int*>(qb)[1] = 0;
      uint16_t* qid = reinterpret_cast<uint16_t*>(qb + 8);
      double* qv = reinterpret_cast<double*>(qb + 32);
      for (int e = 0; e < kCQ; ++e) {
        qid[e] = E + e < nnz ? (uint16_t)idx[q0 + E + e] : (uint16_t)0;
        qv[e] = E + e < nnz ? val[q0 + E + e] : 0.0;
      }
      bits |= 1u;
    }
    if (y_in_tag && y[i] != 0.0) bits |= 2u;   // the response of a binomial fit rides in cmeta
    if (bits) atomicOr(meta + (i >> 4), bits << (2 * (i & 15)));
  }
}

// gradient memory between the K x n array and the records (a mode change, or the host reading / writing it)
__global__ __launch_bounds__(256) void m_move_kernel(SagaDev d, int to_record) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    double* r = reinterpret_cast<double*>(d.cP + (size_t)i * kCStride + kCMOff);
    if (to_record) *r = d.M[i];
    else d.M[i] = *r;
  }
}

// entries E + 4 .. of a long row
template <class F>
__device__ __forceinline__ void row_tail_compact(const SagaDev& d, uint32_t sid, int nnz, int gl, int E, F f) {
  const char* qb = d.cQ + (size_t)sid * kCStride;
  for (int e = E + 4 + gl; e < nnz; e += kLanes8) {
    if (e < E + kCQ) {
      f((int64_t) reinterpret_cast<const uint16_t*>(qb + 8)[e - E],
        reinterpret_cast<const double*>(qb + 32)[e - E]);
    } else {
      const int64_t q0 = d.ptr[sid];
      f((int64_t)d.idx[q0 + e], d.val[q0 + e]);
    }
  }
}

#ifndef SGDNET_LDS_BLOCK
#define SGDNET_LDS_BLOCK 1024
#endif
constexpr int kLdsBlock = SGDNET_LDS_BLOCK;

// Work distribution: a ticket is 32 consecutive draws (one wavefront pass: 8 groups x 4 draws,
// one 128-B line of the sample stream).  A workgroup owns a fixed range of the launch, and its
// 16 wavefronts draw tickets of that range from a counter in LDS: identical shares per
// wavefront left the workgroup waiting for its slowest wavefront (per-pass times vary by tens of
// per cent with the memory system's queues), and a workgroup's time is then the MAXIMUM of 16
// sums of 8 passes instead of their mean.  The LDS counter costs one ds_add_rtn per pass; it is
// used from 4 passes per wavefront (C4 with 8 shards: 0.83 -> 0.79..0.81 ms/epoch, with one shard
// and a single pass per wavefront it only adds latency: 2.08 -> 2.22).
// Tried and removed: handing the last 6-20 % of a launch out ACROSS workgroups from a per-shard
// counter in global memory.  Those tickets serialise on one L2 atomic unit (5-10 ns each) and
// every wavefront reserves three ahead: C4 with 8 shards 1256 -> 1141..1186 epochs/s, with 4
// shards 1016 -> 729; all tickets from a global counter: 160 us per 131 072-draw launch.
constexpr int kTicket = 32;           // draws per wavefront pass

struct TicketSource {
  int* counter;        // LDS, zeroed before the workgroup's barrier
  int lo, hi, m;       // the workgroup's range; m: end of the launch (sentinel)
  int t = 0;
  bool dynamic;
  __device__ __forceinline__ int next() {
    if (!dynamic) {                             // a pass or two per wavefront: nothing to balance
      const int b = lo + (t++ * (kLdsBlock / 64) + (int)(threadIdx.x >> 6)) * kTicket;
      return b < hi ? b : m;
    }
    int b = 0;
    if ((threadIdx.x & 63) == 0)
      b = __hip_atomic_fetch_add(counter, kTicket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    b = lo + __builtin_amdgcn_readfirstlane(b);
    return b < hi ? b : m;
  }
};

// The K == 1 compact gather of one workgroup over the draws sp[0, m) of its shard-batch.  The kernel runs it in
// three steps so that the first round trips of a workgroup overlap the staging of w into LDS instead of
// following it: begin() (static tickets for every wavefront's first two passes, their sample ids requested)
// before the staging, tag_first() (the first pass's cmeta bits) behind it, run() after the workgroup's barrier.
// (Also requesting the first pass's records before the barrier, with the loop's record loads moved to its end,
// carries one pass's registers across the back edge: 128 VGPRs and 112 bytes of scratch.)
struct K1Compact {
  typedef double dpair_t __attribute__((ext_vector_type(2)));
  static constexpr int U = 4;
  // lane geometry
  int E, in_reg, gl, g, q, slot, id_off, v_off;
  bool y_in_meta, is_owner, in_p, half;
  const char* plane;
  char* P;
  const uint32_t* sp;
  const uint32_t* meta;
  int m;
  TicketSource tk;
  int b_cur, b_nxt;
  uint32_t s_cur, s_nxt;      // s_cur: tagged (long row, response); s_nxt: as read from the stream

  __device__ __forceinline__ int own_pos(int base) const { return base + U * g + q < m ? base + U * g + q : base; }
  __device__ __forceinline__ uint32_t tagged(uint32_t sid) const {
    const uint32_t mb = (meta[sid >> 4] >> (2 * (sid & 15))) & 3u;
    return sid | ((mb & 1u) ? kLongBit : 0u) | ((mb & 2u) ? kYBit : 0u);
  }
  // tickets handed out before the LDS counter exists: two per wavefront (the counter starts behind them)
  static __device__ __forceinline__ int static_tickets() { return 2 * (kLdsBlock / 64) * kTicket; }

  // lane geometry, this wavefront's first two (static) tickets: no memory access
  __device__ __forceinline__ void init(const SagaDev& d, const uint32_t* sp_, int m_, int blk, int nblk,
                                       int* ticket_counter) {
    E = d.cE;                                     // entries in plane P: 12 (response in cmeta) or 11
    in_reg = E + 4;                               // entries of a row held in registers
    y_in_meta = E == 12;
    gl = threadIdx.x & (kLanes8 - 1);
    g = (threadIdx.x & 63) >> 3;                  // group inside the wavefront
    q = gl >> 1;
    is_owner = (gl & 1) == 0;
    in_p = gl < 6;                                // this lane's two entries come from plane P
    slot = in_p ? gl : gl - 6;
    P = d.cP;
    plane = in_p ? d.cP : d.cQ;
    id_off = in_p ? 8 * E + 4 * slot : 8 + 4 * slot;
    v_off = in_p ? 16 * slot : 32 + 16 * slot;
    half = in_p && slot == 5 && E == 11;          // entry 11 of plane P does not exist: the bytes are ids and pad
    sp = sp_;
    meta = d.cmeta;
    m = m_;
    const int share = ((m + nblk - 1) / nblk + kTicket - 1) / kTicket * kTicket;
    tk.counter = ticket_counter;
    tk.lo = blk * share;
    tk.hi = tk.lo + share < m ? tk.lo + share : m;
    tk.m = m;
    tk.dynamic = share >= 4 * (kLdsBlock / 64) * kTicket;
    tk.t = 2;
    const int wave = (int)(threadIdx.x >> 6);
    b_cur = tk.lo + wave * kTicket;
    b_nxt = tk.lo + ((kLdsBlock / 64) + wave) * kTicket;
    if (b_cur >= tk.hi) b_cur = m;
    if (b_nxt >= tk.hi) b_nxt = m;
  }
  // the sample ids of those two tickets (requested, not waited for)
  __device__ __forceinline__ void request_first() {
    s_cur = b_cur < m ? sp[own_pos(b_cur)] : 0u;
    s_nxt = b_nxt < m ? sp[own_pos(b_nxt)] : 0u;
  }
  __device__ __forceinline__ void begin(const SagaDev& d, const uint32_t* sp_, int m_, int blk, int nblk,
                                        int* ticket_counter) {
    init(d, sp_, m_, blk, nblk, ticket_counter);
    request_first();
  }

  __device__ __forceinline__ void tag_first() {
    if (b_cur < m) s_cur = tagged(s_cur);
  }

  // all passes of this wavefront; returns the sum of the gradient changes of the draws this lane owns
  __device__ __forceinline__ double run(const SagaDev& d, double b0, const double* wv, double* Dl) {
    double gct = 0.0;
    while (b_cur < m) {
      const bool v_own = b_cur + U * g + q < m;
      uint32_t su[U];
#pragma unroll
      for (int u = 0; u < U; ++u) su[u] = (uint32_t)__shfl((int)s_cur, 2 * u, kLanes8);
      const uint32_t s_this = s_cur & kIdMask;
      double y_own = (s_cur & kYBit) ? 1.0 : 0.0;
      if (!y_in_meta) y_own = *reinterpret_cast<const double*>(P + (size_t)s_this * kCStride + kCYOff);
      int nnz_own = 0;
      if (s_cur & kLongBit) nnz_own = *reinterpret_cast<const int*>(d.cQ + (size_t)s_this * kCStride);
      uint32_t jf[U];
      dpair_t vf[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool on = in_p || (su[u] & kLongBit) != 0;
        const char* base = plane + (size_t)(su[u] & kIdMask) * kCStride;
        jf[u] = 0u;
        vf[u] = dpair_t{0.0, 0.0};
        if (on) {
          jf[u] = *reinterpret_cast<const uint32_t*>(base + id_off);
          vf[u] = *reinterpret_cast<const dpair_t*>(base + v_off);
        }
      }
      // sample ids two passes ahead, their cmeta bits one pass ahead
      const int b_nn = b_nxt < m ? tk.next() : m;
      uint32_t s_nn = 0u;
      if (b_nn < m) s_nn = sp[own_pos(b_nn)];
      if (b_nxt < m) s_nxt = tagged(s_nxt);
      if (half) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          vf[u].y = 0.0;
          jf[u] &= 0xffffu;
        }
      }
      double acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = vf[u].x * wv[jf[u] & 0xffffu] + vf[u].y * wv[jf[u] >> 16];
      const bool own_tail = v_own && nnz_own > in_reg;
      const bool any_tail = __ballot(own_tail) != 0;
      if (any_tail) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int nz = __shfl(own_tail ? nnz_own : 0, 2 * u, kLanes8);
          if (nz > in_reg) {
            double a = 0.0;
            row_tail_compact(d, su[u] & kIdMask, nz, gl, E, [&](int64_t j, double v) { a += v * wv[j]; });
            acc[u] += a;
          }
        }
      }
      const bool hi4 = (gl & 4) != 0, hi2 = (gl & 2) != 0;
      const double r0 = (hi4 ? acc[2] : acc[0]) + __shfl_xor(hi4 ? acc[0] : acc[2], 4, kLanes8);
      const double r1 = (hi4 ? acc[3] : acc[1]) + __shfl_xor(hi4 ? acc[1] : acc[3], 4, kLanes8);
      double t = (hi2 ? r1 : r0) + __shfl_xor(hi2 ? r0 : r1, 2, kLanes8);
      t += __shfl_xor(t, 1, kLanes8);
      const double lp = t + b0;
      const double g0 = d.family == SGDNET_BINOMIAL ? 1.0 - y_own - 1.0 / (1.0 + exp(lp)) : lp - y_own;
      double gcp = 0.0;
      if (is_owner && v_own) {
        // claim, read and update in ONE returning atomic on the line the records came from: a repeated draw of
        // the batch reads back the value just stored (same snapshot, same g0), so its gc is exactly 0
        // (src/saga-sparse.h:281-282)
        const double old = __hip_atomic_exchange(reinterpret_cast<double*>(P + (size_t)s_this * kCStride + kCMOff), g0,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gcp = g0 - old;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double gc = __shfl(gcp, 2 * u, kLanes8);
        if (gc != 0.0) {
          if (vf[u].x != 0.0) scatter_add<true>(Dl + (jf[u] & 0xffffu), vf[u].x * gc);
          if (vf[u].y != 0.0) scatter_add<true>(Dl + (jf[u] >> 16), vf[u].y * gc);
          if (any_tail) {
            const int nz = __shfl(own_tail ? nnz_own : 0, 2 * u, kLanes8);
            if (nz > in_reg)
              row_tail_compact(d, su[u] & kIdMask, nz, gl, E,
                               [&](int64_t j, double v) { scatter_add<true>(Dl + j, v * gc); });
          }
        }
      }
      gct += gcp;
      s_cur = s_nxt;
      s_nxt = s_nn;
      b_cur = b_nxt;
      b_nxt = b_nn;
    }
    return gct;
  }
};

template <int U>
struct K1IdsOnly {
  uint32_t s[U];
  bool valid[U];
  uint32_t s_own;
  bool v_own;
  bool valid_any;
  __device__ __forceinline__ void load(const SagaDev& d, const uint32_t* sp, int i, int hi, int step, int gl,
                                       int safe) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int iu = i + u * step;
      valid[u] = iu < hi;
      s[u] = sp[valid[u] ? iu : safe];
    }
    const int q = U == 4 ? gl >> 2 : gl;
    const int iq = i + q * step;
    v_own = q < U && iq < hi;
    s_own = sp[v_own ? iq : safe];
  }
};

template <int U>
struct K1Draws {
  uint32_t s[U];
  int jf[U], nnz[U];
  double vf[U];
  double gcp;      // on lane owner(u) of the group: gradient change of draw u (0 on the other lanes)
  bool valid[U];
  // this lane's own draw q = draw_of(gl) (its sample and response come from this lane's own loads,
  // not from a selection among the U per-draw registers: such a selection is compiled into an
  // indexed lookup of a private-memory copy of the arrays)
  uint32_t s_own;
  double y_own;
  bool v_own;
  static __device__ __forceinline__ int draw_of(int gl) { return U == 4 ? gl >> 2 : gl; }

  // the lane of the group that evaluates draw u (see gradient())
  static __device__ __forceinline__ int owner(int u) { return U == 4 ? 4 * u : u; }
  static __device__ __forceinline__ bool is_owner(int gl) { return U == 4 ? (gl & 3) == 0 : gl < U; }

  // stream indices + record loads (nothing waits here)
  // `safe` < hi: the draw whose (discarded) record stands in for positions past the end
  __device__ __forceinline__ void load_ids(const SagaDev& d, const uint32_t* sp, int i, int hi, int step, int gl,
                                           int safe) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int iu = i + u * step;
      valid[u] = iu < hi;
      s[u] = sp[valid[u] ? iu : safe];
      if (SGD_ABLATE(d, 16)) s[u] &= 1023u;         // timing only: records from a cache-resident set
    }
    const int q = draw_of(gl);
    const int iq = i + q * step;
    v_own = q < U && iq < hi;
    s_own = sp[v_own ? iq : safe];
    if (SGD_ABLATE(d, 16)) s_own &= 1023u;
  }
  __device__ __forceinline__ void load_records(const SagaDev& d, int gl) {
    const int cap = d.rec_cap;
    y_own = *reinterpret_cast<const double*>(d.rec + (size_t)s_own * d.rec_stride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const char* base = d.rec + (size_t)s[u] * d.rec_stride;
      nnz[u] = *reinterpret_cast<const int*>(base + 8);
      jf[u] = gl < cap ? reinterpret_cast<const int*>(base + 16)[gl] : 0;
      vf[u] = gl < cap ? reinterpret_cast<const double*>(base + d.rec_val_off)[gl] : 0.0;
    }
  }
  __device__ __forceinline__ void load(const SagaDev& d, const uint32_t* sp, int i, int hi, int step, int gl,
                                       int safe) {
    load_ids(d, sp, i, hi, step, gl, safe);
    load_records(d, gl);
  }
  // the ids of another pass, taken over without touching this pass's records
  template <class O>
  __device__ __forceinline__ void take_ids(const O& o) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = o.s[u];
      valid[u] = o.valid[u];
    }
    s_own = o.s_own;
    v_own = o.v_own;
  }
  __device__ __forceinline__ bool in(const SagaDev& d, int u, int gl) const {
    const int cnt0 = nnz[u] < d.rec_cap ? nnz[u] : d.rec_cap;
    return valid[u] && gl < cnt0 && gl < kGroup;
  }
  __device__ __forceinline__ bool tail(const SagaDev& d, int u) const {
    const int cnt0 = nnz[u] < d.rec_cap ? nnz[u] : d.rec_cap;
    return valid[u] && (nnz[u] > cnt0 || cnt0 > kGroup);
  }
  template <class F>
  __device__ __forceinline__ void tail_for_each(const SagaDev& d, int u, int gl, F f) const {
    const char* base = d.rec + (size_t)s[u] * d.rec_stride;
    row_tail_for_each(d, base, nnz[u], *reinterpret_cast<const int*>(base + 12), gl, f);
  }
  // x.w, gradient, and the gradient-memory exchange (issued, not waited for)
  __device__ __forceinline__ void gradient(const SagaDev& d, int gl, double b0, const double* wv) {
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = in(d, u, gl) ? vf[u] * (SGD_ABLATE(d, 8) ? 1.0 : wv[jf[u]]) : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (tail(d, u)) {
        double a = 0.0;
        tail_for_each(d, u, gl, [&](int64_t j, double v) { a += v * wv[j]; });
        acc[u] += a;
      }
    }
    // One gradient evaluation for the U draws of the group.  The exp/division sequence is the
    // bulk of this kernel's vector instructions and costs the same whatever the 64 lanes hold,
    // so the U dot products are reduced TOGETHER: a merged butterfly (U = 4: xor 8 halves four
    // values to two, xor 4 to one, xor 2 and xor 1 finish: 5 shuffles instead of 16) that
    // leaves draw q's x.w in lanes 4q..4q+3 of the group.  Lane owner(q) then evaluates draw
    // q's gradient and issues its exchange: one evaluation and one atomic instruction per U draws.
    double lp_sel = 0.0;
    if (U == 4) {
      const bool hi8 = (gl & 8) != 0, hi4 = (gl & 4) != 0;
      const double r0 = (hi8 ? acc[2] : acc[0]) + __shfl_xor(hi8 ? acc[0] : acc[2], 8, kGroup);
      const double r1 = (hi8 ? acc[3] : acc[1]) + __shfl_xor(hi8 ? acc[1] : acc[3], 8, kGroup);
      double t = (hi4 ? r1 : r0) + __shfl_xor(hi4 ? r0 : r1, 4, kGroup);
      t += __shfl_xor(t, 2, kGroup);
      t += __shfl_xor(t, 1, kGroup);
      lp_sel = t + b0;
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double lp = group_sum(acc[u]) + b0;
        if (gl == owner(u)) lp_sel = lp;
      }
    }
    const double y_sel = y_own;
    const uint32_t s_sel = s_own;
    const bool v_sel = v_own;
    const double g0 = d.family == SGDNET_BINOMIAL ? 1.0 - y_sel - 1.0 / (1.0 + exp(lp_sel)) : lp_sel - y_sel;
    gcp = 0.0;
    if (is_owner(gl) && v_sel) {
      if (SGD_ABLATE(d, 1)) {                        // timing only: no gradient-memory exchange
        gcp = g0;
      } else {
        // claim, read and update in ONE returning atomic: a repeated draw of the batch reads
        // back the value just stored (same snapshot, same g0), so its gc is exactly 0
        const double old = __hip_atomic_exchange(m_slot(d, s_sel), g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gcp = g0 - old;
      }
    }
  }
  // LDS scatter of x * gc; returns the sum of gc (lane 0 of the group only)
  __device__ __forceinline__ double scatter(const SagaDev& d, int gl, double* Dl) const {
    double tot = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double gc = __shfl(gcp, owner(u), kGroup);
      if (gc != 0.0 && !SGD_ABLATE(d, 4)) {
        if (in(d, u, gl)) scatter_add<true>(Dl + jf[u], vf[u] * gc);
        if (tail(d, u)) tail_for_each(d, u, gl, [&](int64_t j, double v) { scatter_add<true>(Dl + j, v * gc); });
      }
    }
    tot = gcp;     // every draw counted once: on its owner lane
    return tot;
  }
};

// --------------------------------------------------------------------------
// gather, LDS-privatised scatter ("LDS staging of the gradient-average slice"):
// a 1024-thread workgroup owns `draws_per_block` consecutive draws and
// accumulates x*gc into a dense LDS copy of D (K*p doubles, ds_add_f64); at the
// end the copy is flushed with line-coalesced global atomics: lanes i..i+7 of a
// wave hit one 64-B line, so a workgroup issues at most K*p/8 atomic requests
// instead of one per non-zero.
// --------------------------------------------------------------------------

// kWLds (K == 1, 2*p doubles fit the CU's LDS): the coefficient snapshot is staged next to the
// accumulator, so the x.w gather -- 64 distinct addresses per wave instruction, which the
// vector-memory address unit serves at about one lane per clock -- becomes ds_read_b64.
// kVS (K == 1, kWLds): virtual shards -- the launch covers the same batch of d.V sample shards;
// workgroup b works for shard b / d.v_bps on that shard's replica of (w, b), its region of the
// sample stream and its own intercept partial.
template <int KMAX, bool kWLds = false, bool kVS = false, int kLanes = kGroup>
__global__ __launch_bounds__(kLdsBlock) void saga_batch_gather_lds_kernel(SagaDev d, const LamParams* lamp,
                                                                          int64_t t0_in_epoch, int m,
                                                                          int batch_id_offset,
                                                                          int draws_per_block) {
  extern __shared__ __attribute__((aligned(16))) double Dl[];
  const int K = KMAX == 1 ? 1 : d.K;
  const int64_t KP = (int64_t)K * d.p;
  __shared__ int ticket_counter;             // work tickets of the compact K == 1 form
  const int vsh = kVS ? (int)blockIdx.x / d.v_bps : 0;          // this workgroup's shard
  const int vblk = kVS ? (int)blockIdx.x - vsh * d.v_bps : (int)blockIdx.x;
  const double* w_src = kVS ? d.vw + (int64_t)vsh * KP : d.w;
  // compact K == 1 form: the sample ids of every wavefront's first two passes are requested before anything else
  constexpr bool kCompactForm = KMAX == 1 && kLanes == kLanes8;
  const bool compact = kCompactForm && d.cP != nullptr;
  K1Compact cg;
  if (threadIdx.x == 0) ticket_counter = compact ? K1Compact::static_tickets() : 0;
  if (compact)
    cg.begin(d, d.stream + lamp->stream_base + t0_in_epoch + (kVS ? (int64_t)vsh * d.v_dps : 0), m, vblk,
             kVS ? d.v_bps : (int)gridDim.x, &ticket_counter);
  PHASE(0);
  // the tables are moved as 16-byte pairs (half the instructions of a double-wise loop: the
  // kernel is bound by the instructions it issues as much as by memory); an odd last element
  // is handled by one thread
  typedef double pair_t __attribute__((ext_vector_type(2)));
  const int64_t KP2 = KP >> 1;
  {
    pair_t* D2 = reinterpret_cast<pair_t*>(Dl);
    for (int64_t i = threadIdx.x; i < KP2; i += kLdsBlock) D2[i] = pair_t{0.0, 0.0};
    if ((KP & 1) && threadIdx.x == 0) Dl[KP - 1] = 0.0;
  }
  if (kWLds) {
    // all loads of a thread in flight before its first LDS store (a plain copy loop waits for
    // every load in turn)
    constexpr int kStage = 8;                   // one round of loads for up to 16 384 coefficients
    double* Wl = Dl + KP + (KP & 1);            // 16-byte aligned
    const pair_t* w2 = reinterpret_cast<const pair_t*>(w_src);
    pair_t* W2 = reinterpret_cast<pair_t*>(Wl);
    for (int64_t i0 = threadIdx.x; i0 < KP2; i0 += (int64_t)kLdsBlock * kStage) {
      pair_t t[kStage];
#pragma unroll
      for (int r = 0; r < kStage; ++r) {
        const int64_t i = i0 + (int64_t)r * kLdsBlock;
        t[r] = i < KP2 ? w2[i] : pair_t{0.0, 0.0};
      }
#pragma unroll
      for (int r = 0; r < kStage; ++r) {
        const int64_t i = i0 + (int64_t)r * kLdsBlock;
        if (i < KP2) W2[i] = t[r];
      }
    }
    if ((KP & 1) && threadIdx.x == 0) Wl[KP - 1] = w_src[KP - 1];
  }
  if (compact) cg.tag_first();                 // the ids have arrived behind the staging loads
  __syncthreads();
  PHASE(1);

  const int gl = threadIdx.x & (kGroup - 1);
  const int group = threadIdx.x / kGroup;
  const int64_t t0 = lamp->stream_base + t0_in_epoch + (kVS ? (int64_t)vsh * d.v_dps : 0);
  const int batch_id = lamp->batch_seq + batch_id_offset;
  const int lo = vblk * draws_per_block;
  const int hi = (lo + draws_per_block < m) ? lo + draws_per_block : m;

  double gct[KMAX], bk[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    gct[k] = 0.0;
    bk[k] = k < K ? (kVS ? d.vb[vsh * K + k] - (d.standardize ? d.vcw[vsh * K + k] : 0.0)
                         : d.b[k] - (d.standardize ? cw_sum(d, batch_id, k) : 0.0))
                  : 0.0;
  }
  if (d.standardize && !kVS) cw_clear_next(d, batch_id);
  constexpr int kGroups = kLdsBlock / kGroup;
  if (KMAX == 1) {
    // K == 1.  8-lane form (both tables in LDS): compact records when the problem has them.
    // 16-lane form: one pass of U = 4 draws per group; the sample ids of the NEXT pass are
    // requested before this pass's records are waited for (a pass is a chain of dependent round
    // trips, ids -> records -> gradient-memory exchange, and this takes the first one off it).
    // (Two software-pipelined half-passes of 2 draws were 2 us slower once the gradient was
    // evaluated once per pass; records one pass ahead as well: DESIGN.md 5, item 9.)
    constexpr int U = 4;
    const double* wv = kWLds ? Dl + KP + (KP & 1) : d.w;
    const uint32_t* sp = d.stream + t0;
    if constexpr (kLanes == kLanes8) {
      if (compact)
        gct[0] = cg.run(d, bk[0], wv, Dl);
      else
        gct[0] = k1_lanes8_draws<kLdsBlock>(d, sp, lo, hi, bk[0], wv, Dl);
    } else if (lo + group < hi) {
      K1Draws<U> A;
      A.load_ids(d, sp, lo + group, hi, kGroups, gl, lo + group);
      for (int i = lo + group; i < hi; i += kGroups * U) {
        A.load_records(d, gl);
        K1IdsOnly<U> N;
        const int in = i + kGroups * U;
        N.valid_any = in < hi;
        if (N.valid_any) N.load(d, sp, in, hi, kGroups, gl, in);
        A.gradient(d, gl, bk[0], wv);
        gct[0] += A.scatter(d, gl, Dl);
        if (N.valid_any) A.take_ids(N);
      }
    }
  } else {
    for (int i = lo + group; i < hi; i += kGroups) {
      double gc[KMAX];
      saga_draw<KMAX, true>(d, d.stream[t0 + i], gl, batch_id, bk, Dl, gc, w_src);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) gct[k] += gc[k];
    }
  }
  PHASE(2);
  __syncthreads();
  PHASE(3);

  // flush the private copy as this workgroup's slab: plain coalesced stores (atomics would
  // cap the flush at the ~1.3 TB/s atomic rate); the sweep sums the slabs in a fixed order
  double* slab = d.slab + (int64_t)blockIdx.x * KP;
  if (!SGD_ABLATE(d, 2)) {
    if ((KP & 1) == 0) {                        // slabs start at multiples of KP doubles: pairs stay aligned
      const pair_t* D2 = reinterpret_cast<const pair_t*>(Dl);
      pair_t* S2 = reinterpret_cast<pair_t*>(slab);
      for (int64_t i = threadIdx.x; i < KP2; i += kLdsBlock) S2[i] = D2[i];
    } else {
      for (int64_t i = threadIdx.x; i < KP; i += kLdsBlock) slab[i] = Dl[i];
    }
  }
  PHASE(4);
  if (kVS) {                                    // one partial per workgroup and class, summed per shard by the sweep
    __shared__ double vpart[kLdsBlock / 64][KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const double t = wave_sum(gct[k]);
        if ((threadIdx.x & 63) == 0) vpart[threadIdx.x >> 6][k] = t;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < K) {
      double tot = 0.0;
      for (int wv = 0; wv < kLdsBlock / 64; ++wv) tot += vpart[wv][threadIdx.x];
      d.vd0[(int64_t)blockIdx.x * K + threadIdx.x] = tot;
    }
  } else if (d.fit_intercept || d.standardize) {
    store_d0_partial<KMAX, kLdsBlock>(d, K, batch_id, gct);
  }
  PHASE(5);
}

// --------------------------------------------------------------------------
// Dense x (src/saga-dense.h) in batched mode: one wavefront per draw.  A sample is p
// contiguous doubles, so the row streams through coalesced 512-byte wave loads; x.w is a
// wave reduction; x*gc goes into the workgroup's LDS copy of D with conflict-free ds_add_f64
// (lane l owns features l, l+64, ...), and the copy leaves as the workgroup's slab exactly as
// in the sparse LDS form, so the sweep kernels are shared.  The second pass over the row (the
// scatter) re-reads it from L1/L2.  Algorithmic bytes per draw: 8p (row) + 8Ky + 16K.
// --------------------------------------------------------------------------
constexpr int kDenseBlock = 256;

// kVS (K == 1): virtual shards as in the sparse LDS gather -- workgroup b works for shard
// b / d.v_bps on that shard's replica of (w, b) and its region of the sample stream.
// kTiled: K x p tables that fit no LDS.  The kernel stops after the gradient: the gradient change of
// draw i goes to d.gcb[i * K + k] (zero for a repeated sample) and saga_dense_tiled_accumulate_kernel
// forms D = X_batch^T gc feature tile by feature tile.
template <int KMAX, int kThreads = kDenseBlock, bool kVS = false, bool kTiled = false>
__global__ __launch_bounds__(kThreads) void saga_batch_gather_dense_kernel(SagaDev d, const LamParams* lamp,
                                                                           int64_t t0_in_epoch, int m,
                                                                           int batch_id_offset,
                                                                           int draws_per_block) {
  extern __shared__ __attribute__((aligned(16))) double Dl[];
  const int K = KMAX == 1 ? 1 : d.K;
  const int64_t p = d.p, KP = (int64_t)K * p;
  const int vsh = kVS ? (int)blockIdx.x / d.v_bps : 0;
  const int vblk = kVS ? (int)blockIdx.x - vsh * d.v_bps : (int)blockIdx.x;
  const double* w_src = kVS ? d.vw + (int64_t)vsh * KP : d.w;
  if (!kTiled) {
    for (int64_t i = threadIdx.x; i < KP; i += kThreads) Dl[i] = 0.0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t0 = lamp->stream_base + t0_in_epoch + (kVS ? (int64_t)vsh * d.v_dps : 0);
  const int batch_id = lamp->batch_seq + batch_id_offset;
  const int lo = vblk * draws_per_block;
  const int hi = (lo + draws_per_block < m) ? lo + draws_per_block : m;
  double bk[KMAX], gct[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    bk[k] = k < K ? (kVS ? d.vb[vsh * K + k] : d.b[k]) : 0.0;
    gct[k] = 0.0;
  }
  for (int i = lo + wave; i < hi; i += kThreads / 64) {
    const uint32_t s = d.stream[t0 + i];
    const double* xs = d.xd + (int64_t)s * p;
    int prev = batch_id;
    double mold[KMAX];
    if (KMAX > 1) {   // claim + old gradient memory: independent of the row
      if (lane == 0)
        prev = __hip_atomic_exchange(d.claim + s, batch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) mold[k] = k < K ? d.M[k + (int64_t)s * K] : 0.0;
    }
    double acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.0;
    // kRowU row chunks per lane requested before the first use (a plain strided loop waits for
    // every load in turn: the trip count is a run-time value)
    constexpr int kRowU = KMAX == 1 ? 8 : 4;
    for (int64_t j0 = lane; j0 < p; j0 += 64 * kRowU) {
      double xv[kRowU];
#pragma unroll
      for (int r = 0; r < kRowU; ++r) {
        const int64_t j = j0 + 64 * r;
        xv[r] = j < p ? xs[j] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < kRowU; ++r) {
        const int64_t j = j0 + 64 * r;
        if (j < p) {
          const double* wj = w_src + j * K;
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) acc[k] += xv[r] * wj[k];
        }
      }
    }
    double lp[KMAX], g[KMAX], gc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      lp[k] = wave_sum(acc[k]) + bk[k];
      gc[k] = 0.0;
    }
    bool first;
    if (KMAX == 1) {
      const double y0 = d.y[(int64_t)s * d.Ky];
      g[0] = d.family == SGDNET_BINOMIAL ? 1.0 - y0 - 1.0 / (1.0 + exp(lp[0])) : lp[0] - y0;
      double gcv = 0.0;
      if (lane == 0) {
        const double old = __hip_atomic_exchange(d.M + s, g[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gcv = g[0] - old;
      }
      gc[0] = __shfl(gcv, 0, 64);
      first = gc[0] != 0.0;
    } else {
      first = __shfl(prev != batch_id ? 1 : 0, 0, 64) != 0;
      if (first) {
        if (d.family == SGDNET_MULTINOMIAL) {
          const double lse = log_sum_exp(lp, K);
          const unsigned cls = (unsigned)(d.y[(int64_t)s * d.Ky] + 0.5);
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            g[k] = 0.0;
            if (k < K) {
              g[k] = exp(lp[k] - lse);
              if ((unsigned)k == cls) g[k] -= 1.0;
            }
          }
        } else if (d.family == SGDNET_MGAUSSIAN) {
          const double* ys = d.y + (int64_t)s * d.Ky;
#pragma unroll
          for (int k = 0; k < KMAX; ++k) g[k] = k < K ? lp[k] - ys[k] : 0.0;
        } else {   // not reached: single-response families have K == 1
#pragma unroll
          for (int k = 0; k < KMAX; ++k) g[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          if (k < K) {
            gc[k] = g[k] - mold[k];
            if (lane == k) d.M[k + (int64_t)s * K] = g[k];
          }
        }
      }
    }
    if (kTiled) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K && lane == k) d.gcb[(int64_t)i * K + k] = first ? gc[k] : 0.0;
    }
    if (first) {
      if (!kTiled) {
        for (int64_t j0 = lane; j0 < p; j0 += 64 * kRowU) {
          double xv[kRowU];
#pragma unroll
          for (int r = 0; r < kRowU; ++r) {
            const int64_t j = j0 + 64 * r;
            xv[r] = j < p ? xs[j] : 0.0;
          }
#pragma unroll
          for (int r = 0; r < kRowU; ++r) {
            const int64_t j = j0 + 64 * r;
            if (j < p) {
              double* dj = Dl + j * K;
#pragma unroll
              for (int k = 0; k < KMAX; ++k)
                if (k < K && gc[k] != 0.0) scatter_add<true>(dj + k, xv[r] * gc[k]);
            }
          }
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) gct[k] += gc[k];
      }
    }
  }
  if (!kTiled) {
    __syncthreads();
    double* slab = d.slab + (int64_t)blockIdx.x * KP;
    for (int64_t i = threadIdx.x; i < KP; i += kThreads) slab[i] = Dl[i];
  }
  if (kVS) {                                    // one partial per workgroup and class, summed per shard by the sweep
    __shared__ double vpart[kThreads / 64][KMAX];
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) vpart[wave][k] = gct[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < K) {
      double tot = 0.0;
      for (int wv = 0; wv < kThreads / 64; ++wv) tot += vpart[wv][threadIdx.x];
      d.vd0[(int64_t)blockIdx.x * K + threadIdx.x] = tot;
    }
  } else if (d.fit_intercept) {
    store_d0_partial<KMAX, kThreads>(d, K, batch_id, gct);
  }
}

// --------------------------------------------------------------------------
// Dense x, K x p beyond the LDS table: D = X_batch^T gc by feature tiles.  A workgroup owns 64
// consecutive features (lane <-> feature: every row segment is one 512-B read) and a chunk of the
// batch's draws (blockIdx.y); its four wavefronts take every fourth draw of the chunk, kTileU rows
// requested before the first is used, and meet in LDS in a fixed order.  One atomic add per
// (feature, class, chunk) into d.D -- chunks x K x p atomics per batch instead of m x K x p.
// Sample ids and gradient changes are wave-uniform (scalar loads); rows whose change is zero
// (repeated samples) are not read.
// --------------------------------------------------------------------------
constexpr int kTileF = 64;
constexpr int kTileU = 8;

template <int KMAX>
__global__ __launch_bounds__(kDenseBlock) void saga_dense_tiled_accumulate_kernel(SagaDev d, const LamParams* lamp,
                                                                                  int64_t t0_in_epoch, int m,
                                                                                  int draws_per_chunk) {
  __shared__ double part[kDenseBlock / 64 - 1][KMAX][kTileF];
  // more than KMAX classes (round 4, 17..64): blockIdx.z takes KMAX of them at a time; KS = the stride of a draw's
  // (and a feature's) class vector, K = the classes of this chunk
  const int KS = KMAX == 1 ? 1 : d.K;
  const int k0 = (int)blockIdx.z * KMAX;
  const int K = KS - k0 < KMAX ? KS - k0 : KMAX;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int kWaves = kDenseBlock / 64;
  const int64_t j = (int64_t)blockIdx.x * kTileF + lane;
  const bool live = j < d.p;
  const int64_t t0 = lamp->stream_base + t0_in_epoch;
  const int lo = (int)blockIdx.y * draws_per_chunk;
  const int hi = (lo + draws_per_chunk < m) ? lo + draws_per_chunk : m;
  double acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.0;
  for (int i0 = lo + wave; i0 < hi; i0 += kWaves * kTileU) {
    double xv[kTileU];
    bool on[kTileU];
#pragma unroll
    for (int u = 0; u < kTileU; ++u) {
      const int i = i0 + u * kWaves;
      on[u] = false;
      xv[u] = 0.0;
      if (i < hi) {
        const double* gci = d.gcb + (int64_t)i * KS + k0;
        bool any = false;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) any = any || (k < K && gci[k] != 0.0);
        on[u] = any;
        if (any && live) xv[u] = d.xd[(int64_t)d.stream[t0 + i] * d.p + j];
      }
    }
#pragma unroll
    for (int u = 0; u < kTileU; ++u) {
      if (on[u]) {
        const double* gci = d.gcb + (int64_t)(i0 + u * kWaves) * KS + k0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (k < K) acc[k] += xv[u] * gci[k];
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) part[wave - 1][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        double tot = acc[k];
#pragma unroll
        for (int wv = 0; wv < kWaves - 1; ++wv) tot += part[wv][k][lane];
        if (tot != 0.0) scatter_add<false>(d.D + j * KS + k0 + k, tot);
      }
    }
  }
}

// --------------------------------------------------------------------------
// Dense x with 17..64 classes (round 4; src/saga-dense.h:149-185 in batched form): the class-lane form.  A wavefront
// per draw, lane k = class k: the row arrives 64 features per load (coalesced), feature j's value is handed to all
// lanes through v_readlane and meets row j of w -- K contiguous doubles, one or a few 128-B lines from L2 -- so x.w
// needs no reduction across lanes and only the softmax does.  The kernel stops after the gradient, like the tiled
// form of fewer classes: the gradient change of draw i goes to d.gcb[i * K + k] (zero for a repeated sample),
// saga_dense_tiled_accumulate_kernel<16> forms D = X_batch^T gc sixteen classes at a time (blockIdx.z) and
// saga_dense_cl_sweep_kernel updates a feature's K coefficients per wavefront.
// Algorithmic bytes per draw: 8 p (row, twice: the accumulate pass reads it again) + 8 K p (w, from L2) + 24 K.
// --------------------------------------------------------------------------
__device__ __forceinline__ double lane_value(double v, int src) {   // src is wave-uniform
  const long long q = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(q & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(q >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__global__ __launch_bounds__(kDenseBlock) void saga_dense_cl_gather_kernel(SagaDev d, const LamParams* lamp,
                                                                           int64_t t0_in_epoch, int m,
                                                                           int batch_id_offset, int draws_per_block) {
  __shared__ double part[kDenseBlock / 64][64];
  const int K = d.K;
  const int64_t p = d.p;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool lane_on = lane < K;
  const int kl = lane_on ? lane : 0;                       // idle lanes read class 0 (addresses stay inside the arrays)
  const int64_t t0 = lamp->stream_base + t0_in_epoch;
  const int batch_id = lamp->batch_seq + batch_id_offset;
  const int lo = (int)blockIdx.x * draws_per_block;
  const int hi = (lo + draws_per_block < m) ? lo + draws_per_block : m;
  const double bk = d.b[kl];
  double gct = 0.0;
  for (int i = lo + wave; i < hi; i += kDenseBlock / 64) {
    const uint32_t s = d.stream[t0 + i];
    const double* xs = d.xd + (int64_t)s * p;
    int prev = batch_id;
    if (lane == 0) prev = __hip_atomic_exchange(d.claim + s, batch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double mold = d.M[kl + (int64_t)s * K];
    double acc = 0.0;
    for (int64_t j0 = 0; j0 < p; j0 += 64) {
      const double xv = j0 + lane < p ? xs[j0 + lane] : 0.0;
      const int cnt = p - j0 < 64 ? (int)(p - j0) : 64;
      const double* wr = d.w + j0 * K + kl;
      for (int e0 = 0; e0 < cnt; e0 += 8) {                // eight rows of w requested before the first is used
        double wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = e0 + u < cnt ? wr[(int64_t)(e0 + u) * K] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (e0 + u < cnt) acc += lane_value(xv, e0 + u) * wv[u];
      }
    }
    const double lp = acc + bk;
    double g;
    if (d.family == SGDNET_MULTINOMIAL) {
      const double mx = wave_max(lane_on ? lp : -HUGE_VAL);
      const double ssum = wave_sum(lane_on ? exp(lp - mx) : 0.0);
      const double lse = log(ssum) + mx;
      g = exp(lp - lse);
      if ((unsigned)lane == (unsigned)(d.y[(int64_t)s * d.Ky] + 0.5)) g -= 1.0;
    } else {                                               // mgaussian: Ky == K responses
      g = lp - d.y[(int64_t)s * d.Ky + kl];
    }
    const bool first = __shfl(prev != batch_id ? 1 : 0, 0, 64) != 0;
    double gc = 0.0;
    if (first && lane_on) {
      gc = g - mold;
      d.M[lane + (int64_t)s * K] = g;
    }
    if (lane_on) d.gcb[(int64_t)i * K + lane] = gc;
    gct += gc;
  }
  if (d.fit_intercept) {                                   // one partial per workgroup and class, summed by the sweep
    part[wave][lane] = gct;
    __syncthreads();
    if ((int)threadIdx.x < K) {
      double tot = 0.0;
#pragma unroll
      for (int wv = 0; wv < kDenseBlock / 64; ++wv) tot += part[wv][threadIdx.x];
      d0_publish(d, batch_id, threadIdx.x, tot);
    }
  }
}

// --------------------------------------------------------------------------
// Class-lane form for 4 < K <= 16 (multinomial / mgaussian with many classes): inside a
// 16-lane group lane l owns class l and the group walks the row's non-zeros together.  Every
// access to the K-fastest arrays (w, D, g_memory) is then K contiguous doubles per group =
// one or two 128-B requests, instead of K separate requests per non-zero, and x.w needs no
// cross-lane reduction (only the softmax does).
// --------------------------------------------------------------------------
__device__ __forceinline__ double group_max(double v) {
  v = fmax(v, __shfl_xor(v, 8, kGroup));
  v = fmax(v, __shfl_xor(v, 4, kGroup));
  v = fmax(v, __shfl_xor(v, 2, kGroup));
  v = fmax(v, __shfl_xor(v, 1, kGroup));
  return v;
}

// every lane of the group visits every non-zero (uniform addresses: broadcast loads)
template <class F>
__device__ __forceinline__ void row_for_each_uniform(const SagaDev& d, const char* base, int nnz, int ovf,
                                                     F f) {
  const int cap = d.rec_cap;
  const int cnt0 = nnz < cap ? nnz : cap;
  const int* ridx = reinterpret_cast<const int*>(base + 16);
  const double* rval = reinterpret_cast<const double*>(base + d.rec_val_off);
  for (int e = 0; e < cnt0; ++e) f((int64_t)ridx[e], rval[e]);
  int rem = nnz - cnt0;
  while (rem > 0) {
    const char* ob = d.ovf + (size_t)ovf * kOvfStride;
    const int next = reinterpret_cast<const int*>(ob)[0];
    const int c = reinterpret_cast<const int*>(ob)[1];
    const int* oi = reinterpret_cast<const int*>(ob + 8);
    const double* ov = reinterpret_cast<const double*>(ob + 8 + 4 * kOvfCap);
    for (int e = 0; e < c; ++e) f((int64_t)oi[e], ov[e]);
    rem -= c;
    ovf = next;
  }
}

// returns the gradient change of this lane's class (0 on lanes >= K and on repeated draws)
template <bool kLds>
__device__ __forceinline__ double saga_draw_classlane(const SagaDev& d, const uint32_t s, const int gl,
                                                      const int batch_id, const double bl, double* Dt,
                                                      const double* wsrc) {
  const int K = d.K;
  const bool lane_on = gl < K;
  const char* base = d.rec + (size_t)s * d.rec_stride;
  int prev = batch_id;
  if (gl == 0)
    prev = __hip_atomic_exchange(d.claim + s, batch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double mold = lane_on ? d.M[gl + (int64_t)s * K] : 0.0;
  const double y0 = *reinterpret_cast<const double*>(base);
  const int nnz = *reinterpret_cast<const int*>(base + 8);
  const int ovf = *reinterpret_cast<const int*>(base + 12);

  double acc = 0.0;
  row_for_each_uniform(d, base, nnz, ovf, [&](int64_t j, double v) {
    if (lane_on) acc += v * wsrc[j * K + gl];
  });
  const double lp = acc + bl;

  double g;
  if (d.family == SGDNET_MULTINOMIAL) {
    const double mx = group_max(lane_on ? lp : -HUGE_VAL);
    const double ssum = group_sum(lane_on ? exp(lp - mx) : 0.0);
    const double lse = log(ssum) + mx;
    g = exp(lp - lse);
    if ((unsigned)gl == (unsigned)(y0 + 0.5)) g -= 1.0;
  } else {
    g = lp - (lane_on ? d.y[(int64_t)s * d.Ky + gl] : 0.0);   // mgaussian: Ky == K responses
  }
  const int first = __shfl(prev != batch_id ? 1 : 0, 0, kGroup);
  double gc = 0.0;
  if (first && lane_on) {
    gc = g - mold;
    d.M[gl + (int64_t)s * K] = g;
  }
  if (first) {
    row_for_each_uniform(d, base, nnz, ovf, [&](int64_t j, double v) {
      if (lane_on && gc != 0.0) scatter_add<kLds>(Dt + j * K + gl, v * gc);
    });
  }
  return gc;
}

// kVS (kLds only; round 3): virtual shards as in saga_batch_gather_lds_kernel -- workgroup b works for shard
// b / d.v_bps on that shard's replica of (w, b), its region of the sample stream and its own intercept partials.
template <bool kLds, bool kVS = false>
__global__ __launch_bounds__(kLds ? kLdsBlock : kBlock) void saga_batch_gather_cl_kernel(
    SagaDev d, const LamParams* lamp, int64_t t0_in_epoch, int m, int batch_id_offset, int draws_per_block) {
  extern __shared__ __attribute__((aligned(16))) double Dl[];
  __shared__ double d0s[16];
  constexpr int kThreads = kLds ? kLdsBlock : kBlock;
  const int K = d.K;
  const int64_t KP = (int64_t)K * d.p;
  const int gl = threadIdx.x & (kGroup - 1);
  const int group = threadIdx.x / kGroup;
  const int vsh = kVS ? (int)blockIdx.x / d.v_bps : 0;
  const int vblk = kVS ? (int)blockIdx.x - vsh * d.v_bps : (int)blockIdx.x;
  const double* w_src = kVS ? d.vw + (int64_t)vsh * KP : d.w;
  const int64_t t0 = lamp->stream_base + t0_in_epoch + (kVS ? (int64_t)vsh * d.v_dps : 0);
  const int batch_id = lamp->batch_seq + batch_id_offset;
  if (kLds)
    for (int64_t i = threadIdx.x; i < KP; i += kThreads) Dl[i] = 0.0;
  if (threadIdx.x < 16) d0s[threadIdx.x] = 0.0;
  __syncthreads();
  double bl = 0.0;
  if (gl < K)
    bl = kVS ? d.vb[vsh * K + gl] - (d.standardize ? d.vcw[vsh * K + gl] : 0.0)
             : d.b[gl] - (d.standardize ? cw_sum(d, batch_id, gl) : 0.0);
  if (d.standardize && !kVS) cw_clear_next(d, batch_id);

  double gct = 0.0;
  if (kLds) {
    const int lo = vblk * draws_per_block;
    const int hi = (lo + draws_per_block < m) ? lo + draws_per_block : m;
    for (int i = lo + group; i < hi; i += kThreads / kGroup)
      gct += saga_draw_classlane<true>(d, d.stream[t0 + i], gl, batch_id, bl, Dl, w_src);
  } else {
    const int i = blockIdx.x * (kThreads / kGroup) + group;
    if (i < m) gct = saga_draw_classlane<false>(d, d.stream[t0 + i], gl, batch_id, bl, d.D, w_src);
  }
  if (gct != 0.0) __hip_atomic_fetch_add(&d0s[gl], gct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  if (kLds) {
    double* slab = d.slab + (int64_t)blockIdx.x * KP;
    for (int64_t i = threadIdx.x; i < KP; i += kThreads) slab[i] = Dl[i];
  }
  if (kVS) {
    if ((int)threadIdx.x < K) d.vd0[(int64_t)blockIdx.x * K + threadIdx.x] = d0s[threadIdx.x];
  } else if ((d.fit_intercept || d.standardize) && (int)threadIdx.x < K) {
    d0_publish(d, batch_id, threadIdx.x, d0s[threadIdx.x]);
  }
}

// --------------------------------------------------------------------------
// sweep: per feature (all K classes: GroupLasso needs the column norm)
//   w_j <- r^m w_j - gamma LS_m G_j - gamma D_j ; prox ; G_j += D_j / n
// --------------------------------------------------------------------------
struct SweepParams {
  int penalty;
  double gamma, beta, r_m, ls_m, m_d, n_d;
};

// Batch factors passed by value instead of read from LamParams (synchronous sharded mode: the
// draw count of a global batch varies by a few draws from round to round).  m <= 0: unused.
struct SweepOverride {
  double r_m, ls_m, m;
};

__device__ __forceinline__ SweepParams load_sweep_params(const SagaDev& d, const LamParams* lamp, int tail,
                                                         const SweepOverride& ov) {
  SweepParams q;
  q.penalty = lamp->penalty;
  q.gamma = lamp->gamma;
  q.beta = lamp->beta;
  q.r_m = tail ? lamp->r_tail : lamp->r_full;
  q.ls_m = tail ? lamp->ls_tail : lamp->ls_full;
  q.m_d = (double)(tail ? lamp->m_tail : lamp->m_full);
  if (ov.m > 0.0) {
    q.r_m = ov.r_m;
    q.ls_m = ov.ls_m;
    q.m_d = ov.m;
  }
  q.n_d = d.n_total;
  return q;
}

// dj: the K scatter sums of feature j (registers); wout receives the updated coefficients
__device__ __forceinline__ void sweep_feature(const SagaDev& d, const SweepParams& q, int64_t j,
                                              const double* dj, double* wout) {
  const int K = d.K;
  double* wj = d.w + j * K;
  double* gj = d.G + j * K;
  const double gls = q.gamma * q.ls_m;
  if (q.penalty == SGDNET_GROUPLASSO) {
    double nrm = 0.0;
    for (int k = 0; k < K; ++k) {
      const double v = q.r_m * wj[k] - gls * gj[k] - q.gamma * dj[k];
      wout[k] = v;
      nrm += v * v;
    }
    nrm = sqrt(nrm);
    const double factor = q.beta * q.gamma * q.ls_m / nrm;
    for (int k = 0; k < K; ++k) {
      wout[k] = factor < 1.0 ? wout[k] * (1.0 - factor) : 0.0;
      wj[k] = wout[k];
      gj[k] += dj[k] / q.n_d;
    }
  } else {
    const double tau = q.beta * q.gamma * q.ls_m;
    for (int k = 0; k < K; ++k) {
      const double dk = dj[k];
      double v = q.r_m * wj[k] - gls * gj[k] - q.gamma * dk;
      if (q.penalty == SGDNET_ELASTICNET) v = soft_threshold(v, tau);
      wj[k] = v;
      wout[k] = v;
      if (dk != 0.0) gj[k] += dk / q.n_d;
    }
  }
}

// d0[k] = sum of the gather kernel's per-block partials, in a fixed order, for every thread of
// the block (result in sh_d0).  Called by whole blocks.
template <int kThreads>
__device__ __forceinline__ void block_d0(const SagaDev& d, int n_parts, int batch_id, double* sh_d0) {
  __shared__ double red[kThreads / 64];
  const int K = d.K;
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    const double* set = d0_set(d, batch_id);
    for (int i = threadIdx.x; i < n_parts; i += kThreads) acc += set[(int64_t)i * K + k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int wv = 0; wv < kThreads / 64; ++wv) tot += red[wv];
      sh_d0[k] = tot;
    }
    __syncthreads();
  }
}

// intercept: gb += d0/n ; b -= gamma (0.01 m gb + d0/n)   (saga-sparse.h:300-304)
__device__ __forceinline__ void sweep_intercept(const SagaDev& d, const SweepParams& q, const double* sh_d0) {
  if ((int)threadIdx.x < d.K) {
    const int k = threadIdx.x;
    const double dk = sh_d0[k] / q.n_d;
    const double gbk = d.gb[k] + dk;
    d.gb[k] = gbk;
    // sparse x: the reference's intercept decay 0.01 (saga-sparse.h:300-304); dense x: none (saga-dense.h:170-173)
    d.b[k] -= q.gamma * (gbk * (d.xd ? 1.0 : 0.01) * q.m_d + dk);
  }
}

// adds this block's sum of c_j * w_new_kj into the next batch's c.w slots
template <int kThreads>
__device__ __forceinline__ void cw_accumulate(const SagaDev& d, int batch_id, const double* cwp) {
  __shared__ double red[kThreads / 64][64];
  const int K = d.K;
  for (int k = 0; k < K; ++k) {
    const double t = wave_sum(cwp[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    double tot = 0.0;
    for (int wv = 0; wv < kThreads / 64; ++wv) tot += red[wv][threadIdx.x];
    double* set = d.cw + (size_t)((batch_id + 1) & 1) * kCwSlots * K;
    if (tot != 0.0) atomic_add_f64(set + (blockIdx.x % kCwSlots) * K + threadIdx.x, tot);
  }
}

// D accumulated by global atomics (saga_batch_gather_kernel).  Ridge / ElasticNet act per
// element: one thread per (class, feature) entry, fully coalesced over the K-fastest arrays.
// GroupLasso needs the column norm: one thread per feature.
template <bool kGrouped>
__global__ __launch_bounds__(kBlock) void saga_batch_sweep_kernel(SagaDev d, LamParams* lamp, int tail,
                                                                  int n_parts, int batch_id_offset,
                                                                  SweepOverride ov) {
  __shared__ double sh_d0[16];
  const SweepParams q = load_sweep_params(d, lamp, tail, ov);
  const int K = d.K;
  const bool need_d0 = d.standardize || (blockIdx.x == 0 && d.fit_intercept);
  const int batch_id = lamp->batch_seq + batch_id_offset;
  if (need_d0) block_d0<kBlock>(d, n_parts, batch_id, sh_d0);
  double cwp[16];
  for (int k = 0; k < K; ++k) cwp[k] = 0.0;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (kGrouped) {
    if (t < d.p) {
      double* dg = d.D + t * K;
      double dj[16], wn[16];
      const double cj = d.standardize ? d.c[t] : 0.0;
      for (int k = 0; k < K; ++k) {
        dj[k] = dg[k] - (d.standardize ? cj * sh_d0[k] : 0.0);
        dg[k] = 0.0;
      }
      sweep_feature(d, q, t, dj, wn);
      for (int k = 0; k < K; ++k) cwp[k] = cj * wn[k];
    }
  } else if (t < (int64_t)K * d.p) {
    const int64_t j = t / K;
    const int k = (int)(t - j * K);
    const double cj = d.standardize ? d.c[j] : 0.0;
    const double raw = d.D[t];
    const double dk = raw - (d.standardize ? cj * sh_d0[k] : 0.0);
    double v = q.r_m * d.w[t] - q.gamma * q.ls_m * d.G[t] - q.gamma * dk;
    if (q.penalty == SGDNET_ELASTICNET) v = soft_threshold(v, q.beta * q.gamma * q.ls_m);
    d.w[t] = v;
    if (dk != 0.0) d.G[t] += dk / q.n_d;
    if (raw != 0.0) d.D[t] = 0.0;
    // every lane keeps its own class slot so that cw_accumulate's wave_sum stays per class
    for (int kk = 0; kk < K; ++kk) cwp[kk] = kk == k ? cj * v : 0.0;
  }
  if (d.standardize) cw_accumulate<kBlock>(d, batch_id, cwp);
  if (blockIdx.x == 0) {
    if (d.fit_intercept) sweep_intercept(d, q, sh_d0);
    double* nxt = d0_set(d, batch_id + 1);      // the next gather may add into it atomically
    for (int i = threadIdx.x; i < kD0Slots * K; i += kBlock) nxt[i] = 0.0;
  }
}

// Dense class-lane form (17..64 classes, saga_dense_cl_gather_kernel): a wavefront per feature, lane k = class k --
// D_j, w_j and G_j are K contiguous doubles each, the group norm is a wavefront sum.  Dense x is standardised
// explicitly, so there is no implicit centring here.
__global__ __launch_bounds__(kBlock) void saga_dense_cl_sweep_kernel(SagaDev d, LamParams* lamp, int tail, int n_parts,
                                                                     int batch_id_offset) {
  __shared__ double sh_d0[64];
  const SweepParams q = load_sweep_params(d, lamp, tail, SweepOverride{0.0, 0.0, 0.0});
  const int K = d.K;
  const int batch_id = lamp->batch_seq + batch_id_offset;
  if (blockIdx.x == 0 && d.fit_intercept) block_d0<kBlock>(d, n_parts, batch_id, sh_d0);
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const bool on = j < d.p && lane < K;
  const int64_t t = on ? j * K + lane : 0;
  const double raw = on ? d.D[t] : 0.0, w_old = on ? d.w[t] : 0.0, g_old = on ? d.G[t] : 0.0;
  double v = on ? q.r_m * w_old - q.gamma * q.ls_m * g_old - q.gamma * raw : 0.0;
  const double tau = q.beta * q.gamma * q.ls_m;
  if (q.penalty == SGDNET_GROUPLASSO) {                    // penalties.h:61-79
    const double factor = tau / sqrt(wave_sum(v * v));
    v = factor < 1.0 ? v * (1.0 - factor) : 0.0;
  } else if (q.penalty == SGDNET_ELASTICNET) {
    v = soft_threshold(v, tau);
  }
  if (on) {
    d.w[t] = v;
    if (raw != 0.0) {
      d.G[t] = g_old + raw / q.n_d;
      d.D[t] = 0.0;
    }
  }
  if (blockIdx.x == 0) {
    if (d.fit_intercept) sweep_intercept(d, q, sh_d0);
    double* nxt = d0_set(d, batch_id + 1);                 // the next gather may add into it atomically
    for (int i = threadIdx.x; i < kD0Slots * K; i += kBlock) nxt[i] = 0.0;
  }
}

// D held as per-workgroup slabs (saga_batch_gather_lds_kernel): a block owns F = 32/K
// features; 8 thread groups each sum an eighth of the slabs (coalesced over the features),
// the partial sums meet in LDS in a fixed order, then one thread per feature updates.
constexpr int kSlabElems = 32;
constexpr int kSlabGroups = kBlock / kSlabElems;

__global__ __launch_bounds__(kBlock) void saga_batch_sweep_slab_kernel(SagaDev d, LamParams* lamp, int tail,
                                                                       int n_parts, int batch_id_offset) {
  __shared__ double part[kSlabGroups][kSlabElems];
  __shared__ double sh_d0[16];
  const SweepParams q = load_sweep_params(d, lamp, tail, SweepOverride{0.0, 0.0, 0.0});
  const int K = d.K;
  const bool need_d0 = d.standardize || (blockIdx.x == 0 && d.fit_intercept);
  const int batch_id = lamp->batch_seq + batch_id_offset;
  if (need_d0) block_d0<kBlock>(d, n_parts, batch_id, sh_d0);
  const int F = kSlabElems / K;              // K <= 16
  const int E = F * K;
  const int64_t KP = (int64_t)K * d.p;
  const int e = threadIdx.x % kSlabElems, g = threadIdx.x / kSlabElems;
  const int64_t elem = (int64_t)blockIdx.x * E + e;
  double acc = 0.0;
  if (e < E && elem < KP) {
    const double* sp = d.slab + elem;
    int bidx = g;
    for (; bidx + 3 * kSlabGroups < n_parts; bidx += 4 * kSlabGroups) {   // 4 loads in flight
      const double a0 = sp[(int64_t)bidx * KP], a1 = sp[(int64_t)(bidx + kSlabGroups) * KP];
      const double a2 = sp[(int64_t)(bidx + 2 * kSlabGroups) * KP];
      const double a3 = sp[(int64_t)(bidx + 3 * kSlabGroups) * KP];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; bidx < n_parts; bidx += kSlabGroups) acc += sp[(int64_t)bidx * KP];
  }
  part[g][e] = acc;
  __syncthreads();
  double cwp[16];
  for (int k = 0; k < K; ++k) cwp[k] = 0.0;
  if ((int)threadIdx.x < F) {
    const int64_t j = (int64_t)blockIdx.x * F + threadIdx.x;
    if (j < d.p) {
      double dj[16], wn[16];
      const double cj = d.standardize ? d.c[j] : 0.0;
      for (int k = 0; k < K; ++k) {
        const int ee = threadIdx.x * K + k;
        double t = 0.0;
        for (int gg = 0; gg < kSlabGroups; ++gg) t += part[gg][ee];
        dj[k] = t - (d.standardize ? cj * sh_d0[k] : 0.0);
      }
      sweep_feature(d, q, j, dj, wn);
      for (int k = 0; k < K; ++k) cwp[k] = cj * wn[k];
    }
  }
  if (d.standardize) cw_accumulate<kBlock>(d, batch_id, cwp);
  if (blockIdx.x == 0) {
    if (d.fit_intercept) sweep_intercept(d, q, sh_d0);
    double* nxt = d0_set(d, batch_id + 1);      // the next gather may add into it atomically
    for (int i = threadIdx.x; i < kD0Slots * K; i += kBlock) nxt[i] = 0.0;
  }
}

// c.w of the current w into the slot set batch `batch_id` will read; clears the other set.
__global__ __launch_bounds__(kBlock) void saga_cw_init_kernel(SagaDev d, const LamParams* lamp) {
  __shared__ double red[kBlock / 64];
  const int K = d.K;
  const int batch_id = lamp->batch_seq;
  for (int i = threadIdx.x; i < 2 * kCwSlots * K; i += kBlock) d.cw[i] = 0.0;
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < d.p; j += kBlock) acc += d.c[j] * d.w[k + j * K];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int wv = 0; wv < kBlock / 64; ++wv) tot += red[wv];
      d.cw[(size_t)(batch_id & 1) * kCwSlots * K + k] = tot;
    }
    __syncthreads();
  }
}

// Advances the epoch bookkeeping that graph replays read.
// --------------------------------------------------------------------------
// Virtual shards (K == 1): sweep of all V replicas in one launch.  Block b serves shard
// b / nfb and the 32 features (b % nfb) * 32 ...; it sums that shard's v_bps slabs in a fixed
// order, updates the shard's replica of (w, g_sum) with the shard's own normalisation, and the
// first block of every shard updates the shard's intercept pair.
// --------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(kBlock) void saga_vs_sweep_kernel(SagaDev d, LamParams* lamp, int tail, int nfb) {
  __shared__ double part[kSlabGroups][kSlabElems];
  __shared__ double red[kBlock / 64];
  __shared__ double sh_d0[KMAX];
  const SweepParams q = load_sweep_params(d, lamp, tail, SweepOverride{0.0, 0.0, 0.0});
  const int K = KMAX == 1 ? 1 : d.K;
  const int F = kSlabElems / K;              // features per block (K <= 16)
  const int E = F * K;
  const int v = (int)blockIdx.x / nfb, fb = (int)blockIdx.x - v * nfb;
  const int64_t KP = (int64_t)K * d.p;
  const double n_d = d.v_size[v];
  const int e = threadIdx.x % kSlabElems, g = threadIdx.x / kSlabElems;
  const int64_t elem = (int64_t)fb * E + e;
  // the updating threads' own coefficients and gradient averages: requested with the slabs, not behind them
  const int64_t j_own = (int64_t)fb * F + threadIdx.x;
  const bool updates = (int)threadIdx.x < F && j_own < d.p;
  double w_old[KMAX], g_old[KMAX], c_own = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    w_old[k] = g_old[k] = 0.0;
    if (updates && k < K) {
      w_old[k] = d.vw[(int64_t)v * KP + j_own * K + k];
      g_old[k] = d.vG[(int64_t)v * KP + j_own * K + k];
    }
  }
  if (updates && d.standardize) c_own = d.c[j_own];
  double acc = 0.0;
  if (e < E && elem < KP) {
    const double* sp = d.slab + (int64_t)v * d.v_bps * KP + elem;
    int bidx = g;
    for (; bidx + 3 * kSlabGroups < d.v_bps; bidx += 4 * kSlabGroups) {   // 4 loads in flight
      const double a0 = sp[(int64_t)bidx * KP], a1 = sp[(int64_t)(bidx + kSlabGroups) * KP];
      const double a2 = sp[(int64_t)(bidx + 2 * kSlabGroups) * KP];
      const double a3 = sp[(int64_t)(bidx + 3 * kSlabGroups) * KP];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; bidx < d.v_bps; bidx += kSlabGroups) acc += sp[(int64_t)bidx * KP];
  }
  part[g][e] = acc;
  const bool need_d0 = fb == 0 || d.standardize;  // the shard's intercept accumulator = sum of gc, per class
  if (need_d0) {
    for (int k = 0; k < K; ++k) {
      double a = 0.0;
      for (int i = threadIdx.x; i < d.v_bps; i += kBlock) a += d.vd0[(int64_t)(v * d.v_bps + i) * K + k];
      a = wave_sum(a);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
        for (int wv = 0; wv < kBlock / 64; ++wv) t += red[wv];
        sh_d0[k] = t;
      }
      __syncthreads();
    }
  } else {
    __syncthreads();
  }
  if (updates) {
    double dj[KMAX], val[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      dj[k] = 0.0;
      if (k < K) {
        const int ee = (int)threadIdx.x * K + k;
        double t = 0.0;
        for (int gg = 0; gg < kSlabGroups; ++gg) t += part[gg][ee];
        dj[k] = t - (d.standardize ? c_own * sh_d0[k] : 0.0);   // implicit centring: D_j -= c_j * sum(gc)
      }
    }
    const double gls = q.gamma * q.ls_m;
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      val[k] = k < K ? q.r_m * w_old[k] - gls * g_old[k] - q.gamma * dj[k] : 0.0;
      nrm += val[k] * val[k];
    }
    const double tau = q.beta * q.gamma * q.ls_m;
    const double factor = q.penalty == SGDNET_GROUPLASSO ? tau / sqrt(nrm) : 0.0;   // penalties.h:61-79
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        double out = val[k];
        if (q.penalty == SGDNET_ELASTICNET) out = soft_threshold(out, tau);
        else if (q.penalty == SGDNET_GROUPLASSO) out = factor < 1.0 ? out * (1.0 - factor) : 0.0;
        d.vw[(int64_t)v * KP + j_own * K + k] = out;
        if (dj[k] != 0.0 || q.penalty == SGDNET_GROUPLASSO)
          d.vG[(int64_t)v * KP + j_own * K + k] = g_old[k] + dj[k] / n_d;
      }
    }
  }
  if (fb == 0 && (int)threadIdx.x < K && d.fit_intercept) {   // saga-sparse.h:300-304, batched form
    const int k = threadIdx.x;
    const double dk = sh_d0[k] / n_d;
    const double gbk = d.vgb[v * K + k] + dk;
    d.vgb[v * K + k] = gbk;
    d.vb[v * K + k] -= q.gamma * (gbk * (d.xd ? 1.0 : 0.01) * q.m_d + dk);
  }
}

// c . w of every replica and class (implicit centring)
__global__ __launch_bounds__(kBlock) void saga_vs_cw_kernel(SagaDev d) {
  __shared__ double red[kBlock / 64];
  const int K = d.K;
  const int v = (int)blockIdx.x / K, k = (int)blockIdx.x - v * K;
  const double* wv = d.vw + (int64_t)v * K * d.p + k;
  double a = 0.0;
  for (int64_t j = threadIdx.x; j < d.p; j += kBlock) a += d.c[j] * wv[j * K];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int wvi = 0; wvi < kBlock / 64; ++wvi) t += red[wvi];
    d.vcw[blockIdx.x] = t;
  }
}

// replicated state, flattened: [g_sum (K p) | w (K p) | g_sum_intercept (K) | intercept (K)]
__device__ __forceinline__ double* vs_slot(const SagaDev& d, int v, int64_t i, int64_t KP, int K) {
  if (i < KP) return d.vG + (int64_t)v * KP + i;
  if (i < 2 * KP) return d.vw + (int64_t)v * KP + (i - KP);
  if (i < 2 * KP + K) return d.vgb + (int64_t)v * K + (i - 2 * KP);
  return d.vb + (int64_t)v * K + (i - 2 * KP - K);
}
__device__ __forceinline__ double* vs_own_slot(const SagaDev& d, int64_t i, int64_t KP, int K) {
  if (i < KP) return d.G + i;
  if (i < 2 * KP) return d.w + (i - KP);
  if (i < 2 * KP + K) return d.gb + (i - 2 * KP);
  return d.b + (i - 2 * KP - K);
}

// every replica (and the snapshot) <- the solver's current (w, g_sum, b, g_sum_b)
__global__ __launch_bounds__(kBlock) void saga_vs_broadcast_kernel(SagaDev d) {
  const int K = d.K;
  const int64_t KP = (int64_t)K * d.p, len = 2 * KP + 2 * K;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    const double val = *vs_own_slot(d, i, KP, K);
    d.vref[i] = val;
    for (int v = 0; v < d.V; ++v) *vs_slot(d, v, i, KP, K) = val;
  }
}

// periodic average: every replica (and the snapshot) <- snapshot + sum_v (size_v / n) (replica_v - snapshot);
// final_merge also stores the result as the solver's state
// epoch_end != nullptr (the epoch's last merge): also the epoch's bookkeeping (saga_epoch_end_kernel), one launch less
__global__ __launch_bounds__(kBlock) void saga_vs_merge_kernel(SagaDev d, int final_merge, LamParams* epoch_end,
                                                               int batches) {
  if (epoch_end && blockIdx.x == 0 && threadIdx.x == 0) end_epoch(epoch_end, batches);
  const int K = d.K;
  const int64_t KP = (int64_t)K * d.p, len = 2 * KP + 2 * K;
  double tot_size = 0.0;
  for (int v = 0; v < d.V; ++v) tot_size += d.v_size[v];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    const double ref = d.vref[i];
    double val = ref;
    for (int v = 0; v < d.V; ++v) val += (d.v_size[v] / tot_size) * (*vs_slot(d, v, i, KP, K) - ref);
    d.vref[i] = val;
    for (int v = 0; v < d.V; ++v) *vs_slot(d, v, i, KP, K) = val;
    if (final_merge) *vs_own_slot(d, i, KP, K) = val;
  }
}

__global__ void saga_epoch_end_kernel(LamParams* lamp, int batches) {
  if (threadIdx.x == 0 && blockIdx.x == 0) end_epoch(lamp, batches);
}

// --------------------------------------------------------------------------
// Fused epoch of the virtual shards (round 4): ONE launch per epoch instead of
// broadcast + 10 x (gather, sweep) + 5 merges.  Sparse x, one response, compact records.
//
// Between two merges the V shards are independent chains gather -> sweep -> gather ...
// (src/saga-sparse.h:258-337 in batches, per replica), so nothing but a merge needs the whole
// grid: the S workgroups of a shard synchronise among themselves (two counters per shard), and a
// merge needs only the V workgroups that own the same feature slice (one counter per slice).  The
// shards therefore drift apart in time: while one shard's workgroups publish slabs, wait, sweep
// and restage w -- all latency -- the other shards' draw loops keep the memory system busy.  With
// one launch per batch every workgroup of the chip went through those phases at the same moment
// (26 % of a C4 epoch was outside the draw loops: VERDICT round 3).
//
// Round r of workgroup (v, i) -- shard v, slice i of S:
//   stage    W (LDS) <- replica w_v ; b0 <- b_v [- c.w_v]                  (round 0: the solver's own state)
//   draws    its share of the shard's batch against (W, b0): K1Compact, D_i in LDS   (saga-sparse.h:274-282, 306-335)
//   publish  D_i -> slab (v, i), sum of gc -> vd0 ; arrive cnt1[v] ; wait for all S
//   sweep    features [i F, (i+1) F): sum of the S slabs in a fixed order, w_j <- r^m w_j - gamma LS_m G_j
//            - gamma D_j ; prox ; G_j += D_j / n_v (:316-325, 340-348 batched; penalties.h); (i == 0) intercept (:300-304)
//   merge    (every `every` rounds and at the end) slice -> pub[parity][v] ; arrive col[i] ; wait for all V ;
//            slice <- ref + sum_u (n_u / n) (pub_u - ref), the same expression in all V workgroups
//   arrive cnt2[v] ; wait for all S (next round's staging reads every slice of w_v)
//
// Visibility follows the write-through form of the CDNA4 guide (inter-workgroup hand-off, form R1): EVERY byte
// another workgroup reads is stored with sc1 (buffer_store ... sc1 / agent-scope atomic store), every storing wave
// drains vmcnt before the workgroup's barrier, ONE lane then adds to the counter (agent-scope atomic), ONE lane polls
// it with sc1 loads, the other waves wait at a workgroup barrier, and EVERY load of handed-off bytes is an sc1 load
// (no L1 hit on a stale line).  Nothing depends on which XCD a workgroup runs on.
//
// Residency: the workgroups spin on each other, so all of them must be resident at once (one per CU: the LDS
// tables fill it).  The kernel does not assume that: it opens with a start barrier that ONE arbiter word decides
// (compare-and-swap 0 -> 1 "go" by workgroup 0 once everybody has arrived, 0 -> 2 "abort" by whoever waited too
// long); before that decision nothing is modified, so an aborted launch (a GPU shared with another process, fewer
// CUs than the grid) leaves the solver's state as it was, LamParams::fused_abort = 1 tells the host, and the epoch is
// run again as separate launches.  Every later wait is bounded too (fused_abort = 2: a bug, the epoch is void).
// The last workgroup to leave zeroes the counters for the next launch.
// --------------------------------------------------------------------------
constexpr int kSyncLine = 32;                  // unsigned words per 128-B line: every polled word has a line of its own
constexpr int kSyncGo = 0, kSyncExit = 1, kSyncStart = 2, kSyncCnt1 = 3, kSyncCnt2 = kSyncCnt1 + 8,
              kSyncXcd = kSyncCnt2 + 8;
constexpr int kFusedMaxBps = 128;              // workgroups per shard
constexpr int kSyncLines = kSyncXcd + 8;         // (the slice counters col[i] of the merges have an array of their own: SagaDev::vcol)
constexpr int kSyncSticky = kSyncLines;        // abort code of any launch since the host last looked (never reset on the device)
constexpr int kSyncSeq = kSyncLines + 1;       // linked solvers: launches since they were linked (their slice counters run on)
constexpr int kFusedChunks = 3;                // 64-lane chunks of 16-byte pairs in a workgroup's feature slice
constexpr long long kFusedStartTicks = 2000000;      // 20 ms of the 100 MHz wall clock: the start barrier
constexpr long long kFusedWaitTicks = 200000000;     // 2 s: every later wait (never reached unless there is a bug)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fused_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// 16-byte write-through store / L1-bypassing load (aux 16 = sc1)
__device__ __forceinline__ void st2_sc1(f64x2_t x, __amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ f64x2_t ld2_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
  return __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16));
}
// intra-shard hand-offs.  local: every workgroup of the shard runs on ONE XCD (verified at the start barrier from
// XCC_ID), whose L2 is the coherence point of its CUs: a plain store lands there and stays, and the consumers'
// L1-bypassing (sc1) loads are served from it -- nothing crosses the fabric.  Otherwise the write-through forms.
__device__ __forceinline__ void st2_shard(f64x2_t x, __amdgpu_buffer_rsrc_t rs, uint32_t byte_off, bool local) {
  if (local)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), rs, (int)byte_off, 0, 0);
  else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void st_shard(double* q, double x, bool local) {
  if (local)
    *q = x;
  else
    __hip_atomic_store(q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* q, double x) {
  __hip_atomic_store(q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// across GPUs (fine-grained memory of a peer or of this device that peers write): system scope
__device__ __forceinline__ void st_sys(double* q, double x) {
  __hip_atomic_store(q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_sys(const double* q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned sync_load(unsigned* sync, int word) {
  return __hip_atomic_load(sync + word * kSyncLine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every wave of the workgroup has drained its stores; ONE lane signals
__device__ __forceinline__ void fused_arrive(unsigned* sync, int word) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add(sync + word * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: counter >= target, or give up (returns false; *fail set on a timeout of this lane's own)
// (system: the counter receives remote adds of linked solvers on other GPUs)
// ctr: the array the counter lives in (vsync, or vcol for the merges' slice counters); sync: vsync (go / abort words)
__device__ __forceinline__ bool fused_poll(unsigned* ctr, int word, unsigned target, LamParams* lamp, bool system = false,
                                           unsigned* sync = nullptr) {
  if (!sync) sync = ctr;
  const long long t0 = wall_clock64();
  for (unsigned spins = 1;; ++spins) {
    const unsigned now = system ? __hip_atomic_load(ctr + word * kSyncLine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                : sync_load(ctr, word);
    if (now >= target) return true;
    __builtin_amdgcn_s_sleep(4);
    if ((spins & 63u) == 0) {
      if (sync_load(sync, kSyncGo) != 1u) return false;          // somebody else gave up
      if (wall_clock64() - t0 > kFusedWaitTicks) {
        __hip_atomic_store(sync + kSyncGo * kSyncLine, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&lamp->fused_abort, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + kSyncSticky * kSyncLine, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// The start barrier (one lane per workgroup).  Returns true when the arbiter word says "go".
__device__ __forceinline__ bool fused_start(unsigned* sync, LamParams* lamp, unsigned grid) {
  unsigned* go = sync + kSyncGo * kSyncLine;
  const long long t0 = wall_clock64();
  if (blockIdx.x == 0) {
    unsigned expect = 0u;
    for (;;) {
      if (sync_load(sync, kSyncStart) >= grid) {
        __hip_atomic_compare_exchange_strong(go, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      if (sync_load(sync, kSyncGo) != 0u) break;
      if (wall_clock64() - t0 > kFusedStartTicks) {
        __hip_atomic_compare_exchange_strong(go, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  } else {
    for (;;) {
      if (sync_load(sync, kSyncGo) != 0u) break;
      if (wall_clock64() - t0 > 2 * kFusedStartTicks) {     // workgroup 0 itself is not running
        unsigned expect = 0u;
        __hip_atomic_compare_exchange_strong(go, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  const unsigned decision = sync_load(sync, kSyncGo);
  if (decision != 1u && decision != 3u) {
    __hip_atomic_store(&lamp->fused_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(sync + kSyncSticky * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return decision == 1u;
}

// deterministic sum over the workgroup (every thread gets it); two barriers
__device__ __forceinline__ double fused_block_sum(double a, double* red) {
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int wv = 0; wv < kLdsBlock / 64; ++wv) t += red[wv];
  __syncthreads();
  return t;
}

// every workgroup leaves through here: the last one out resets the counters for the next launch (and moves the
// generators on when this launch produced a generation of the sample order)
__device__ __forceinline__ void fused_leave(const SagaDev& d, LamParams* lamp, int nb, bool epoch_done) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* sync = d.vsync;
    if (epoch_done && blockIdx.x == 0) {
      end_epoch(lamp, nb);
      // linked solvers: one more launch whose merges the slice counters have counted (this workgroup says so, not
      // the last one out: that may be a generators' workgroup, which knows nothing of the epoch)
      if (d.n_peers > 1) __hip_atomic_fetch_add(sync + kSyncSeq * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned prev = __hip_atomic_fetch_add(sync + kSyncExit * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == gridDim.x) {
      // (linked solvers: the other ranks add to this rank's slice counters whenever THEY get there -- those run on
      //  from launch to launch, with the launch count as their base)
      for (int wd = 0; wd < kSyncLines; ++wd)
        __hip_atomic_store(sync + wd * kSyncLine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(d.n_peers > 1))
        for (int wd = 0; wd < kFusedMaxBps; ++wd)
          __hip_atomic_store(d.vcol + wd * kSyncLine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d.rngdev && lamp->rng_generate) d.rngdev->gen += 1u;
    }
  }
}

// kPeers: the instantiation for linked solvers (the merge reaches over the ranks).  A template parameter, not a run-time
// branch: the draw loop has no register to spare, and the extra live values of the linked merge pushed three of its
// values into scratch (+20 % per epoch on ONE GPU, measured) -- the unlinked instantiation does not carry them.
template <bool kPeers>
__global__ __launch_bounds__(kLdsBlock) void saga_vs_epoch_kernel(SagaDev d, LamParams* lamp, int nb, int every) {
  extern __shared__ __attribute__((aligned(16))) double Dl[];
  __shared__ int ticket_counter;
  __shared__ int sh_ok;
  __shared__ double sh_red[kLdsBlock / 64];
  __shared__ double sh_val[4];                  // [0] b0 of the round, [1] sum of gc of the shard's batch
  const int tid = threadIdx.x;
  const int64_t p = d.p;                        // K == 1, p even
  const int64_t P2 = p >> 1;
  const int S = d.v_bps, V = d.V;
  if ((int)blockIdx.x >= V * S) {
    // ---- the generators' workgroups: the NEXT epoch's raw words, then the generators' start states one epoch on ----
    // (r_rng_bodies.hpp; nothing here waits for the epoch's workgroups, nor they for this)
    if (d.rngdev && lamp->rng_generate) {
      const RngDev* R = d.rngdev;
      const unsigned gen = R->gen;
      uint32_t* lds = reinterpret_cast<uint32_t*>(Dl);
      const int rb = (int)blockIdx.x - V * S, nrb = (int)gridDim.x - V * S;
      mt_state_body(rb, nrb, reinterpret_cast<uint32_t(*)[2][kMtN + 1]>(lds), R->state[gen & 1u], R->ends,
                    R->stream + (int64_t)(gen & 1u) * R->n, R->n, R->seg, R->gens);
      __syncthreads();
      mt_jump_body(rb, nrb, lds, R->state[gen & 1u], R->state[(gen + 1u) & 1u], R->poly, R->gens);
    }
    fused_leave(d, lamp, nb, false);
    return;
  }
  // workgroups are dealt round-robin over the XCDs: shard = index modulo V puts a shard's workgroups on one XCD when
  // V == 8 (speed only; what the hardware really did is checked at the start barrier)
  const int v = (int)blockIdx.x % V, wi = (int)blockIdx.x / V;
  const int F = 2 * (int)((p + 2 * S - 1) / (2 * S));            // features of a workgroup's slice (even)
  const int j0 = wi * F;
  const int jn = p - j0 < F ? (p - j0 > 0 ? (int)(p - j0) : 0) : F;
  double* Wl = Dl + p;
  f64x2_t* D2 = reinterpret_cast<f64x2_t*>(Dl);
  f64x2_t* W2 = reinterpret_cast<f64x2_t*>(Wl);
  // per-lambda parameters: read once (the epoch's bookkeeping rewrites LamParams at the end)
  const int penalty = lamp->penalty;
  // (the round's scalars are kept in LDS and read where they are used: as kernel-long register values they -- with the
  //  other uniform values of this kernel -- overflowed the scalar registers into vector registers the draw loop needs)
  __shared__ double sh_par[10];                 // gamma, beta, r_full, ls_full, r_tail, ls_tail, n_v, lo_v, total samples
  if (tid == 0) {
    sh_par[0] = lamp->gamma;
    sh_par[1] = lamp->beta;
    sh_par[2] = lamp->r_full;
    sh_par[3] = lamp->ls_full;
    sh_par[4] = lamp->r_tail;
    sh_par[5] = lamp->ls_tail;
    double tot = 0.0, lo = 0.0;
    for (int u = 0; u < d.V; ++u) {
      tot += d.v_size[u];
      if (u < (int)blockIdx.x % d.V) lo += d.v_size[u];
    }
    sh_par[6] = d.v_size[(int)blockIdx.x % d.V];
    sh_par[7] = lo;
    sh_par[8] = tot;
  }
  const int64_t m_full = lamp->m_full;
  const int64_t sbase = lamp->stream_base;
  const bool raw_words = lamp->stream_raw != 0;   // the stream holds the generators' raw words: every workgroup turns its own share into draws
  const int64_t dps = d.v_dps;
  const bool std_x = d.standardize != 0;
  const int64_t L = 2 * p + 2;                  // [g_sum | w | g_sum_intercept | intercept]
  double* vwv = d.vw + (int64_t)v * p;
  double* vGv = d.vG + (int64_t)v * p;
  double* refv = d.vx + (int64_t)v * L;
  double* cwp = d.vx + (int64_t)V * L;          // c . w of every workgroup's slice: V x kFusedMaxBps
  unsigned* sync = d.vsync;
  const __amdgpu_buffer_rsrc_t rs_slab = fused_rsrc(d.slab, (int64_t)V * S * p * 8);
  const __amdgpu_buffer_rsrc_t rs_w = fused_rsrc(vwv, p * 8);
  // this workgroup's share of a shard-batch of m draws (the ranges K1Compact::begin hands out): [lo, hi)
  uint32_t* const stream_v = const_cast<uint32_t*>(d.stream) + sbase + (int64_t)v * dps;
  // raw words -> draws from shard v's sample range, in place, in two steps: the words of this workgroup's share
  // are requested early (conv_load), converted and stored a phase later (conv_store) -- no round trip is waited for
  constexpr int kC = 8;                         // words per thread: shares of up to 8192 draws
  auto conv_range = [&](int m, int& lo, int& hi) {
    const int share = ((m + S - 1) / S + kTicket - 1) / kTicket * kTicket;
    lo = wi * share;
    hi = lo + share < m ? lo + share : m;
  };
  auto conv_load = [&](int64_t t0, int m, uint32_t (&x)[kC]) {
    int lo, hi;
    conv_range(m, lo, hi);
    const uint32_t* q = stream_v + t0;
#pragma unroll
    for (int c = 0; c < kC; ++c) {
      const int i = lo + (int)threadIdx.x + c * kLdsBlock;
      x[c] = i < hi ? q[i] : 0u;
    }
  };
  auto conv_store = [&](int64_t t0, int m, const uint32_t (&x)[kC]) {
    int lo, hi;
    conv_range(m, lo, hi);
    uint32_t* q = stream_v + t0;
    const double n_d = sh_par[6], lo_v = sh_par[7];
#pragma unroll
    for (int c = 0; c < kC; ++c) {
      const int i = lo + (int)threadIdx.x + c * kLdsBlock;
      if (i < hi) q[i] = (uint32_t)lo_v + word_to_draw(x[c], n_d);
    }
    for (int i = lo + (int)threadIdx.x + kC * kLdsBlock; i < hi; i += kLdsBlock)   // longer shares: one word at a time
      q[i] = (uint32_t)lo_v + word_to_draw(q[i], n_d);
  };

#ifdef SGDNET_PHASE_TIMING
  // thread 0's time per phase, summed over the rounds: dbg[workgroup * 16 + phase]; slots 14 / 15: first and last stamp
  unsigned long long ph_t = 0;
  if (d.dbg && tid == 0) d.dbg[(size_t)blockIdx.x * 16 + 14] = ph_t = phase_stamp();
#define FPH(slot)                                                 \
  do {                                                            \
    if (d.dbg && tid == 0) {                                      \
      const unsigned long long now = phase_stamp();               \
      d.dbg[(size_t)blockIdx.x * 16 + (slot)] += now - ph_t;      \
      d.dbg[(size_t)blockIdx.x * 16 + 15] = ph_t = now;           \
    }                                                             \
  } while (0)
#else
#define FPH(slot) ((void)0)
#endif
  // ---- start barrier: before "go" nothing is modified ---------------------------------------------
  if (tid == 0) {
    // which XCD runs this workgroup: the shard's workgroups OR their bits together before they arrive
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;        // HW_REG_XCC_ID
    __hip_atomic_fetch_or(sync + (kSyncXcd + v) * kSyncLine, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(sync + kSyncStart * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_ok = fused_start(sync, lamp, (unsigned)(V * S)) ? 1 : 0;
    if (sh_ok) {
      const unsigned mask = sync_load(sync, kSyncXcd + v);
      sh_ok = (mask & (mask - 1u)) == 0u ? 3 : 1;        // bit 1: the whole shard on one XCD
    }
  }
  for (int64_t i = tid; i < P2; i += kLdsBlock) D2[i] = f64x2_t{0.0, 0.0};
  __syncthreads();
  bool done = false;
  bool alive = sh_ok != 0;
  unsigned col_base = 0u;                       // linked solvers: merges of the launches before this one
  if (kPeers) {
    int merges = 0;
    for (int r = 0; r < nb; ++r) merges += (r + 1 == nb || (r + 1) % every == 0) ? 1 : 0;
    col_base = sync_load(sync, kSyncSeq) * (unsigned)merges;
  }
  const bool local = (sh_ok & 2) != 0 && d.vs_xcd_local != 0;
  int mi = 0;                                   // merges so far
  // the sample ids of a round's first two passes are requested a phase ahead; only they are carried over (the
  // rest of K1Compact is lane geometry, set up again at the top of the round: fewer registers live across the phases)
  uint32_t s_first = 0u, s_second = 0u;
  if (alive) {
    const int m0 = (int)(dps < m_full ? dps : m_full);
    if (raw_words) {
      uint32_t x0[kC];
      conv_load(0, m0, x0);
      conv_store(0, m0, x0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                          // the ids below are read by other waves than the ones that wrote them
    }
    K1Compact nx;
    nx.begin(d, stream_v, m0, wi, S, &ticket_counter);
    s_first = nx.s_cur;
    s_second = nx.s_nxt;
  }
  for (int r = 0; alive && r < nb; ++r) {
    const int64_t t0 = (int64_t)r * m_full;
    const int m = (int)(dps - t0 < m_full ? dps - t0 : m_full);
    const bool tail = m != m_full;
    const bool last = r + 1 == nb;
    const bool merge_due = last || (r + 1) % every == 0;

    // ---- stage: long-row bits of the first draws (their ids were requested a phase ago), w, intercept --------
    if (tid == 0) ticket_counter = K1Compact::static_tickets();
    K1Compact cg;
    cg.init(d, stream_v + t0, m, wi, S, &ticket_counter);
    cg.s_cur = s_first;
    cg.s_nxt = s_second;
    cg.tag_first();
    int ts = threadIdx.x;                       // (opaque, as tq below)
    asm volatile("" : "+v"(ts));
    double cw = 0.0, bv = 0.0;                  // wave 0: requested in front of the staging loads (one round trip for all)
    if ((ts >> 6) == 0) {
      if (std_x && r > 0)
        for (int k = ts & 63; k < S; k += 64) cw += ld_sc1(cwp + v * kFusedMaxBps + k);
      bv = r == 0 ? d.b[0] : ld_sc1(d.vb + v);
    }
    {
      constexpr int kStage = 8;                 // one round of loads for up to 16 384 coefficients
      for (int64_t i0 = ts; i0 < P2; i0 += (int64_t)kLdsBlock * kStage) {
        f64x2_t t[kStage];
#pragma unroll
        for (int q = 0; q < kStage; ++q) {
          const int64_t i = i0 + (int64_t)q * kLdsBlock;
          t[q] = f64x2_t{0.0, 0.0};
          if (i < P2) t[q] = r == 0 ? reinterpret_cast<const f64x2_t*>(d.w)[i] : ld2_sc1(rs_w, (uint32_t)(i * 16));
        }
#pragma unroll
        for (int q = 0; q < kStage; ++q) {
          const int64_t i = i0 + (int64_t)q * kLdsBlock;
          if (i < P2) W2[i] = t[q];
        }
      }
    }
    if ((ts >> 6) == 0) {                       // b0 = b_v - c . w_v
      cw = wave_sum(cw);
      if ((ts & 63) == 0) sh_val[0] = bv - cw;
    }
    __syncthreads();
    if (r == 0 && std_x) {                      // c . w of the state the epoch starts from
      double a = 0.0;
      for (int64_t j = ts; j < p; j += kLdsBlock) a += d.c[j] * Wl[j];
      a = fused_block_sum(a, sh_red);
      if (ts == 0) sh_val[0] -= a;
      __syncthreads();
    }
    const double b0 = sh_val[0];
    FPH(0);

    // ---- draws ------------------------------------------------------------------------------
    const double gct = cg.run(d, b0, Wl, Dl);
    __syncthreads();
    FPH(1);
    // (an opaque copy of the thread id: nothing computed from it below can be hoisted out of the round and kept in
    //  registers across the draw loop, which has none to spare)
    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
    const int lane = tq & 63, wave = tq >> 6;
    const int64_t t0n = t0 + m_full;
    const int mn = (int)(dps - t0n < m_full ? dps - t0n : m_full);
    uint32_t xraw[kC];
    const bool conv_next = !last && raw_words;

    // ---- publish the slab (write-through) and the sum of the gradient changes -----------------------
    {
      const uint32_t base = (uint32_t)(((int64_t)blockIdx.x * p) * 8);
      for (int64_t i = tq; i < P2; i += kLdsBlock) {
        st2_shard(D2[i], rs_slab, base + (uint32_t)(i * 16), local);
        D2[i] = f64x2_t{0.0, 0.0};
      }
      const double t = wave_sum(gct);
      if (lane == 0) sh_red[wave] = t;
      __syncthreads();
      if (tq == 0) {
        double tot = 0.0;
        for (int wv = 0; wv < kLdsBlock / 64; ++wv) tot += sh_red[wv];
        st_shard(d.vd0 + blockIdx.x, tot, local);
      }
    }
    fused_arrive(sync, kSyncCnt1 + v);
    FPH(2);
    // the next round's share of the sample order: requested behind the arrival (in front of it the arrival's drain
    // waited for these loads: +3 us per round for every workgroup), converted and stored behind the poll below
    if (conv_next) conv_load(t0n, mn, xraw);
    // behind the arrival, while the rest of the shard finishes: this workgroup's own coefficients (nobody else
    // writes them), and the next round's share of the sample order.
    // A thread owns one PAIR (A, B) of the replicated state: (g_sum_j, w_j) of its feature j, or -- the first thread
    // past the slice in the shard's workgroup 0 -- (g_sum_intercept, intercept): sweep and merge treat both alike
    // (fewer values in flight than two code paths: the phases of this kernel compete with the draw loop for registers)
    const bool upd = tq < jn;
    const bool icpt = wi == 0 && tq == jn;
    const bool act = upd || icpt;
    const int64_t j = j0 + tq;
    const int64_t oa = upd ? j : 2 * p, ob = upd ? p + j : 2 * p + 1;      // offsets in [g_sum | w | g_sum_b | b]
    double* const rep_a = upd ? vGv + j : d.vgb + v;                        // the replica's pair
    double* const rep_b = upd ? vwv + j : d.vb + v;
    const double* const own_a = upd ? d.G + j : d.gb;                       // the solver's own state (the epoch's start)
    const double* const own_b = upd ? d.w + j : d.b;
    double a_old = 0.0, b_old = 0.0, c_own = 0.0, ra = 0.0, rb = 0.0;
    if (act) {
      a_old = r == 0 ? *own_a : ld_sc1(rep_a);
      b_old = r == 0 ? *own_b : ld_sc1(rep_b);
      if (std_x && upd) c_own = d.c[j];
      if (merge_due) {
        ra = mi == 0 ? *own_a : ld_sc1(refv + oa);
        rb = mi == 0 ? *own_b : ld_sc1(refv + ob);
      }
    }
    if (tq == 0) sh_ok = fused_poll(sync, kSyncCnt1 + v, (unsigned)S * (unsigned)(r + 1), lamp) ? 1 : 0;
    __syncthreads();
    if (!sh_ok) break;
    FPH(3);
    if (conv_next) conv_store(t0n, mn, xraw);   // (the words arrived while the counter was polled)

    // ---- sweep of this workgroup's feature slice ------------------------------------------------
    {
      double gsum = 0.0;                        // requested in front of the slab loads
      if (wave == kLdsBlock / 64 - 1)
        for (int k = lane; k < S; k += 64) gsum += ld_sc1(d.vd0 + k * V + v);
      f64x2_t acc[kFusedChunks];
#pragma unroll
      for (int c = 0; c < kFusedChunks; ++c) acc[c] = f64x2_t{0.0, 0.0};
      for (int k = wave; k < S; k += 2 * (kLdsBlock / 64)) {
        const bool two = k + kLdsBlock / 64 < S;
        const uint32_t o0 = (uint32_t)((((int64_t)(k * V + v)) * p + j0) * 8);
        const uint32_t o1 = (uint32_t)((((int64_t)((k + kLdsBlock / 64) * V + v)) * p + j0) * 8);
        f64x2_t a0[kFusedChunks], a1[kFusedChunks];
#pragma unroll
        for (int c = 0; c < kFusedChunks; ++c) {
          const int pi = c * 64 + lane;
          a0[c] = a1[c] = f64x2_t{0.0, 0.0};
          if (2 * pi < jn) {
            a0[c] = ld2_sc1(rs_slab, o0 + (uint32_t)(pi * 16));
            if (two) a1[c] = ld2_sc1(rs_slab, o1 + (uint32_t)(pi * 16));
          }
        }
#pragma unroll
        for (int c = 0; c < kFusedChunks; ++c) {
          acc[c] += a0[c];
          acc[c] += a1[c];
        }
      }
#pragma unroll
      for (int c = 0; c < kFusedChunks; ++c) {
        const int pi = c * 64 + lane;
        if (2 * pi < F) reinterpret_cast<f64x2_t*>(Wl + (int64_t)wave * F)[pi] = acc[c];
      }
      if (wave == kLdsBlock / 64 - 1) {         // the shard's sum of gc (intercept accumulator; implicit centring)
        gsum = wave_sum(gsum);
        if (lane == 0) sh_val[1] = gsum;
      }
    }
    __syncthreads();
    double a_new = a_old, b_new = b_old;        // (g_sum, w) of the feature, or (g_sum_intercept, intercept)
    {
      const double d0 = sh_val[1];
      const double n_d = sh_par[6], gamma = sh_par[0];
      if (upd) {
        double dj = 0.0;
#pragma unroll
        for (int wv = 0; wv < kLdsBlock / 64; ++wv) dj += Wl[(int64_t)wv * F + tq];
        if (std_x) dj -= c_own * d0;            // implicit centring: D_j -= c_j * sum(gc)
        const double beta = sh_par[1];
        const double r_m = tail ? sh_par[4] : sh_par[2], ls_m = tail ? sh_par[5] : sh_par[3];
        const double val = r_m * b_old - (gamma * ls_m) * a_old - gamma * dj;
        const double tau = beta * gamma * ls_m;
        b_new = val;
        if (penalty == SGDNET_ELASTICNET) {
          b_new = soft_threshold(val, tau);
        } else if (penalty == SGDNET_GROUPLASSO) {          // penalties.h:61-79 with one response
          const double factor = tau / sqrt(val * val);
          b_new = factor < 1.0 ? val * (1.0 - factor) : 0.0;
        }
        a_new = (dj != 0.0 || penalty == SGDNET_GROUPLASSO) ? a_old + dj / n_d : a_old;
      } else if (icpt && d.fit_intercept) {     // saga-sparse.h:300-304 in batches
        const double dk = d0 / n_d;
        a_new = a_old + dk;
        b_new = b_old - gamma * (a_new * (d.xd ? 1.0 : 0.01) * (double)m + dk);
      }
    }
    FPH(4);
    if (merge_due) {
      // ---- periodic average of the replicas, slice by slice ------------------------------------------
      // Linked solvers (one per GPU of a node, SagaDev::peers): the average runs over the replicas of EVERY rank.
      // A workgroup publishes its slice in its own exchange buffer, adds to counter col[i] of every rank (remote
      // atomics over xGMI), waits for its own counter and reads the ranks' buffers with direct loads: the
      // "all-reduce" of this scheme is 2 (p / S) doubles per workgroup and shard, inside the epoch's one launch.
      // Those buffers and counters are fine-grained allocations, the accesses system-scope (sc0 sc1).
      const FusedPeers* PR = kPeers ? d.peers : nullptr;
      const int NP = kPeers ? d.n_peers : 1;
      double* pubv = d.vpub + (int64_t)((mi & 1) * V + v) * L;
      if (PR) {
        if (act) {
          st_sys(pubv + oa, a_new);
          st_sys(pubv + ob, b_new);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tq == 0)
          for (int q = 0; q < NP; ++q)
            __hip_atomic_fetch_add(PR->sync[q] + wi * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        if (act) {
          st_sc1(pubv + oa, a_new);
          st_sc1(pubv + ob, b_new);
        }
        fused_arrive(d.vcol, wi);
      }
      if (tq == 0)
        sh_ok = fused_poll(d.vcol, wi, (unsigned)(NP * V) * (col_base + (unsigned)(mi + 1)), lamp, PR != nullptr, sync) ? 1 : 0;
      __syncthreads();
      if (!sh_ok) break;
      double ma = ra, mb = rb;
      const double tot_all = PR ? PR->tot_size : sh_par[8];
      for (int q = 0; q < NP; ++q) {              // rank after rank, shard after shard: the same order everywhere
        const double* pub0 = (PR ? PR->pub[q] : d.vpub) + (int64_t)((mi & 1) * V) * L;
        double xa[8], xb[8];                      // every load of a rank's exchange first: one round trip
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xa[u] = xb[u] = 0.0;
          if (u < V && act) {
            xa[u] = PR ? ld_sys(pub0 + (int64_t)u * L + oa) : ld_sc1(pub0 + (int64_t)u * L + oa);
            xb[u] = PR ? ld_sys(pub0 + (int64_t)u * L + ob) : ld_sc1(pub0 + (int64_t)u * L + ob);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u < V) {
            const double wt = (PR ? PR->vsize[q][u] : d.v_size[u]) / tot_all;
            ma += wt * (xa[u] - ra);
            mb += wt * (xb[u] - rb);
          }
        }
      }
      if (act) {
        a_new = ma;
        b_new = mb;
        st_sc1(refv + oa, ma);
        st_sc1(refv + ob, mb);
        if (last && v == 0) {                     // the solver's own state: what the epoch returns
          *const_cast<double*>(own_a) = ma;
          *const_cast<double*>(own_b) = mb;
        }
      }
      ++mi;
      FPH(5);
    }
    if (act) {
      st_shard(rep_a, a_new, local);
      st_shard(rep_b, b_new, local);
    }
    if (std_x) {                                // c . w of the slice, for the next round's linear predictors
      const double a = fused_block_sum(upd ? c_own * b_new : 0.0, sh_red);
      if (tq == 0) st_shard(cwp + v * kFusedMaxBps + wi, a, local);
    }
    if (last) {
      done = true;
      break;
    }
    fused_arrive(sync, kSyncCnt2 + v);
    FPH(6);
    // the next round's first sample ids, requested before the wait (converted by this workgroup a phase ago)
    {
      K1Compact nx;
      nx.begin(d, stream_v + t0n, mn, wi, S, &ticket_counter);
      s_first = nx.s_cur;
      s_second = nx.s_nxt;
    }
    if (tq == 0) sh_ok = fused_poll(sync, kSyncCnt2 + v, (unsigned)S * (unsigned)(r + 1), lamp) ? 1 : 0;
    __syncthreads();
    if (!sh_ok) break;
    FPH(7);
  }

  fused_leave(d, lamp, nb, done);
}

// ConvergenceCheck (src/utils.h:240-262): max |w - w_prev| and max |w|, then w_prev = w.
__global__ __launch_bounds__(kBlock) void saga_convergence_kernel(SagaDev d, LamParams* lamp) {
  const int64_t len = (int64_t)d.K * d.p;
  double max_change = 0.0, max_size = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    const double v = d.w[i];
    max_change = fmax(max_change, fabs(v - d.w_prev[i]));
    max_size = fmax(max_size, fabs(v));
    d.w_prev[i] = v;
  }
  max_change = wave_max(max_change);
  max_size = wave_max(max_size);
  if ((threadIdx.x & 63) == 0) {
    atomic_max_bits(&lamp->max_change_bits, max_change);
    atomic_max_bits(&lamp->max_size_bits, max_size);
  }
}

// Sum of per-sample losses: Deviance / 2 (src/utils.h:304-329) or n * EpochLoss (:199-227).
template <bool kSparse>
__global__ __launch_bounds__(kBlock) void saga_loss_kernel(SagaDev d, LamParams* lamp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lps = reinterpret_cast<double*>(smem);   // [groups per block][K]
  const int K = d.K;
  const int gl = threadIdx.x & (kGroup - 1);
  const int gib = threadIdx.x / kGroup;
  const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kGroup;
  const int64_t ngroups = (int64_t)gridDim.x * (kBlock / kGroup);
  double* lp = lps + (size_t)gib * K;
  // implicit centring (saga-sparse.h:276-277): the reference subtracts sum_j w_kj c_j from every
  // sample's linear predictor; it is the same K numbers for all samples, so each workgroup
  // computes them once (per sample it was O(p K): 94 ms per deviance at 500k x 20k x 10)
  double* cw = lps + (size_t)(kBlock / kGroup) * K;
  if (kSparse && d.standardize) {
    __shared__ double red[kBlock / 64];
    for (int k = 0; k < K; ++k) {
      double a = 0.0;
      for (int64_t j = threadIdx.x; j < d.p; j += kBlock) a += d.w[k + j * K] * d.c[j];
      a = wave_sum(a);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
        for (int wv = 0; wv < kBlock / 64; ++wv) t += red[wv];
        cw[k] = t;
      }
      __syncthreads();
    }
  }
  double loss = 0.0;
  if (kSparse && K > 1 && K <= kGroup) {
    // several classes of sparse x (round 4): lane k of the group = class k, the group walks the row together -- a non-zero
    // is ONE request for the K contiguous coefficients of its feature instead of K requests of 8 bytes, and the row is
    // read once instead of K times (config 5: 37 -> 9 ms per deviance, a hundred of them along the path)
    const int kl = gl < K ? gl : 0;
    const double off = (gl < K ? d.b[kl] : 0.0) - (d.standardize ? cw[kl] : 0.0);
    for (int64_t s = group; s < d.n; s += ngroups) {
      const int64_t q0 = d.ptr[s], q1 = d.ptr[s + 1];
      double acc = 0.0;
      for (int64_t q = q0; q < q1; q += 4) {
        double xv[4];
        int64_t jv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = q + u < q1 ? d.val[q + u] : 0.0;
          jv[u] = q + u < q1 ? (int64_t)d.idx[q + u] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += xv[u] * d.w[kl + jv[u] * K];
      }
      if (gl < K) lp[gl] = acc + off;
      __builtin_amdgcn_wave_barrier();
      if (gl == 0) loss += family_loss(d.family, K, lp, d.y + s * d.Ky);
      __builtin_amdgcn_wave_barrier();
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0 && loss != 0.0) atomic_add_f64(&lamp->loss_acc, loss);
    return;
  }
  for (int64_t s = group; s < d.n; s += ngroups) {
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      if (kSparse) {
        for (int64_t q = d.ptr[s] + gl; q < d.ptr[s + 1]; q += kGroup)
          acc += d.val[q] * d.w[k + (int64_t)d.idx[q] * K];
      } else {
        for (int64_t j = gl; j < d.p; j += kGroup) acc += d.xd[s * d.p + j] * d.w[k + j * K];
      }
      acc = group_sum(acc);
      if (gl == 0) lp[k] = acc - (kSparse && d.standardize ? cw[k] : 0.0) + d.b[k];
    }
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) loss += family_loss(d.family, K, lp, d.y + s * d.Ky);
    __builtin_amdgcn_wave_barrier();
  }
  loss = wave_sum(loss);
  if ((threadIdx.x & 63) == 0 && loss != 0.0) atomic_add_f64(&lamp->loss_acc, loss);
}

// Multi-GPU merge helpers.  Layout: [dG (Kp) | dw (Kp) | dgb (K) | db (K)].
__global__ __launch_bounds__(kBlock) void saga_delta_export_kernel(SagaDev d, const double* ref,
                                                                   double* out, double weight) {
  const int64_t KP = (int64_t)d.K * d.p;
  const int64_t len = 2 * KP + 2 * d.K;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    double cur;
    if (i < KP) cur = d.G[i];
    else if (i < 2 * KP) cur = d.w[i - KP];
    else if (i < 2 * KP + d.K) cur = d.gb[i - 2 * KP];
    else cur = d.b[i - 2 * KP - d.K];
    out[i] = weight * (cur - ref[i]);
  }
}

// the merged state also becomes the new reference (the next local run's snapshot)
__global__ __launch_bounds__(kBlock) void saga_delta_apply_kernel(SagaDev d, double* ref, const double* merged,
                                                                  double w_weight) {
  const int64_t KP = (int64_t)d.K * d.p;
  const int64_t len = 2 * KP + 2 * d.K;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    const bool coef = (i >= KP && i < 2 * KP) || i >= 2 * KP + d.K;
    const double v = ref[i] + (coef ? w_weight : 1.0) * merged[i];
    ref[i] = v;
    if (i < KP) d.G[i] = v;
    else if (i < 2 * KP) d.w[i - KP] = v;
    else if (i < 2 * KP + d.K) d.gb[i - 2 * KP] = v;
    else d.b[i - 2 * KP - d.K] = v;
  }
}

// --------------------------------------------------------------------------
// Binned form: K x p tables that fit no LDS (config 5: K = 10, p = 100 000 -> 8 MB).
//
// With the scatter accumulator in global memory every non-zero of every draw costs K fp64
// atomics, and those execute at the memory side at a fixed chip-wide byte rate (~1.3 TB/s of
// added bytes, MI355X_MICROARCH.md "Global float atomics"): 800 B per draw at K = 10, z = 10,
// i.e. < 1.6 G draws/s whatever the kernel does.  Here the features are cut into R contiguous
// ranges of equal non-zero mass whose K x width slice fits a workgroup's LDS, and a batch runs
// as two kernels without a single global fp atomic:
//
//   gather+bin  (a workgroup per 512 draws, class-lane form): record, x.w from the L2-resident
//               w, gradient, gradient-memory update; the draw's gradient change goes to
//               gcb[t][0..K) and every non-zero becomes a 16-byte entry {t, j, x_tj} staged in
//               LDS, counted per range, and written out behind ONE returning atomic per
//               (workgroup, range) that reserves the run's place in the range's bin;
//   range sweep (a workgroup per range): D[:, lo..hi) in LDS <- sum over the bin's entries of
//               x_tj * gcb[t] (ds_add_f64), then the reference's per-feature update
//               (saga-sparse.h:316-335 / penalties.h via sweep_feature) for its own features,
//               the intercept and centring scalars exactly as in the other sweep kernels.
//
// The entries of a batch (~16 B x z per draw) are written and read once and stay in the
// Infinity Cache between the two kernels; the sums are order-dependent in the last bits like
// every other scatter of this file.
// --------------------------------------------------------------------------
struct __attribute__((aligned(16))) BinEntry {
  uint32_t t;   // draw index inside the batch
  uint32_t j;   // feature
  double x;
};
static_assert(sizeof(BinEntry) == 16, "bin entries are 16 bytes");

constexpr int kBinBlock = SGDNET_BIN_BLOCK;   // gather+bin threads: one 16-lane group per draw in flight
constexpr int kBinDraws = kBinBlock / 2;      // draws per gather workgroup (8 passes)
constexpr int kBinEntCap = 7 * kBinBlock;     // LDS staging capacity (entries); beyond it entries go out one by one
constexpr int kBinW = SGDNET_BIN_W;           // reads of w requested together (8 or 16)
constexpr int kRangeBlock = SGDNET_RANGE_BLOCK;   // range sweep threads (78 VGPRs: 24 waves per CU)
constexpr size_t kRangeLdsBytes = 64 * 1024;

size_t binned_max_range_features(int K) { return kRangeLdsBytes / (sizeof(double) * (size_t)K); }

__global__ __launch_bounds__(256) void col_count_kernel(const int32_t* idx, int64_t nnz, unsigned* counts) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * 256)
    atomicAdd(counts + idx[q], 1u);
}

__global__ __launch_bounds__(256) void wpad_refresh_kernel(const double* w, double* wpad, int K, int KS, int64_t p) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < p * KS) {
    const int64_t j = t / KS;
    const int k = (int)(t - j * KS);
    wpad[t] = k < K ? w[j * K + k] : 0.0;
  }
}

// Second moment of the entries one sample sends to a feature range: sumsq[r] = sum_i c_ir^2 with c_ir the
// non-zeros of sample i inside range r.  A batch of m uniformly drawn samples sends range r a sum of m such
// counts -- mean m * mass_r / n, variance <= m * sumsq[r] / n -- and that, not a Poisson model of independent
// entries, is what the bins have to hold: rows that put 16 entries into one range (block-structured x)
// arrive 16 at a time.  Feature ids ascend inside a row and ranges are contiguous, so a row is a few runs.
__global__ __launch_bounds__(256) void range_moment_kernel(const int64_t* ptr, const int32_t* idx, int64_t n,
                                                           const uint16_t* feat_range, unsigned long long* sumsq) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t q1 = ptr[i + 1];
    int cur = -1;
    unsigned long long c = 0;
    for (int64_t q = ptr[i]; q < q1; ++q) {
      const int r = feat_range[idx[q]];
      if (r != cur) {
        if (c) atomicAdd(sumsq + cur, c * c);
        cur = r;
        c = 0;
      }
      ++c;
    }
    if (c) atomicAdd(sumsq + cur, c * c);
  }
}

int launch_range_moment(const SagaDev& d, const uint16_t* feat_range, unsigned long long* sumsq, int R,
                        hipStream_t st) {
  SGD_HIP_TRY(hipMemsetAsync(sumsq, 0, sizeof(unsigned long long) * (size_t)R, st));
  int64_t grid = (d.n + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(range_moment_kernel, dim3((unsigned)grid), dim3(256), 0, st, d.ptr, d.idx, d.n, feat_range, sumsq);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// the padded copy of w the binned gather reads: refreshed at the start of every epoch (w may have
// been set from the host, merged across GPUs or advanced by an exact-mode run in between)
int launch_wpad_refresh(const SagaDev& d, hipStream_t st) {
  if (!d.wpad || d.wpad == d.w) return SGDNET_OK;
  const int64_t tot = d.p * d.KS;
  hipLaunchKernelGGL(wpad_refresh_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d.w, d.wpad, d.K, d.KS,
                     d.p);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_col_count(const SagaDev& d, int64_t nnz, unsigned* counts, hipStream_t st) {
  SGD_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(unsigned) * (size_t)d.p, st));
  int64_t grid = (nnz + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(col_count_kernel, dim3((unsigned)grid), dim3(256), 0, st, d.idx, nnz, counts);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// class-lane groups of the binned kernels: 16 lanes (K <= 16) or a whole wavefront (K <= 64)
template <int kGrp>
__device__ __forceinline__ double grp_sum(double v) {
#pragma unroll
  for (int off = kGrp / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kGrp);
  return v;
}
template <int kGrp>
__device__ __forceinline__ double grp_max(double v) {
#pragma unroll
  for (int off = kGrp / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, kGrp));
  return v;
}

// one entry straight into its bin (staging full: a workgroup that drew unusually long rows)
__device__ __forceinline__ void bin_push_global(const SagaDev& d, const BinEntry& en, unsigned r) {
  const unsigned pos = atomicAdd(d.bin_count + r, 1u);
  const int64_t b0 = d.bin_off[r];
  if ((int64_t)pos < d.bin_off[r + 1] - b0) reinterpret_cast<BinEntry*>(d.bins)[b0 + pos] = en;
  else atomicExch(d.bin_err, 1);
}

// Entries of the row beyond the 32 a group keeps in registers (record slots >= 32 and the overflow
// chain).  Uniform form: every lane sees every entry (x.w); lane form: lane gl takes entries
// gl, gl + 16, ... of every stretch (staging).
template <int kGrp, class F>
__device__ __forceinline__ void row_rest_uniform(const SagaDev& d, const char* base, int nnz, int ovf, F f) {
  const int cap = d.rec_cap;
  const int cnt0 = nnz < cap ? nnz : cap;
  const int* ridx = reinterpret_cast<const int*>(base + 16);
  const double* rval = reinterpret_cast<const double*>(base + d.rec_val_off);
  for (int e = 2 * kGrp; e < cnt0; ++e) f((uint32_t)ridx[e], rval[e]);
  int rem = nnz - cnt0;
  while (rem > 0) {
    const char* ob = d.ovf + (size_t)ovf * kOvfStride;
    const int next = reinterpret_cast<const int*>(ob)[0];
    const int c = reinterpret_cast<const int*>(ob)[1];
    const int* oi = reinterpret_cast<const int*>(ob + 8);
    const double* ov = reinterpret_cast<const double*>(ob + 8 + 4 * kOvfCap);
    for (int e = 0; e < c; ++e) f((uint32_t)oi[e], ov[e]);
    rem -= c;
    ovf = next;
  }
}

template <int kGrp, class F>
__device__ __forceinline__ void row_rest_lane(const SagaDev& d, const char* base, int nnz, int ovf, int gl, F f) {
  const int cap = d.rec_cap;
  const int cnt0 = nnz < cap ? nnz : cap;
  const int* ridx = reinterpret_cast<const int*>(base + 16);
  const double* rval = reinterpret_cast<const double*>(base + d.rec_val_off);
  for (int e = 2 * kGrp + gl; e < cnt0; e += kGrp) f((uint32_t)ridx[e], rval[e]);
  int rem = nnz - cnt0;
  while (rem > 0) {
    const char* ob = d.ovf + (size_t)ovf * kOvfStride;
    const int next = reinterpret_cast<const int*>(ob)[0];
    const int c = reinterpret_cast<const int*>(ob)[1];
    const int* oi = reinterpret_cast<const int*>(ob + 8);
    const double* ov = reinterpret_cast<const double*>(ob + 8 + 4 * kOvfCap);
    for (int e = gl; e < c; e += kGrp) f((uint32_t)oi[e], ov[e]);
    rem -= c;
    ovf = next;
  }
}

// What a 16-lane group holds of one draw before it works on it: requested one pass ahead, so the
// record's round trip to HBM overlaps the previous draw's trip to the L2-resident w.
struct BinDraw {
  uint32_t s;
  int i;              // draw index inside the batch, -1: none
  double y0;
  int nnz, ovf;
  int j0;             // record slot gl (whatever the row length: slots past it hold 0)
  double v0;
  int prev;           // lane 0: the claim this draw's exchange returned
};

__device__ __forceinline__ BinDraw bin_fetch(const SagaDev& d, int i, uint32_t s, bool valid, int gl, int batch_id) {
  BinDraw q;
  q.s = s;
  q.i = valid ? i : -1;
  q.y0 = 0.0; q.nnz = 0; q.ovf = 0; q.j0 = 0; q.v0 = 0.0; q.prev = batch_id;
  if (!valid) return q;
  const char* base = d.rec + (size_t)s * d.rec_stride;
  const int cap = d.rec_cap;
  const int* ridx = reinterpret_cast<const int*>(base + 16);
  const double* rval = reinterpret_cast<const double*>(base + d.rec_val_off);
  q.y0 = *reinterpret_cast<const double*>(base);
  const int2 h = *reinterpret_cast<const int2*>(base + 8);
  q.nnz = h.x;
  q.ovf = h.y;
  if (gl < cap) {
    q.j0 = ridx[gl];
    q.v0 = rval[gl];
  }
  if (gl == 0)
    q.prev = __hip_atomic_exchange(d.claim + s, batch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return q;
}

template <int kGrp>
__device__ __forceinline__ double shfl_d(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl((int)(b & 0xffffffffll), src, kGrp);
  const int hi = __shfl((int)(b >> 32), src, kGrp);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// kMulti: the multinomial family alone (config 5's) -- the other families' code and their pointers leave the draw loop,
// which sits at the 128-register limit of a 1024-thread workgroup (round 4: two scratch reloads per draw otherwise)
template <int kGrp, bool kMulti = false>
__global__ __launch_bounds__(kBinBlock) void saga_binned_gather_kernel(SagaDev d, const LamParams* lamp,
                                                                       int64_t t0_in_epoch, int m,
                                                                       int batch_id_offset) {
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  __shared__ double d0s[kGrp];
  __shared__ unsigned n_ent;
  __shared__ int n_next;                                 // the next draw of this workgroup nobody has taken yet
  BinEntry* ent = reinterpret_cast<BinEntry*>(bsm);
  unsigned* cnt = reinterpret_cast<unsigned*>(bsm + sizeof(BinEntry) * kBinEntCap);
  unsigned* rbase = cnt + d.R;
  int* rlo = reinterpret_cast<int*>(rbase + d.R);       // R + 1 range boundaries: the range of a feature
  const int K = d.K, KS = d.KS;                          // is found by bisection in LDS, not by a table
                                                         // look-up that costs an L2 request per non-zero
  // ... and the bisection starts from a coarse table (round 4): the range of the first feature of the 2^shift-feature cell
  // the feature lies in, and of the next cell's -- one or two steps instead of log2(R) dependent LDS reads
  unsigned short* rcl = reinterpret_cast<unsigned short*>(rlo + d.R + 1);
  const int cshift = d.coarse_shift;
  const int gl = threadIdx.x & (kGrp - 1);
  const int group = threadIdx.x / kGrp;
  const int lane = threadIdx.x & 63;
  const bool lane_on = gl < K;
  const int64_t t0 = lamp->stream_base + t0_in_epoch;
  const int batch_id = lamp->batch_seq + batch_id_offset;
  PHASE(0);
  for (int r = threadIdx.x; r < d.R; r += kBinBlock) cnt[r] = 0u;
  for (int r = threadIdx.x; r <= d.R; r += kBinBlock) rlo[r] = d.range_lo[r];
  for (int c = threadIdx.x; c <= d.n_coarse; c += kBinBlock) rcl[c] = d.range_coarse[c];
  if (threadIdx.x < kGrp) d0s[threadIdx.x] = 0.0;
  if (threadIdx.x == 0) {
    n_ent = 0u;
    n_next = (int)blockIdx.x * kBinDraws;
  }
  __syncthreads();
  const double bl = lane_on ? d.b[gl] - (d.standardize ? cw_sum(d, batch_id, gl) : 0.0) : 0.0;
  if (d.standardize) cw_clear_next(d, batch_id);
  auto range_of = [&](int j) {
    const int c = j >> cshift;
    int a = rcl[c], b = rcl[c + 1] + 1;       // rlo[a] <= j < rlo[b]
    while (b - a > 1) {
      const int mid = (a + b) >> 1;
      if (j >= rlo[mid]) a = mid; else b = mid;
    }
    return (unsigned)a;
  };

  // stage one entry per active lane: the slots of a wavefront's entries come from one LDS atomic;
  // the staged copy carries its range in the upper 12 bits of the draw index (batch <= 2^20)
  auto stage = [&](bool active, int i, uint32_t j, double v, unsigned r) {
    const unsigned long long mask = __ballot(active);
    if (mask == 0ull) return;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned slot0 = 0u;
    if (lane == leader) slot0 = atomicAdd(&n_ent, (unsigned)__popcll(mask));
    slot0 = (unsigned)__shfl((int)slot0, leader, 64);
    if (!active) return;
    const unsigned slot = slot0 + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    if (slot < (unsigned)kBinEntCap) {
      ent[slot] = BinEntry{(uint32_t)i | (r << 20), j, v};
      atomicAdd(cnt + r, 1u);
    } else {
      bin_push_global(d, BinEntry{(uint32_t)i, j, v}, r);
    }
  };

  PHASE(1);
  // Draws are handed out wavefront by wavefront (round 4): a wavefront takes the next 64 / kGrp draws of the workgroup
  // from an LDS counter, two takes ahead of the one it works on (sample id and record stay requested a pass ahead).
  // Rows differ in length and in how their entries stage, and with a fixed share per group the workgroup's barrier
  // waited 10 us of a 53 us loop for its slowest wavefront.  The four groups of a wavefront stay together, so the
  // ballots of stage() remain wave-wide.
  constexpr int kGpw = 64 / kGrp;
  const int lo = blockIdx.x * kBinDraws;
  const int hi = (lo + kBinDraws < m) ? lo + kBinDraws : m;
  (void)group;
  auto take = [&]() -> int {
    int b = 0;
    if (lane == 0) b = atomicAdd(&n_next, kGpw);
    return __builtin_amdgcn_readfirstlane(b) + lane / kGrp;
  };
  double gct = 0.0;
  int i = take();
  int i_nxt = take();
  uint32_t s_nxt = i_nxt < hi ? d.stream[t0 + i_nxt] : 0u;
  BinDraw cur = bin_fetch(d, i, i < hi ? d.stream[t0 + i] : 0u, i < hi, gl, batch_id);
  while (i - lane / kGrp < hi) {                        // (the wavefront's first draw: the same for all its lanes)
    const int i_nn = take();
    const uint32_t s_nn = i_nn < hi ? d.stream[t0 + i_nn] : 0u;
    const BinDraw nxt = bin_fetch(d, i_nxt, s_nxt, i_nxt < hi, gl, batch_id);
    // the class index of this lane, opaque to the compiler inside the loop: it otherwise keeps (array + 8 gl) of every
    // K-fastest array in a register pair across the loop, and the loop is at the 128-register limit (spills reloaded
    // per draw behind s_waitcnt vmcnt(0), i.e. behind the prefetched record)
    int glo = gl;
    asm volatile("" : "+v"(glo));
    // ---- the current draw ----
    const bool have = cur.i >= 0;
    const int cap = d.rec_cap;
    const int cnt0 = cur.nnz < cap ? cur.nnz : cap;
    const int creg = cnt0 < 2 * kGrp ? cnt0 : 2 * kGrp;
    const bool rest = have && (cur.nnz > creg);
    const char* base = d.rec + (size_t)cur.s * d.rec_stride;
    // x . w: the feature ids sit in the group's registers, so the K-contiguous reads of w are all
    // requested before the first one is used
    const unsigned r0 = range_of(cur.j0);
    const double mold = (have && lane_on) ? d.M[glo + (int64_t)cur.s * K] : 0.0;
    double acc = 0.0;
#pragma unroll
    for (int e0 = 0; e0 < kGrp; e0 += kBinW) {
      if (e0 > 0 && !__any(have && creg > e0)) break;
      double wv[kBinW];
#pragma unroll
      for (int e = 0; e < kBinW; ++e) {
        int j = __shfl(cur.j0, e0 + e, kGrp);
        // (timing only, -DSGDNET_EXPERIMENTS: every coefficient row from a 1 MB window -- an upper bound of what
        //  feature ranges held in one XCD's L2 could buy: profiles/r04_c5_xcd_bound.txt)
        if (SGD_ABLATE(d, 32)) j &= 8191;
        wv[e] = (have && e0 + e < creg && lane_on) ? d.wpad[(int64_t)j * KS + glo] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < kBinW; ++e) acc += shfl_d<kGrp>(cur.v0, e0 + e) * wv[e];
    }
    // record slots 16..31 (3 % of the rows at 10 non-zeros per sample): read where they are needed
    int j1 = 0;
    double v1 = 0.0;
    if (__any(have && creg > kGrp)) {
      if (have && kGrp + gl < creg) {
        j1 = reinterpret_cast<const int*>(base + 16)[kGrp + gl];
        v1 = reinterpret_cast<const double*>(base + d.rec_val_off)[kGrp + gl];
      }
      for (int e = 0; e < kGrp; ++e) {
        const int j = __shfl(j1, e, kGrp);
        const double v = shfl_d<kGrp>(v1, e);
        if (have && kGrp + e < creg && lane_on) acc += v * d.wpad[(int64_t)j * KS + glo];
      }
    }
    if (rest)
      row_rest_uniform<kGrp>(d, base, cur.nnz, cur.ovf, [&](uint32_t j, double v) {
        if (lane_on) acc += v * d.wpad[(int64_t)j * KS + glo];
      });
    const double lp = acc + bl;
    double g;
    if (kMulti || d.family == SGDNET_MULTINOMIAL) {
      // softmax as exp(lp - max) / sum: the same number as families.h:235-260's exp(lp - logsumexp) up to rounding
      // (batched parity is a 1e-9 tolerance), one exp and no log per class lane, and none of the log's sixteen
      // constant registers in a loop that sits at the register limit
      const double mx = grp_max<kGrp>(lane_on ? lp : -HUGE_VAL);
      const double ex = lane_on ? exp(lp - mx) : 0.0;
      g = ex / grp_sum<kGrp>(ex);
      if ((unsigned)gl == (unsigned)(cur.y0 + 0.5)) g -= 1.0;
    } else if (d.family == SGDNET_BINOMIAL) {
      g = 1.0 - cur.y0 - 1.0 / (1.0 + exp(lp));
    } else {
      g = lp - ((have && lane_on) ? d.y[(int64_t)cur.s * d.Ky + glo] : 0.0);
    }
    // a repeat inside the batch sees the same snapshot: gradient change 0, nothing to stage
    const bool first = have && (__shfl(cur.prev != batch_id ? 1 : 0, 0, kGrp) != 0);
    if (first && lane_on) {
      const double gc = g - mold;
      d.M[glo + (int64_t)cur.s * K] = g;
      d.gcb[(int64_t)cur.i * KS + glo] = gc;
      gct += gc;
    }
    stage(first && gl < creg, cur.i, (uint32_t)cur.j0, cur.v0, r0);
    if (__any(first && creg > kGrp)) {
      const bool a1 = first && kGrp + gl < creg;
      const unsigned r1 = range_of(j1);
      stage(a1, cur.i, (uint32_t)j1, v1, r1);
    }
    if (first && rest)
      row_rest_lane<kGrp>(d, base, cur.nnz, cur.ovf, gl, [&](uint32_t j, double v) {
        stage(true, cur.i, j, v, range_of((int)j));
      });
    cur = nxt;
    i = i_nxt;
    i_nxt = i_nn;
    s_nxt = s_nn;
  }
  PHASE(2);
  if (gct != 0.0) __hip_atomic_fetch_add(&d0s[gl], gct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  PHASE(3);
  // reserve this workgroup's run in every bin it has entries for, then place the entries
  // (rbase[r] becomes the ABSOLUTE entry index of the run's start, 0xffffffff if the bin is full)
  for (int r = threadIdx.x; r < d.R; r += kBinBlock) {
    const unsigned c = cnt[r];
    unsigned at = 0xffffffffu;
    if (c) {
      const unsigned pos = atomicAdd(d.bin_count + r, c);
      const int64_t b0 = d.bin_off[r], room = d.bin_off[r + 1] - b0;
      if ((int64_t)pos + c <= room) at = (unsigned)(b0 + pos);
      else atomicExch(d.bin_err, 1);
    }
    rbase[r] = at;
    cnt[r] = 0u;
  }
  __syncthreads();
  PHASE(4);
  const unsigned staged = n_ent < (unsigned)kBinEntCap ? n_ent : (unsigned)kBinEntCap;
  static_assert(sizeof(BinEntry) == sizeof(uint4), "an entry moves as one 16-byte vector");
  for (unsigned e = threadIdx.x; e < staged; e += kBinBlock) {
    const uint4 en = reinterpret_cast<const uint4*>(ent)[e];  // (a modified struct copy went through scratch memory here)
    const unsigned r = en.x >> 20;
    const unsigned at = rbase[r];
    const unsigned k = atomicAdd(cnt + r, 1u);
    if (at != 0xffffffffu) reinterpret_cast<uint4*>(d.bins)[(size_t)at + k] = make_uint4(en.x & 0xfffffu, en.y, en.z, en.w);
  }
  if ((d.fit_intercept || d.standardize) && (int)threadIdx.x < K)
    d0_publish(d, batch_id, threadIdx.x, d0s[threadIdx.x]);
  PHASE(5);
}

// kGrouped: the group-lasso update (a feature's K coefficients in one thread's registers); false: ridge / elastic net
// per element -- that instantiation (config 5's) keeps nothing in scratch memory: a kernel that declares a private
// segment pays for its set-up at every dispatch, and this one is launched 382 times per epoch
template <int kGrp, bool kGrouped>
__global__ __launch_bounds__(kRangeBlock) void saga_binned_sweep_kernel(SagaDev d, LamParams* lamp, int tail,
                                                                        int n_parts, int batch_id_offset) {
  extern __shared__ __attribute__((aligned(16))) double Dl[];
  __shared__ double sh_d0[kGrp];
  __shared__ double sh_cw[kGrp];
  const SweepParams q = load_sweep_params(d, lamp, tail, SweepOverride{0.0, 0.0, 0.0});
  const int K = d.K;
  const int r = blockIdx.x;
  const int batch_id = lamp->batch_seq + batch_id_offset;
  if (r == d.R) {
    // the extra workgroup: intercept update and the reset of the next batch's accumulator slots.
    // (Summing the gather's partials class by class takes ~9 us; inside a range's workgroup that
    // was the tail every launch waited for.)
    if (d.fit_intercept) {
      block_d0<kRangeBlock>(d, n_parts, batch_id, sh_d0);
      sweep_intercept(d, q, sh_d0);
    }
    double* nxt = d0_set(d, batch_id + 1);
    for (int i = threadIdx.x; i < kD0Slots * K; i += kRangeBlock) nxt[i] = 0.0;
    return;
  }
  const int lo = d.range_lo[r], hi = d.range_lo[r + 1];
  const int E = (hi - lo) * K;
  const bool need_d0 = d.standardize != 0;
  // ---- the bin's entries into the LDS slice ----
  // A 16-lane group takes 16 consecutive entries with one coalesced load (lane q holds entry q),
  // then works through them with lane = class: the K-contiguous gradient changes of the 16 draws
  // are requested together, the products go into the slice with ds_add_f64.  The next 16 entries
  // are requested before the current ones are used.
  const int gl = threadIdx.x & (kGrp - 1);
  const int group = threadIdx.x / kGrp;
  constexpr int kGrps = kRangeBlock / kGrp;
  constexpr int kEnt = 16;                        // entries a group takes per round (held by its first 16 lanes)
  unsigned cntb = d.bin_count[r];
  const int64_t b0 = d.bin_off[r], bcap = d.bin_off[r + 1] - b0;
  if ((int64_t)cntb > bcap) cntb = (unsigned)bcap;
  const BinEntry* bin = reinterpret_cast<const BinEntry*>(d.bins) + b0;
  const bool lane_on = gl < K;
  const BinEntry none{0u, (uint32_t)lo, 0.0};
  PHASE(6);
  unsigned e0 = (unsigned)group * kEnt;
  BinEntry mine = (gl < kEnt && e0 + gl < cntb) ? bin[e0 + gl] : none;
  for (int i = threadIdx.x; i < E; i += kRangeBlock) Dl[i] = 0.0;
  if (need_d0) block_d0<kRangeBlock>(d, n_parts, batch_id, sh_d0);
  __syncthreads();
  PHASE(7);
  for (; e0 < cntb; e0 += kGrps * kEnt) {
    const unsigned en = e0 + kGrps * kEnt;
    const BinEntry nxt = (gl < kEnt && en + gl < cntb) ? bin[en + gl] : none;
    double gq[kEnt];
#pragma unroll
    for (int qq = 0; qq < kEnt; ++qq) {
      int t = __shfl((int)mine.t, qq, kGrp);
      if (SGD_ABLATE(d, 64)) t &= 8191;         // (timing only: every gradient-change row from a 1 MB window)
      gq[qq] = (lane_on && e0 + qq < cntb) ? d.gcb[(int64_t)t * d.KS + gl] : 0.0;
    }
#pragma unroll
    for (int qq = 0; qq < kEnt; ++qq) {
      const int j = __shfl((int)mine.j, qq, kGrp);
      const double x = shfl_d<kGrp>(mine.x, qq);
      if (lane_on && e0 + qq < cntb) scatter_add<true>(Dl + (j - lo) * K + gl, x * gq[qq]);
    }
    mine = nxt;
  }
  PHASE(8);
  __syncthreads();
  PHASE(9);
  if (threadIdx.x == 0) d.bin_count[r] = 0u;                 // the next batch fills the bin again
  // ---- per-feature update of this range ----
  if (kGrouped) {
    double cwp[kGrp];
    for (int k = 0; k < K; ++k) cwp[k] = 0.0;
    for (int f = threadIdx.x; f < hi - lo; f += kRangeBlock) {
      const int64_t j = lo + f;
      double dj[kGrp], wn[kGrp];
      const double cj = d.standardize ? d.c[j] : 0.0;
      for (int k = 0; k < K; ++k) dj[k] = Dl[f * K + k] - (d.standardize ? cj * sh_d0[k] : 0.0);
      sweep_feature(d, q, j, dj, wn);
      for (int k = 0; k < K; ++k) cwp[k] += cj * wn[k];
      if (d.wpad != d.w)
        for (int k = 0; k < K; ++k) d.wpad[j * d.KS + k] = wn[k];
    }
    if (d.standardize) cw_accumulate<kRangeBlock>(d, batch_id, cwp);
  } else {
    const double tau = q.beta * q.gamma * q.ls_m, gls = q.gamma * q.ls_m;
    if (d.standardize) {                        // c . w_new of this range, class by class, through the LDS
      if ((int)threadIdx.x < kGrp) sh_cw[threadIdx.x] = 0.0;
      __syncthreads();
    }
    for (int i = threadIdx.x; i < E; i += kRangeBlock) {
      const int f = i / K, k = i - f * K;
      const int64_t t = (int64_t)lo * K + i;
      const double cj = d.standardize ? d.c[lo + f] : 0.0;
      const double dk = Dl[i] - (d.standardize ? cj * sh_d0[k] : 0.0);
      double v = q.r_m * d.w[t] - gls * d.G[t] - q.gamma * dk;
      if (q.penalty == SGDNET_ELASTICNET) v = soft_threshold(v, tau);
      d.w[t] = v;
      if (d.wpad != d.w) d.wpad[(int64_t)(lo + f) * d.KS + k] = v;
      if (dk != 0.0) d.G[t] += dk / q.n_d;
      if (d.standardize && cj * v != 0.0)
        __hip_atomic_fetch_add(&sh_cw[k], cj * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (d.standardize) {                        // (as cw_accumulate: into the next batch's c.w slots)
      __syncthreads();
      if ((int)threadIdx.x < K) {
        const double tot = sh_cw[threadIdx.x];
        double* set = d.cw + (size_t)((batch_id + 1) & 1) * kCwSlots * K;
        if (tot != 0.0) atomic_add_f64(set + (blockIdx.x % kCwSlots) * K + threadIdx.x, tot);
      }
    }
  }
  PHASE(10);
}

// ------------------------------ launchers ---------------------------------
int batched_max_classes() { return 64; }   // 17..64: sparse x only (binned form, a wavefront per draw)

// The LDS-privatised gather forms pin one workgroup per CU (their tables fill the LDS).  When the
// sample order is generated beside the epoch (solver_rng_*), its G workgroups need CUs of their own:
// a gather launch of 256 workgroups would otherwise wait for them and run a second round (C4: 930
// epochs/s with 256 + 32, 1055 with 224 + 32).  SGDNET_LDS_GRID overrides (experiments).
int lds_target_grid(const SagaDev& d) {
  static const int forced = exp_env_int("SGDNET_LDS_GRID", 0);
  if (forced > 0) return forced;
  const int cus = d.cu_budget > 0 && d.cu_budget < 256 ? d.cu_budget : 256;
  const int g = cus - d.cu_reserve;
  return g < 64 ? 64 : g;
}

// Launch geometry of the gather for a batch of m draws.  The LDS-privatised form
// needs the dense K*p table in LDS twice per CU (2 workgroups per CU) and enough
// draws per workgroup to amortise its flush.
struct GatherPlan {
  bool binned;  // range-binned form (tables that fit no LDS)
  bool lds;     // per-workgroup LDS copies of D flushed as slabs (sparse LDS form and dense form)
  bool dense;   // dense x: saga_batch_gather_dense_kernel
  bool tiled;   // dense x, table beyond LDS: gradient changes to d.gcb, D by feature tiles (global sweep)
  int chunks;   // tiled: draw chunks of the accumulate kernel (blockIdx.y)
  int draws_per_chunk;
  bool w_lds;   // K == 1: the coefficient snapshot is staged in LDS as well
  int grid;
  int draws_per_block;
  size_t lds_bytes;
};
constexpr size_t kLdsPerCu = 160 * 1024;        // gfx950
constexpr size_t kLdsStaticReserve = 2 * 1024;  // static __shared__ of the LDS gather kernels

static GatherPlan plan_gather(const SagaDev& d, int m) {
  GatherPlan g{};
  const size_t table = sizeof(double) * (size_t)d.K * (size_t)d.p;
  static const int force = [] {
    const char* e = exp_env_str("SGDNET_GATHER");   // "lds" | "global": experiments only
    return !e ? 0 : (e[0] == 'l' ? 1 : 2);
  }();
  const int target_grid = lds_target_grid(d);
  const bool fits = table <= 80 * 1024;
  if (d.xd) {   // dense x: wave per draw; LDS table + slabs, or the tiled form for larger tables
    g.dense = true;
    g.lds = d.slab != nullptr && fits;
    const int waves = kDenseBlock / 64;
    if (d.gcb && d.K <= 64 && (!fits || d.K > 16)) {     // 17..64 classes: always (the class-lane gather, round 4)
      g.tiled = true;
      int dpb = (m + 8191) / 8192;               // a row is >= 5 KB here: one or a few draws per wavefront
      dpb = (dpb + waves - 1) / waves * waves;
      if (dpb < waves) dpb = waves;
      g.draws_per_block = dpb;
      g.grid = (m + dpb - 1) / dpb;
      if (g.grid < 1) g.grid = 1;
      const int64_t tiles = (d.p + kTileF - 1) / kTileF;
      int64_t chunks = (2048 + tiles - 1) / tiles;   // ~2048 workgroups over the chip
      const int64_t most = (m + 4 * waves - 1) / (4 * waves);
      if (chunks > most) chunks = most;
      if (chunks < 1) chunks = 1;
      if (chunks > 65535) chunks = 65535;
      g.draws_per_chunk = (int)((m + chunks - 1) / chunks);
      g.chunks = (int)((m + g.draws_per_chunk - 1) / g.draws_per_chunk);
      return g;
    }
    int dpb = (m + target_grid - 1) / target_grid;
    dpb = (dpb + waves - 1) / waves * waves;
    if (dpb < waves) dpb = waves;
    g.draws_per_block = dpb;
    g.grid = (m + dpb - 1) / dpb;
    if (g.grid < 1) g.grid = 1;
    g.lds_bytes = table;
    return g;
  }
  // worthwhile once the batch's non-zeros outnumber the table ~48x: below that the fixed
  // cost of writing and re-reading one table per workgroup exceeds the atomics it saves
  if (d.R > 0 && d.bins && !d.force_global && force != 2 && m <= (1 << 20)) {
    g.binned = true;
    g.draws_per_block = kBinDraws;
    g.grid = (m + kBinDraws - 1) / kBinDraws;
    if (g.grid < 1) g.grid = 1;
    g.lds_bytes = sizeof(BinEntry) * (size_t)kBinEntCap + sizeof(unsigned) * (3 * (size_t)d.R + 1) +
                  ((sizeof(unsigned short) * ((size_t)d.n_coarse + 1) + 15) & ~size_t(15));
    return g;
  }
  const bool pays = (double)m * (double)d.avg_nnz >= 48.0 * (double)d.K * (double)d.p;
  g.lds = d.slab != nullptr && !d.force_global && fits && force != 2 && (force == 1 || pays);
  if (g.lds) {
    int dpb = (m + target_grid - 1) / target_grid;
    const int per_round = kLdsBlock / kGroup;
    if (dpb < per_round) dpb = per_round;
    g.draws_per_block = dpb;
    g.grid = (m + dpb - 1) / dpb;
    g.lds_bytes = table;
    static const bool w_lds_on = exp_env_int("SGDNET_W_LDS", 1) != 0;
    g.w_lds = d.K == 1 && w_lds_on && 2 * table + kLdsStaticReserve <= kLdsPerCu;
    if (g.w_lds) g.lds_bytes = 2 * table + 16;   // + alignment slack of the second table
  } else {
    g.draws_per_block = kBlock / kGroup;
    g.grid = (m + g.draws_per_block - 1) / g.draws_per_block;
  }
  if (g.grid < 1) g.grid = 1;
  return g;
}

int batch_gather_blocks(const SagaDev& d, int m) { return plan_gather(d, m).grid; }
bool binned_active(const SagaDev& d, int m) { return !d.xd && plan_gather(d, m).binned; }

// Doubles of slab storage the LDS-privatised gather needs for batches of m draws (0: the
// global-atomic form is used).
// the 8-lane K == 1 form reads two entries per lane: records must hold 16 entries
static bool lanes8_ok(const SagaDev& d) {
  static const int allow = exp_env_int("SGDNET_LANES8", 1);
  return allow && (d.cP || d.rec_cap >= kInReg8) && !SGD_ABLATE(d, ~0);
}

// Compact planes for a K == 1 sparse problem (d.ptr / d.idx / d.val / d.y resident).
bool compact_eligible(const SagaDev& d) {
  static const int allow = exp_env_int("SGDNET_COMPACT", 1);
  if (!allow || d.K != 1 || d.Ky != 1 || d.xd || !d.ptr || d.p > 65536) return false;
  if (2 * sizeof(double) * (size_t)d.p + 16 + kLdsStaticReserve > (size_t)kLdsPerCu) return false;  // no LDS form
  return (double)d.n * 2.0 * kCStride <= 48e9 && d.n < (int64_t)kIdMask;
}

// (a binomial response in another coding -- proportions, -1 / +1: sgdnet_solver_create does not forbid it, only
//  sgdnet_fit_* does -- keeps its value in the record: 11 entries)
int compact_entries(const SagaDev& d) { return d.family == SGDNET_BINOMIAL && d.y_binary ? 12 : 11; }

int launch_pack_compact(const SagaDev& d, char* P, char* Q, uint32_t* meta, hipStream_t st) {
  SGD_HIP_TRY(hipMemsetAsync(meta, 0, sizeof(uint32_t) * (size_t)((d.n + 15) / 16 + 1), st));
  int64_t grid = (d.n + 255) / 256;
  if (grid > 65536) grid = 65536;
  const int E = compact_entries(d);
  hipLaunchKernelGGL(pack_compact_kernel, dim3((unsigned)grid), dim3(256), 0, st, d.ptr, d.idx, d.val, d.y, d.n, E,
                     E == 12 ? 1 : 0, P, Q, meta);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_m_move(const SagaDev& d, int to_record, hipStream_t st) {
  int64_t grid = (d.n + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(m_move_kernel, dim3((unsigned)grid), dim3(256), 0, st, d, to_record);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int64_t batch_gather_slab_doubles(const SagaDev& d, int m) {
  SagaDev probe = d;
  probe.slab = reinterpret_cast<double*>(1);
  const GatherPlan g = plan_gather(probe, m);
  return g.lds ? (int64_t)g.grid * d.K * d.p : 0;
}

// ev0/ev1 (optional): dispatch start/stop timestamps of exactly this kernel
// (hipExtLaunchKernelGGL), used by the benchmark's per-kernel timing.
int launch_batch_gather(const SagaDev& d, LamParams* lam, int64_t t0_in_epoch, int m, int tail,
                        int batch_id_offset, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  (void)tail;
  const GatherPlan g = plan_gather(d, m);
  if (d.K > 16 && !g.binned && !g.tiled) {
    set_error("batched mode with more than 16 classes needs the binned form (sparse x) or the class-lane form (dense x), "
              "n_classes <= 64; got %d", d.K);
    return SGDNET_EUNSUPPORTED;
  }
  if (g.tiled && d.K > 16) {
    hipExtLaunchKernelGGL(saga_dense_cl_gather_kernel, dim3(g.grid), dim3(kDenseBlock), 0, st, ev0, nullptr, 0, d, lam,
                          t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    SGD_HIP_TRY(hipGetLastError());
    const dim3 agrid((unsigned)((d.p + kTileF - 1) / kTileF), (unsigned)g.chunks, (unsigned)((d.K + 15) / 16));
    hipExtLaunchKernelGGL(saga_dense_tiled_accumulate_kernel<16>, agrid, dim3(kDenseBlock), 0, st, nullptr, ev1, 0, d, lam,
                          t0_in_epoch, m, g.draws_per_chunk);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  if (g.tiled) {
    if (d.K == 1)
      hipExtLaunchKernelGGL((saga_batch_gather_dense_kernel<1, kDenseBlock, false, true>), dim3(g.grid), dim3(kDenseBlock),
                            0, st, ev0, nullptr, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else if (d.K <= 4)
      hipExtLaunchKernelGGL((saga_batch_gather_dense_kernel<4, kDenseBlock, false, true>), dim3(g.grid), dim3(kDenseBlock),
                            0, st, ev0, nullptr, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else
      hipExtLaunchKernelGGL((saga_batch_gather_dense_kernel<16, kDenseBlock, false, true>), dim3(g.grid), dim3(kDenseBlock),
                            0, st, ev0, nullptr, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    SGD_HIP_TRY(hipGetLastError());
    const dim3 agrid((unsigned)((d.p + kTileF - 1) / kTileF), (unsigned)g.chunks);
    if (d.K == 1)
      hipExtLaunchKernelGGL(saga_dense_tiled_accumulate_kernel<1>, agrid, dim3(kDenseBlock), 0, st, nullptr, ev1, 0, d,
                            lam, t0_in_epoch, m, g.draws_per_chunk);
    else if (d.K <= 4)
      hipExtLaunchKernelGGL(saga_dense_tiled_accumulate_kernel<4>, agrid, dim3(kDenseBlock), 0, st, nullptr, ev1, 0, d,
                            lam, t0_in_epoch, m, g.draws_per_chunk);
    else
      hipExtLaunchKernelGGL(saga_dense_tiled_accumulate_kernel<16>, agrid, dim3(kDenseBlock), 0, st, nullptr, ev1, 0, d,
                            lam, t0_in_epoch, m, g.draws_per_chunk);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  if (g.dense) {
    if (!g.lds) {
      set_error("batched mode on dense x with n_classes * n_features > 10240 needs n_classes <= 16 (tiled form)");
      return SGDNET_EUNSUPPORTED;
    }
    // function attributes are per device: one process may drive several GPUs (cv_sgdnet fan-out)
    static bool dense_attr_done_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& dense_attr_done = dense_attr_done_dev[cur_dev & 63];
    if (!dense_attr_done) {
      const int cap = 96 * 1024;
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_dense_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_dense_kernel<4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_dense_kernel<16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      dense_attr_done = true;
    }
    if (d.K == 1)
      hipExtLaunchKernelGGL(saga_batch_gather_dense_kernel<1>, dim3(g.grid), dim3(kDenseBlock), g.lds_bytes, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else if (d.K <= 4)
      hipExtLaunchKernelGGL(saga_batch_gather_dense_kernel<4>, dim3(g.grid), dim3(kDenseBlock), g.lds_bytes, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else
      hipExtLaunchKernelGGL(saga_batch_gather_dense_kernel<16>, dim3(g.grid), dim3(kDenseBlock), g.lds_bytes, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  if (g.binned) {
    static bool battr_done_dev[64] = {};
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (!battr_done_dev[cur & 63]) {
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_gather_kernel<16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_gather_kernel<16, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_gather_kernel<64>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_sweep_kernel<16, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRangeLdsBytes));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_sweep_kernel<16, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRangeLdsBytes));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_sweep_kernel<64, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRangeLdsBytes));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_binned_sweep_kernel<64, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRangeLdsBytes));
      battr_done_dev[cur & 63] = true;
    }
    if (d.K <= 16 && d.family == SGDNET_MULTINOMIAL)
      hipExtLaunchKernelGGL((saga_binned_gather_kernel<16, true>), dim3(g.grid), dim3(kBinBlock), g.lds_bytes, st, ev0, ev1, 0,
                            d, lam, t0_in_epoch, m, batch_id_offset);
    else if (d.K <= 16)
      hipExtLaunchKernelGGL(saga_binned_gather_kernel<16>, dim3(g.grid), dim3(kBinBlock), g.lds_bytes, st, ev0, ev1, 0,
                            d, lam, t0_in_epoch, m, batch_id_offset);
    else
      hipExtLaunchKernelGGL(saga_binned_gather_kernel<64>, dim3(g.grid), dim3(kBinBlock), g.lds_bytes, st, ev0, ev1, 0,
                            d, lam, t0_in_epoch, m, batch_id_offset);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  if (g.lds) {
    static bool attr_done_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& attr_done = attr_done_dev[cur_dev & 63];
    if (!attr_done) {
      const int cap = 96 * 1024;   // the dense table is limited to 80 KiB (plan_gather)
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<1, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kLdsPerCu - kLdsStaticReserve));
      SGD_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<1, true, false, kLanes8>),
          hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_cl_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      attr_done = true;
    }
    if (d.K == 1 && g.w_lds && lanes8_ok(d))
      hipExtLaunchKernelGGL((saga_batch_gather_lds_kernel<1, true, false, kLanes8>), dim3(g.grid), dim3(kLdsBlock),
                            g.lds_bytes, st, ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset,
                            g.draws_per_block);
    else if (d.K == 1 && g.w_lds)
      hipExtLaunchKernelGGL((saga_batch_gather_lds_kernel<1, true>), dim3(g.grid), dim3(kLdsBlock), g.lds_bytes,
                            st, ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else if (d.K == 1)
      hipExtLaunchKernelGGL(saga_batch_gather_lds_kernel<1>, dim3(g.grid), dim3(kLdsBlock), g.lds_bytes, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else if (d.K <= 4)
      hipExtLaunchKernelGGL(saga_batch_gather_lds_kernel<4>, dim3(g.grid), dim3(kLdsBlock), g.lds_bytes, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
    else
      hipExtLaunchKernelGGL(saga_batch_gather_cl_kernel<true>, dim3(g.grid), dim3(kLdsBlock), g.lds_bytes,
                            st, ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
  } else {
    if (d.K == 1)
      hipExtLaunchKernelGGL(saga_batch_gather_kernel<1>, dim3(g.grid), dim3(kBlock), 0, st, ev0, ev1, 0, d,
                            lam, t0_in_epoch, m, batch_id_offset);
    else if (d.K <= 4)
      hipExtLaunchKernelGGL(saga_batch_gather_kernel<4>, dim3(g.grid), dim3(kBlock), 0, st, ev0, ev1, 0, d,
                            lam, t0_in_epoch, m, batch_id_offset);
    else
      hipExtLaunchKernelGGL(saga_batch_gather_cl_kernel<false>, dim3(g.grid), dim3(kBlock), 0, st, ev0, ev1,
                            0, d, lam, t0_in_epoch, m, batch_id_offset, g.draws_per_block);
  }
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_cw_init(const SagaDev& d, const LamParams* lam, hipStream_t st) {
  hipLaunchKernelGGL(saga_cw_init_kernel, dim3(1), dim3(kBlock), 0, st, d, lam);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_batch_sweep(const SagaDev& d, LamParams* lam, int penalty, int tail, int m, int batch_id_offset,
                       hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, double ov_r, double ov_ls, double ov_m) {
  const GatherPlan g = plan_gather(d, m);
  // synchronous sharded mode (ov_m > 0): the slots were summed across ranks whose gather grids
  // may differ by one workgroup, so all of them are read (unused slots are zero)
  const int n_parts = ov_m > 0.0 ? kD0Slots : (g.grid < kD0Slots ? g.grid : kD0Slots);
  const SweepOverride ov{ov_r, ov_ls, ov_m};
  if (g.binned) {
    const size_t lds = sizeof(double) * (size_t)d.K * (size_t)d.range_max;
    const dim3 grid(d.R + 1), block(kRangeBlock);
    if (d.K <= 16 && penalty != SGDNET_GROUPLASSO)
      hipExtLaunchKernelGGL((saga_binned_sweep_kernel<16, false>), grid, block, lds, st, ev0, ev1, 0, d, lam, tail, n_parts,
                            batch_id_offset);
    else if (d.K <= 16)
      hipExtLaunchKernelGGL((saga_binned_sweep_kernel<16, true>), grid, block, lds, st, ev0, ev1, 0, d, lam, tail, n_parts,
                            batch_id_offset);
    else if (penalty != SGDNET_GROUPLASSO)
      hipExtLaunchKernelGGL((saga_binned_sweep_kernel<64, false>), grid, block, lds, st, ev0, ev1, 0, d, lam, tail, n_parts,
                            batch_id_offset);
    else
      hipExtLaunchKernelGGL((saga_binned_sweep_kernel<64, true>), grid, block, lds, st, ev0, ev1, 0, d, lam, tail, n_parts,
                            batch_id_offset);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  if (g.tiled && d.K > 16) {
    const int grid = (int)((d.p + kBlock / 64 - 1) / (kBlock / 64));
    hipExtLaunchKernelGGL(saga_dense_cl_sweep_kernel, dim3(grid < 1 ? 1 : grid), dim3(kBlock), 0, st, ev0, ev1, 0, d, lam,
                          tail, n_parts, batch_id_offset);
  } else if (g.lds) {
    const int F = kSlabElems / d.K;
    const int grid = (int)((d.p + F - 1) / F);
    hipExtLaunchKernelGGL(saga_batch_sweep_slab_kernel, dim3(grid < 1 ? 1 : grid), dim3(kBlock), 0, st, ev0,
                          ev1, 0, d, lam, tail, g.grid, batch_id_offset);
  } else if (penalty == SGDNET_GROUPLASSO) {
    const int grid = (int)((d.p + kBlock - 1) / kBlock);
    hipExtLaunchKernelGGL(saga_batch_sweep_kernel<true>, dim3(grid < 1 ? 1 : grid), dim3(kBlock), 0, st, ev0,
                          ev1, 0, d, lam, tail, n_parts, batch_id_offset, ov);
  } else {
    const int grid = (int)(((int64_t)d.K * d.p + kBlock - 1) / kBlock);
    hipExtLaunchKernelGGL(saga_batch_sweep_kernel<false>, dim3(grid < 1 ? 1 : grid), dim3(kBlock), 0, st, ev0,
                          ev1, 0, d, lam, tail, n_parts, batch_id_offset, ov);
  }
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// Virtual shards need the K == 1 LDS gather with w staged in LDS and a grid that splits evenly.
// Virtual shards need an LDS gather form and a grid that splits evenly: K == 1 with w staged in LDS (sparse or
// dense x), or 2..16 classes of sparse x whose K x p accumulator fits (round 3; the replica of w is read through L2).
bool vs_eligible(const SagaDev& d, int m) {
  (void)m;
  if (d.V < 2 || d.K < 1 || d.K > 16 || (d.standardize && !(d.vcw && d.c)) || d.force_global || !d.vw) return false;
  const size_t table = sizeof(double) * (size_t)d.K * (size_t)d.p;
  if (d.xd) return table <= 80 * 1024;                           // dense x (1..16 classes): only the accumulator is staged
  if (d.K > 1) return d.rec && table <= 80 * 1024;
  return 2 * table + 16 + kLdsStaticReserve <= kLdsPerCu;     // accumulator + coefficient snapshot in LDS
}

static int vs_grid(const SagaDev& d) { return d.v_bps * d.V; }

int launch_vs_broadcast(const SagaDev& d, hipStream_t st) {
  int grid = (int)((2 * (int64_t)d.K * d.p + 2 * d.K + kBlock - 1) / kBlock);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(saga_vs_broadcast_kernel, dim3(grid), dim3(kBlock), 0, st, d);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_vs_cw(const SagaDev& d, hipStream_t st) {
  if (!d.standardize) return SGDNET_OK;
  hipLaunchKernelGGL(saga_vs_cw_kernel, dim3(d.V * d.K), dim3(kBlock), 0, st, d);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_vs_merge(const SagaDev& d, int final_merge, hipStream_t st, LamParams* epoch_end, int batches) {
  int grid = (int)((2 * (int64_t)d.K * d.p + 2 * d.K + kBlock - 1) / kBlock);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(saga_vs_merge_kernel, dim3(grid), dim3(kBlock), 0, st, d, final_merge, epoch_end, batches);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_vs_gather(const SagaDev& d, LamParams* lam, int64_t t0_in_epoch, int m, hipStream_t st, hipEvent_t ev0,
                     hipEvent_t ev1, int batch_index) {
  const int grid = vs_grid(d);
  if (grid / d.V != d.v_bps || grid > kD0Slots) {
    set_error("internal: virtual-shard geometry (%d workgroups, %d per shard)", grid, d.v_bps);
    return SGDNET_EINVAL;
  }
  if (d.xd) {
    static bool dense_vs_attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    constexpr int kDenseVsBlock = 1024;           // 16 wavefronts share one LDS copy of the accumulator
    if (!dense_vs_attr_done[dev & 63]) {
      SGD_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void*>(saga_batch_gather_dense_kernel<1, kDenseVsBlock, true>),
          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_dense_kernel<4, kDenseBlock, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_dense_kernel<16, kDenseBlock, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      dense_vs_attr_done[dev & 63] = true;
    }
    const int waves = (d.K > 1 ? kDenseBlock : kDenseVsBlock) / 64;
    int dpb = (m + d.v_bps - 1) / d.v_bps;
    dpb = (dpb + waves - 1) / waves * waves;
    const size_t lds = sizeof(double) * (size_t)d.K * (size_t)d.p;
    if (d.K == 1)
      hipExtLaunchKernelGGL((saga_batch_gather_dense_kernel<1, kDenseVsBlock, true>), dim3(grid), dim3(kDenseVsBlock), lds, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, 0, dpb);
    // 2..16 classes (round 4): four wavefronts per workgroup (a draw holds per-class registers), the first-occurrence
    // claims of a sample are per batch (batch_id_offset = the batch's index in the epoch)
    else if (d.K <= 4)
      hipExtLaunchKernelGGL((saga_batch_gather_dense_kernel<4, kDenseBlock, true>), dim3(grid), dim3(kDenseBlock), lds, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_index, dpb);
    else
      hipExtLaunchKernelGGL((saga_batch_gather_dense_kernel<16, kDenseBlock, true>), dim3(grid), dim3(kDenseBlock), lds, st,
                            ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_index, dpb);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  int dpb = (m + d.v_bps - 1) / d.v_bps;
  const int per_round = kLdsBlock / kGroup;
  if (dpb < per_round) dpb = per_round;
  if (d.K > 1) {
    // 2..4 classes: the 16-lane draw of the LDS form against the shard's replica (batch_id_offset = the batch's
    // index in the epoch: the first-occurrence claims of a sample are per batch)
    static bool k4_attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!k4_attr_done[dev & 63]) {
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<4, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_cl_kernel<true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      k4_attr_done[dev & 63] = true;
    }
    if (d.K <= 4)
      hipExtLaunchKernelGGL((saga_batch_gather_lds_kernel<4, false, true>), dim3(grid), dim3(kLdsBlock),
                            sizeof(double) * (size_t)d.K * (size_t)d.p, st, ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_index,
                            dpb);
    else        // 5..16 classes: the class-lane form
      hipExtLaunchKernelGGL((saga_batch_gather_cl_kernel<true, true>), dim3(grid), dim3(kLdsBlock),
                            sizeof(double) * (size_t)d.K * (size_t)d.p, st, ev0, ev1, 0, d, lam, t0_in_epoch, m, batch_index,
                            dpb);
    SGD_HIP_TRY(hipGetLastError());
    return SGDNET_OK;
  }
  const size_t lds = 2 * sizeof(double) * (size_t)d.p + 16;
  static bool attr_done_dev[64] = {};
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  if (!attr_done_dev[cur_dev & 63]) {
    SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<1, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kLdsPerCu - kLdsStaticReserve));
    SGD_HIP_TRY(hipFuncSetAttribute(
        reinterpret_cast<const void*>(saga_batch_gather_lds_kernel<1, true, true, kLanes8>),
        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
    attr_done_dev[cur_dev & 63] = true;
  }
  if (lanes8_ok(d))
    hipExtLaunchKernelGGL((saga_batch_gather_lds_kernel<1, true, true, kLanes8>), dim3(grid), dim3(kLdsBlock), lds,
                          st, ev0, ev1, 0, d, lam, t0_in_epoch, m, 0, dpb);
  else
    hipExtLaunchKernelGGL((saga_batch_gather_lds_kernel<1, true, true>), dim3(grid), dim3(kLdsBlock), lds, st, ev0,
                          ev1, 0, d, lam, t0_in_epoch, m, 0, dpb);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_vs_sweep(const SagaDev& d, LamParams* lam, int tail, int m, hipStream_t st, hipEvent_t ev0,
                    hipEvent_t ev1) {
  (void)m;
  const int F = kSlabElems / d.K;            // features per block
  const int nfb = (int)((d.p + F - 1) / F);
  if (d.K == 1)
    hipExtLaunchKernelGGL(saga_vs_sweep_kernel<1>, dim3(nfb * d.V), dim3(kBlock), 0, st, ev0, ev1, 0, d, lam, tail, nfb);
  else if (d.K <= 4)
    hipExtLaunchKernelGGL(saga_vs_sweep_kernel<4>, dim3(nfb * d.V), dim3(kBlock), 0, st, ev0, ev1, 0, d, lam, tail, nfb);
  else
    hipExtLaunchKernelGGL(saga_vs_sweep_kernel<16>, dim3(nfb * d.V), dim3(kBlock), 0, st, ev0, ev1, 0, d, lam, tail, nfb);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

// ---- the fused epoch of the virtual shards (saga_vs_epoch_kernel) ----
static int64_t fused_slice(const SagaDev& d) { return 2 * ((d.p + 2 * (int64_t)d.v_bps - 1) / (2 * (int64_t)d.v_bps)); }
static size_t fused_lds_bytes(const SagaDev& d) {
  const int64_t part = (int64_t)(kLdsBlock / 64) * fused_slice(d);     // the slice sweep's per-wavefront partial sums
  return sizeof(double) * (size_t)(d.p + (part > d.p ? part : d.p)) + 16;
}
size_t vs_fused_sync_words() { return (size_t)(kSyncLines + 2) * kSyncLine; }
size_t vs_fused_sync_sticky_word() { return (size_t)kSyncSticky * kSyncLine; }
size_t vs_fused_col_words() { return (size_t)kFusedMaxBps * kSyncLine; }
// local: V reference copies [g_sum | w | g_sum_b | b] + V x 128 c.w partials; published: 2 parities x V slices
size_t vs_fused_exchange_doubles(const SagaDev& d, int n_shards) {
  return (size_t)n_shards * (size_t)(2 * d.p + 2) + (size_t)n_shards * kFusedMaxBps;
}
size_t vs_fused_publish_doubles(const SagaDev& d, int n_shards) { return (size_t)2 * n_shards * (size_t)(2 * d.p + 2); }

// sparse x, one response, compact records, an even number of features, slices of at most 384 features
bool vs_fused_eligible(const SagaDev& d) {
  if (!vs_eligible(d, 0) || d.K != 1 || d.xd || !d.cP || !lanes8_ok(d) || (d.p & 1) || !d.vsync || !d.vx || !d.vcol || !d.vpub) return false;
  if (d.v_bps < 1 || d.v_bps > kFusedMaxBps || d.V * d.v_bps > 1024) return false;
  if (fused_slice(d) > 2 * 64 * kFusedChunks) return false;
  if ((int64_t)d.V * d.v_bps * d.p * 8 >= (1ll << 31)) return false;
  return fused_lds_bytes(d) + kLdsStaticReserve <= kLdsPerCu;
}

// workgroups added to the launch for the sample-order generators (one per reserved CU)
int vs_fused_rng_workgroups(const SagaDev& d) { return d.rngdev ? d.cu_reserve : 0; }

int launch_vs_epoch(const SagaDev& d, LamParams* lam, int nb, int every, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  static bool attr_done_dev[64] = {};
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  if (!attr_done_dev[cur_dev & 63]) {
    SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_vs_epoch_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
    SGD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(saga_vs_epoch_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu - kLdsStaticReserve));
    attr_done_dev[cur_dev & 63] = true;
  }
  if (nb < 1 || every < 1) return SGDNET_EINVAL;
  const int rng_wgs = vs_fused_rng_workgroups(d);
  size_t lds = fused_lds_bytes(d);
  if (rng_wgs > 0 && lds < kJumpLds) lds = kJumpLds;
  if (d.n_peers > 1 && d.peers)
    hipExtLaunchKernelGGL(saga_vs_epoch_kernel<true>, dim3(vs_grid(d) + rng_wgs), dim3(kLdsBlock), lds, st, ev0, ev1, 0, d,
                          lam, nb, every);
  else
    hipExtLaunchKernelGGL(saga_vs_epoch_kernel<false>, dim3(vs_grid(d) + rng_wgs), dim3(kLdsBlock), lds, st, ev0, ev1, 0, d,
                          lam, nb, every);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_epoch_end(LamParams* lam, int batches, hipStream_t st) {
  hipLaunchKernelGGL(saga_epoch_end_kernel, dim3(1), dim3(64), 0, st, lam, batches);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_convergence(const SagaDev& d, LamParams* lam, hipStream_t st) {
  const int64_t len = (int64_t)d.K * d.p;
  int grid = (int)((len + kBlock * 4 - 1) / (kBlock * 4));
  if (grid < 1) grid = 1;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(saga_convergence_kernel, dim3(grid), dim3(kBlock), 0, st, d, lam);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_loss(const SagaDev& d, LamParams* lam, bool sparse, hipStream_t st) {
  const int64_t groups = d.n;
  int grid = (int)((groups + (kBlock / kGroup) * 8 - 1) / ((kBlock / kGroup) * 8));
  if (grid < 1) grid = 1;
  if (grid > 4096) grid = 4096;
  const size_t lds = sizeof(double) * ((size_t)(kBlock / kGroup) + 1) * (size_t)d.K;
  if (sparse)
    hipLaunchKernelGGL(saga_loss_kernel<true>, dim3(grid), dim3(kBlock), lds, st, d, lam);
  else
    hipLaunchKernelGGL(saga_loss_kernel<false>, dim3(grid), dim3(kBlock), lds, st, d, lam);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_delta_export(const SagaDev& d, const double* ref, double* out, double weight, hipStream_t st) {
  const int64_t len = 2 * (int64_t)d.K * d.p + 2 * d.K;
  int grid = (int)((len + kBlock - 1) / kBlock);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(saga_delta_export_kernel, dim3(grid), dim3(kBlock), 0, st, d, ref, out, weight);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

int launch_delta_apply(const SagaDev& d, double* ref, const double* merged, double w_weight, hipStream_t st) {
  const int64_t len = 2 * (int64_t)d.K * d.p + 2 * d.K;
  int grid = (int)((len + kBlock - 1) / kBlock);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(saga_delta_apply_kernel, dim3(grid), dim3(kBlock), 0, st, d, ref, merged, w_weight);
  SGD_HIP_TRY(hipGetLastError());
  return SGDNET_OK;
}

}  // namespace sgdnet
