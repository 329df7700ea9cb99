// Floors of candidate formulations of the K = 1 batched gather's gradient-memory traffic on
// gfx950.  Product geometry throughout: 256 workgroups x 1024 threads,
// 8 lanes per draw, 1 048 576 draws per launch, one random 128-byte record per draw out of a
// 1.28 GB table.  What varies is how the old gradient of the drawn sample is read and how the new
// one is written:
//
//   rec            records only
//   xchg           + ONE dependent returning device-scope exchange on an 80 MB table (round 2's kernel)
//   ld             + an INDEPENDENT plain 8-byte load from the 80 MB table, nothing written
//   ld+log         + the same load, new value appended to a log in draw order (coalesced 8-byte stores)
//   ring+log       + an independent 8-byte load from a 640 MB ring (the logs of 8 epochs) at the position
//                    of the sample's previous draw, new value appended in draw order
//   ld+st          + the independent load and a fire-and-forget plain scattered 8-byte store
//   inrec+st       old value inside the record (free), plain scattered 8-byte store into the record
//
//   hipcc -O3 --offload-arch=gfx950 gather_patterns.hip -o gather_patterns && ./gather_patterns
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s failed %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

enum { REC = 1, XCHG = 2, LD = 4, LOG = 8, RING = 16, ST = 32, INREC = 64, NT = 128, WT = 256, NTST = 512, XREC = 1024, XRECW = 2048 };

struct Args {
  char* rec;
  double* M;
  double* ring;          // 8 epochs of logs
  double* log;           // this launch's region of the ring
  const unsigned* stream;
  const unsigned* prev;  // ring position of the previous draw of the same sample
  double* sink;
  int draws_per_group;
};

template <int U, int MODE>
__global__ __launch_bounds__(1024) void k(Args a) {
  typedef double dpair_t __attribute__((ext_vector_type(2)));
  const int gl = threadIdx.x & 7;
  const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const size_t base = group * a.draws_per_group;
  const unsigned* my = a.stream + base;
  const unsigned* myp = a.prev + base;
  double acc = 0.0;
  for (int it = 0; it < a.draws_per_group; it += U) {
    unsigned s[U], pv[U];
    dpair_t v[U];
    double old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = my[it + u];
      if (MODE & RING) pv[u] = myp[it + u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const dpair_t* src = reinterpret_cast<const dpair_t*>(a.rec + (size_t)s[u] * 128 + 16 * gl);
      if (MODE & REC) v[u] = (MODE & NT) ? __builtin_nontemporal_load(src) : *src;
      else v[u] = dpair_t{0.0, 0.0};
      old[u] = 0.0;
      if (gl == 0) {
        if (MODE & RING) old[u] = a.ring[pv[u]];
        else if (MODE & LD) old[u] = a.M[s[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (gl == 0) {
        const double nv = (MODE & REC) ? v[u].x + 1.0 : (double)s[u];   // depends on the record, like the gradient
        if (MODE & XCHG)
          old[u] = __longlong_as_double((long long)__hip_atomic_exchange(
              reinterpret_cast<unsigned long long*>(a.M + s[u]), (unsigned long long)__double_as_longlong(nv),
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (MODE & XREC)                       // the returning exchange aimed at the last 8 bytes of the record just read
          old[u] = __longlong_as_double((long long)__hip_atomic_exchange(
              reinterpret_cast<unsigned long long*>(a.rec + (size_t)s[u] * 128 + 120), (unsigned long long)__double_as_longlong(nv),
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (MODE & XRECW)                      // the same exchange at workgroup scope: performed in this XCD's L2 (valid
                                               // only if every draw of a sample runs on one XCD)
          old[u] = __longlong_as_double((long long)__hip_atomic_exchange(
              reinterpret_cast<unsigned long long*>(a.rec + (size_t)s[u] * 128 + 120), (unsigned long long)__double_as_longlong(nv),
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (MODE & LOG) a.log[base + it + u] = nv + old[u];
        if (MODE & ST) a.M[s[u]] = nv + old[u];
        if (MODE & INREC) {
          double* dst = reinterpret_cast<double*>(a.rec + (size_t)s[u] * 128 + 120);
          if (MODE & WT) __hip_atomic_store(dst, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through
          else if (MODE & NTST) __builtin_nontemporal_store(nv, dst);
          else *dst = nv;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + old[u];
  }
  if (acc == 12345.678) a.sink[0] = acc;
}

template <int U, int MODE>
double run(const char* name, Args a, long draws, int blocks, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long groups = (long)blocks * 1024 / 8;
  a.draws_per_group = (int)(draws / groups);
  k<U, MODE><<<blocks, 1024, 0, st>>>(a);
  CK(hipStreamSynchronize(st));
  float best = 1e30f, sum = 0.f;
  const int reps = 7;
  const unsigned* stream0 = a.stream;
  const unsigned* prev0 = a.prev;
  for (int rep = 0; rep < reps; ++rep) {
    // a fresh segment of the stream per repetition: nothing a launch reads is left in the Infinity
    // Cache by the launch before it (the first version of this benchmark replayed one segment)
    a.stream = stream0 + (size_t)(rep + 1) * draws;
    a.prev = prev0 + (size_t)(rep + 1) * draws;
    CK(hipEventRecord(e0, st));
    k<U, MODE><<<blocks, 1024, 0, st>>>(a);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double n = (double)groups * a.draws_per_group;
  printf("%-40s U=%-2d best %7.2f us  mean %7.2f us per %.0f draws  %6.2f G draws/s = %4.1f %% of 8 TB/s at 152 B per draw\n",
         name, U, best * 1e3, sum / reps * 1e3, n, n / best / 1e6, n * 152 / best / 1e6 / 80.0);
  return best * 1e3;
}

int main(int argc, char** argv) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t n = 10000000;                 // config 4: 10M samples
  const size_t ring_n = 8 * n;
  Args a;
  CK(hipMalloc(&a.rec, n * 128));
  CK(hipMemset(a.rec, 0, n * 128));
  CK(hipMalloc(&a.M, n * 8));
  CK(hipMemset(a.M, 0, n * 8));
  CK(hipMalloc(&a.ring, ring_n * 8));
  CK(hipMemset(a.ring, 0, ring_n * 8));
  CK(hipMalloc(&a.sink, 8));
  const long draws = 1 << 20;                // one product launch: 8 virtual shards x 131072 draws
  const long total = draws * 8;              // 1 warm-up segment + 7 timed ones
  a.log = a.ring + 5 * n;                    // somewhere in the ring
  std::vector<unsigned> h(total), hp(total), hp_recent(total), hs(total);
  unsigned x = 12345u;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
  for (long i = 0; i < total; ++i) {         // with replacement, like the product's sample order
    h[i] = (unsigned)(((unsigned long long)rnd() * n) >> 32);
    hp[i] = (unsigned)(((unsigned long long)rnd() * ring_n) >> 32);          // uniform over 8 epochs
    // realistic: the previous draw of a sample is Exp(mean n) draws back
    const double u = (rnd() + 0.5) / 4294967296.0;
    double back = -std::log(u) * (double)n;
    if (back > 7.0 * n) back = 7.0 * n;
    hp_recent[i] = (unsigned)((5 * n + (size_t)i + ring_n - (size_t)back) % ring_n);
  }
  // the same draws sorted inside each shard-batch of 131072 (order inside a batch is free)
  hs = h;
  for (long b = 0; b < total; b += 131072) std::sort(hs.begin() + b, hs.begin() + b + 131072);
  unsigned *stream, *sorted, *prev, *prev_recent;
  CK(hipMalloc(&stream, total * 4));
  CK(hipMalloc(&sorted, total * 4));
  CK(hipMalloc(&prev, total * 4));
  CK(hipMalloc(&prev_recent, total * 4));
  CK(hipMemcpy(stream, h.data(), total * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sorted, hs.data(), total * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(prev, hp.data(), total * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(prev_recent, hp_recent.data(), total * 4, hipMemcpyHostToDevice));
  a.stream = stream;
  a.prev = prev;
  printf("# 256 workgroups x 1024 threads, 8 lanes per draw, U draws in flight per group\n");
  run<4, REC>("rec", a, draws, 256, st);
  run<8, REC>("rec", a, draws, 256, st);
  run<4, REC | XCHG>("rec + dependent xchg (round 2)", a, draws, 256, st);
  run<4, REC | LD>("rec + ld", a, draws, 256, st);
  run<8, REC | LD>("rec + ld", a, draws, 256, st);
  run<4, REC | LD | LOG>("rec + ld + log", a, draws, 256, st);
  run<8, REC | LD | LOG>("rec + ld + log", a, draws, 256, st);
  run<4, REC | RING | LOG>("rec + ring(uniform 8 epochs) + log", a, draws, 256, st);
  run<8, REC | RING | LOG>("rec + ring(uniform 8 epochs) + log", a, draws, 256, st);
  a.prev = prev_recent;
  run<4, REC | RING | LOG>("rec + ring(exp back) + log", a, draws, 256, st);
  run<8, REC | RING | LOG>("rec + ring(exp back) + log", a, draws, 256, st);
  run<8, REC | RING | LOG | NT>("rec(nt) + ring(exp back) + log", a, draws, 256, st);
  run<16, REC | RING | LOG>("rec + ring(exp back) + log", a, draws, 256, st);
  a.prev = prev;
  run<4, REC | LD | ST>("rec + ld + scattered st", a, draws, 256, st);
  run<4, REC | INREC>("rec + st into the record", a, draws, 256, st);
  run<4, REC | XREC>("rec + dependent xchg INTO the record", a, draws, 256, st);
  run<8, REC | XREC>("rec + dependent xchg INTO the record", a, draws, 256, st);
  run<4, REC | XRECW>("rec + xchg INTO the record, workgroup scope", a, draws, 256, st);
  run<8, REC | XRECW>("rec + xchg INTO the record, workgroup scope", a, draws, 256, st);
  run<4, REC | XREC>("rec + dependent xchg INTO the record (again)", a, draws, 256, st);
  run<4, REC | XCHG>("rec + dependent xchg (again)", a, draws, 256, st);
  run<4, REC | INREC>("rec + st into the record (again)", a, draws, 256, st);
  run<4, REC | INREC | WT>("rec + st into the record (sc0 sc1)", a, draws, 256, st);
  run<4, REC | INREC | NTST>("rec + st into the record (nt)", a, draws, 256, st);
  run<8, REC | INREC>("rec + st into the record", a, draws, 256, st);
  run<4, LD>("ld only", a, draws, 256, st);
  printf("# half the chip (128 workgroups): is the random-access rate a per-CU or a chip limit?\n");
  run<4, REC>("rec, 128 workgroups", a, draws, 128, st);
  run<8, REC>("rec, 128 workgroups", a, draws, 128, st);
  run<4, REC | INREC>("rec + st into the record, 128 workgroups", a, draws, 128, st);
  run<4, RING>("ring only", a, draws, 256, st);
  printf("# draws sorted inside each 131072-draw shard-batch\n");
  a.stream = sorted;
  run<4, REC>("sorted: rec", a, draws, 256, st);
  run<4, REC | XCHG>("sorted: rec + dependent xchg", a, draws, 256, st);
  run<4, REC | LD | ST>("sorted: rec + ld + st", a, draws, 256, st);
  run<4, REC | INREC>("sorted: rec + st into the record", a, draws, 256, st);
  printf("# two workgroups per CU\n");
  a.stream = stream;
  a.prev = prev_recent;
  run<4, REC | RING | LOG>("rec + ring(exp back) + log", a, draws, 512, st);
  run<8, REC | RING | LOG>("rec + ring(exp back) + log", a, draws, 512, st);
  return 0;
}
