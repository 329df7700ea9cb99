// Floor of the K = 1 batched gather's compulsory memory pattern on gfx950:
// per draw ONE random 128-byte record (8 lanes x 16 B, like the compact plane P of
// saga_batch_gather_lds_kernel<1,true,true,8>) out of a 1.28 GB table and ONE returning
// device-scope 8-byte exchange on a random entry of an 80 MB table (the gradient memory),
// nothing else: no x.w, no exp, no LDS scatter, no slab.  Same launch geometry as the product
// kernel (256 workgroups x 1024 threads, 1 048 576 draws per launch) and, as a second figure, the
// geometry that gives this pattern the most parallelism.  Sample ids come from a resident
// uint32 stream (4 B per draw, coalesced), as in the product.
//
//   hipcc -O3 --offload-arch=gfx950 gather_exchange.hip -o gather_exchange && ./gather_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s failed %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// mode bit 0: record read, bit 1: exchange.  U draws in flight per 8-lane group.
template <int U, int MODE>
__global__ __launch_bounds__(1024) void k(const char* rec, double* M, const unsigned* stream, int draws_per_group,
                                          double* sink) {
  const int gl = threadIdx.x & 7;
  const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const unsigned* my = stream + group * draws_per_group;
  double acc = 0.0;
  for (int it = 0; it < draws_per_group; it += U) {
    unsigned s[U];
    double2 v[U];
    double old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = my[it + u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE & 1) v[u] = *reinterpret_cast<const double2*>(rec + (size_t)s[u] * 128 + 16 * gl);
      else v[u] = double2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      old[u] = 0.0;
      if ((MODE & 2) && gl == 0) {
        const double nv = (MODE & 1) ? v[u].x + 1.0 : (double)s[u];       // depends on the record, like the gradient
        old[u] = __longlong_as_double((long long)__hip_atomic_exchange(
            reinterpret_cast<unsigned long long*>(M + s[u]), (unsigned long long)__double_as_longlong(nv),
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + old[u];
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int U, int MODE>
double run(const char* name, const char* rec, double* M, const unsigned* stream, long draws, int blocks, double* sink,
           hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long groups = (long)blocks * 1024 / 8;
  const int dpg = (int)(draws / groups);
  k<U, MODE><<<blocks, 1024, 0, st>>>(rec, M, stream, dpg, sink);
  CK(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    k<U, MODE><<<blocks, 1024, 0, st>>>(rec, M, stream, dpg, sink);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double n = (double)groups * dpg;
  printf("%-44s blocks %4d U=%d: %8.2f us per %.0f draws  %6.2f G draws/s  = %6.0f GB/s at 152 B per draw (%4.1f %% of 8 TB/s)\n",
         name, blocks, U, best * 1e3, n, n / best / 1e6, n * 152 / best / 1e6, n * 152 / best / 1e6 / 80.0);
  return best * 1e3;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t n = 10000000;                 // config 4: 10M samples
  char* rec;
  double* M;
  unsigned* stream;
  double* sink;
  CK(hipMalloc(&rec, n * 128));
  CK(hipMemset(rec, 0, n * 128));
  CK(hipMalloc(&M, n * 8));
  CK(hipMemset(M, 0, n * 8));
  CK(hipMalloc(&sink, 8));
  const long draws = 1 << 20;                // one product launch: 8 virtual shards x 131072 draws
  std::vector<unsigned> h(draws);
  unsigned x = 12345u;
  for (long i = 0; i < draws; ++i) {         // with replacement, like the product's sample order
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    h[i] = (unsigned)(((unsigned long long)x * n) >> 32);
  }
  CK(hipMalloc(&stream, draws * 4));
  CK(hipMemcpy(stream, h.data(), draws * 4, hipMemcpyHostToDevice));
  printf("# product geometry: 256 workgroups x 1024 threads, 8 lanes per draw, draws per group consecutive in the stream\n");
  run<4, 1>("records only", rec, M, stream, draws, 256, sink, st);
  run<4, 2>("exchanges only", rec, M, stream, draws, 256, sink, st);
  run<4, 3>("record + dependent exchange", rec, M, stream, draws, 256, sink, st);
  run<8, 3>("record + dependent exchange", rec, M, stream, draws, 256, sink, st);
  run<16, 3>("record + dependent exchange", rec, M, stream, draws, 256, sink, st);
  printf("# two workgroups per CU (not available to the product kernel: its LDS tables pin one per CU)\n");
  run<4, 3>("record + dependent exchange", rec, M, stream, draws, 512, sink, st);
  run<8, 3>("record + dependent exchange", rec, M, stream, draws, 512, sink, st);
  run<8, 1>("records only", rec, M, stream, draws, 512, sink, st);
  run<8, 2>("exchanges only", rec, M, stream, draws, 512, sink, st);
  return 0;
}
