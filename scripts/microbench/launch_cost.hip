// What does a dispatch cost by its shape alone?  Empty kernels (one store per workgroup) launched back to back in one
// stream, 200 per shape, timed with events: threads per workgroup x workgroups x dynamic LDS -- the shapes of the
// binned pair of config 5 among them (gather 256 x 1024 threads with 143 KB, sweep 501 x 512 threads with 64 KB).
//   hipcc --offload-arch=gfx950 -O3 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(int* out) {
  extern __shared__ int lds[];
  if (threadIdx.x == 0) {
    lds[0] = (int)blockIdx.x;
    out[blockIdx.x] = lds[0];
  }
}

int main() {
  int* out;
  CK(hipMalloc(&out, 4096 * sizeof(int)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Shape { int grid, block, lds; } shapes[] = {
      {1, 64, 0}, {256, 256, 0}, {256, 1024, 0}, {256, 1024, 143 * 1024}, {501, 512, 0}, {501, 512, 64 * 1024}, {501, 512, 32 * 1024},
      {2048, 256, 0}};
  for (const Shape& s : shapes) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(s.grid), dim3(s.block), s.lds, st, out);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1)
        printf("%4d workgroups x %4d threads, %3d KB of LDS: %.2f us per dispatch (back to back, one stream)\n", s.grid, s.block,
               s.lds / 1024, ms * 1e3 / 200);
    }
  }
  return 0;
}
