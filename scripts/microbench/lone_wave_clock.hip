// What shader clock does a lone wavefront get?  One wavefront runs a dependent chain of f64 FMAs and reads both
// clock64() (shader cycles) and wall_clock64() (100 MHz): cycles per FMA and the effective MHz -- alone, and with a
// "heater" (many workgroups of independent FMAs on another stream) running beside it.
//   hipcc --offload-arch=gfx950 -O3 -o lone_wave_clock lone_wave_clock.hip && ./lone_wave_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void chain(double* out, unsigned long long* t, int n) {
  double a = out[0], b = 1.0000001, c = 1e-9;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a = __builtin_fma(a, b, c);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) {
    t[0] = c1 - c0;
    t[1] = w1 - w0;
  }
}

__global__ void heater(double* out, const int* stop, int rounds) {
  double a = threadIdx.x, b = 1.0000001, c = 1e-9, e = 0.5, f = 0.25;
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < 4096; ++i) {
      a = __builtin_fma(a, b, c);
      e = __builtin_fma(e, b, c);
      f = __builtin_fma(f, b, c);
    }
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
  }
  if (a + e + f == 12345.0) out[0] = a;
}

int main() {
  double* out;
  unsigned long long* t;
  int* stop;
  CK(hipMalloc(&out, 4096 * sizeof(double)));
  CK(hipMalloc(&t, 2 * sizeof(unsigned long long)));
  CK(hipMalloc(&stop, sizeof(int)));
  CK(hipMemset(out, 0, 4096 * sizeof(double)));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int n = 1 << 20;      // 16M dependent FMAs
  for (int heat = 0; heat <= 3; ++heat) {
    const int wgs = heat == 0 ? 0 : heat == 1 ? 8 : heat == 2 ? 64 : 248;
    CK(hipMemset(stop, 0, sizeof(int)));
    if (wgs) hipLaunchKernelGGL(heater, dim3(wgs), dim3(256), 0, s2, out + 64, stop, 1 << 14);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, out, t, n);
      CK(hipStreamSynchronize(s1));
      unsigned long long h[2];
      CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
      printf("heater workgroups %3d: %.2f shader cycles per dependent f64 FMA, %.3f ns per FMA, shader clock %.0f MHz\n", wgs,
             (double)h[0] / (16.0 * n), (double)h[1] * 10.0 / (16.0 * n), (double)h[0] / ((double)h[1] * 10.0) * 1e3);
    }
    int one = 1;
    CK(hipMemcpy(stop, &one, sizeof(int), hipMemcpyHostToDevice));
    CK(hipStreamSynchronize(s2));
  }
  return 0;
}
