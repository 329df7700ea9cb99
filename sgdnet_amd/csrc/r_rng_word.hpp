// One word of R's Mersenne-Twister -> one draw of the sample order (device side).
#pragma once

#include <stdint.h>

#include <hip/hip_runtime.h>

namespace sgdnet {

// tempering + unif_rand() scaling/fixup + floor(n * u), as r_rng.cpp does on the host:
// floor(R::runif(0, n)) of src/saga-sparse.h:261 / src/saga-dense.h:152
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

}  // namespace sgdnet
