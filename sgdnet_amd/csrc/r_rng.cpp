// R-compatible Mersenne-Twister for the built-in sample order.
//
// The reference draws one sample index per inner iteration with
// floor(R::runif(0, n_samples)) (src/saga-sparse.h:261, src/saga-dense.h:152)
// under Rcpp::RNGScope (src/RcppExports.cpp:14,27).  With R's default
// RNGkind this is MT19937 seeded by set.seed()'s linear-congruential
// scrambling; reproducing it lets a fit driven from Python or C consume the
// very stream `set.seed(s); sgdnet(...)` consumes.
#include <math.h>

#include "sgdnet_hip.h"

extern "C" {

void sgdnet_rng_seed(sgdnet_rng* r, uint32_t seed) {
  for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;   // initial scrambling
  seed = 69069u * seed + 1u;                                // i_seed[0] (overwritten: mti = 624)
  for (int j = 0; j < 624; ++j) {
    seed = 69069u * seed + 1u;
    r->mt[j] = seed;
  }
  r->mti = 624;
}

static inline uint32_t mt_word(sgdnet_rng* r) {
  constexpr int N = 624, M = 397;
  constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrixA = 0x9908b0dfu;
  if (r->mti >= (uint32_t)N) {
    uint32_t* mt = r->mt;
    for (int k = 0; k < N; ++k) {
      const uint32_t y = (mt[k] & kUpper) | (mt[(k + 1) % N] & kLower);
      mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
    }
    r->mti = 0;
  }
  uint32_t y = r->mt[r->mti++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

double sgdnet_rng_unif(sgdnet_rng* r) {
  // unif_rand(): MT output scaled to [0,1), then pushed inside (0,1)
  const double i2_32m1 = 2.328306437080797e-10;
  const double x = (double)mt_word(r) * 2.3283064365386963e-10;
  if (x <= 0.0) return 0.5 * i2_32m1;
  if (1.0 - x <= 0.0) return 1.0 - 0.5 * i2_32m1;
  return x;
}

void sgdnet_rng_fill(sgdnet_rng* r, uint32_t n_samples, uint32_t* out, int64_t count) {
  const double n = (double)n_samples;
  for (int64_t i = 0; i < count; ++i) {
    double u;
    do {
      u = sgdnet_rng_unif(r);
    } while (u <= 0.0 || u >= 1.0);
    out[i] = (uint32_t)floor(0.0 + (n - 0.0) * u);
  }
}

}  // extern "C"
