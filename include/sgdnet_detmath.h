/* sgdnet_detmath.h -- exp() and log() as plain IEEE-754 double arithmetic: + - * / and bit moves only,
 * no FMA contraction, no libm.  The same source compiled for the host (gcc/clang, -ffp-contract=off) and
 * for gfx950 (hipcc -ffp-contract=off) returns the SAME BITS for every argument.
 *
 * Why: in exact mode the HIP kernels replay the reference iteration draw by draw (saga_exact.hip).  All of
 * it is IEEE arithmetic in the reference's order except the family gradients' exp/log (src/families.h:161-168,
 * 244-260, src/math.h:25-33), where the device math library and glibc differ in the last bit now and then --
 * enough to move a stopping epoch when a coefficient flickers around the soft threshold at lambda_max
 * (DESIGN.md 4.1), after which the two runs consume different draws.  With these two functions on both
 * sides the exact kernels are bit-identical to the CPU restatement built with -DORC_DET_MATH.
 *
 * Accuracy: table-driven in the manner of Tang (ACM TOMS 15(2) 1989, 16(4) 1990): exp < 1 ulp, log < 1.5 ulp
 * (checked against libm on 10^7 arguments in tests/test_detmath.py) -- the accuracy class of the libms the
 * reference runs on; the oracle built on these functions still reproduces the outputs R printed
 * (tests/test_refdocs_oracle.py).
 */
#ifndef SGDNET_DETMATH_H_
#define SGDNET_DETMATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define SGD_DM_FN __host__ __device__ static inline
/* the forms that read the built-in tables: device only in a HIP translation unit (the tables are device symbols
 * there, a host call would have nothing to read -- and does not compile) */
#define SGD_DM_TABFN __device__ static inline
#ifndef SGD_TABLE_QUAL
#define SGD_TABLE_QUAL static __device__ const
#endif
#else
#define SGD_DM_FN static inline
#define SGD_DM_TABFN static inline
#ifndef SGD_TABLE_QUAL
#define SGD_TABLE_QUAL static const
#endif
#endif

#include "sgdnet_detmath_tables.h"

#define SGD_EXP_TABPTR (&sgd_exp_tab[0][0])
#define SGD_LOG_TABPTR (&sgd_log_tab[0][0])

SGD_DM_FN uint64_t sgd_d2u(double x) {
  uint64_t u;
  memcpy(&u, &x, sizeof(u));
  return u;
}
SGD_DM_FN double sgd_u2d(uint64_t u) {
  double x;
  memcpy(&x, &u, sizeof(x));
  return x;
}

/* `tab`: sgd_exp_tab as 128 consecutive doubles, or a copy of it (a kernel may keep one in on-chip memory, under
 * its own pointer type: SGD_DEFINE_EXP(name, pointer type) writes the same function for it) */
#define SGD_DEFINE_EXP(NAME, TABPTR_T) \
SGD_DM_FN double NAME(double x, TABPTR_T tab) { \
  if (x != x) return x; \
  if (x > 709.782712893384) return sgd_u2d(0x7ff0000000000000ull);      /* +inf */ \
  if (x < -745.2) return 0.0; \
  const double ax = x < 0.0 ? -x : x; \
  if (ax < 5.551115123125783e-17) return 1.0 + x;                          /* 2^-54 */ \
  /* x = (64 m + j) ln2/64 + r */ \
  const double magic = 6755399441055744.0;                                 /* 1.5 * 2^52: round to nearest integer */ \
  const double t = x * SGD_EXP_INVL + magic; \
  const int32_t N = (int32_t)(uint32_t)(sgd_d2u(t) & 0xffffffffull); \
  const double kd = t - magic; \
  const double r1 = x - kd * SGD_EXP_L1;                                   /* exact: L1 has 32 bits */ \
  const double r2 = kd * SGD_EXP_L2; \
  const double r = r1 - r2; \
  const double q = r * r * (0.5 + r * (0.16666666666666666 + r * (0.041666666666666664 + \
                   r * (0.008333333333333333 + r * 0.001388888888888889)))); \
  const double p = r1 - (r2 - q); \
  const int j = (int)((uint32_t)N & 63u); \
  const int m = (N - j) / 64; \
  const double hi = tab[2 * j], lo = tab[2 * j + 1]; \
  const double res = hi + (lo + (hi + lo) * p); \
  if (m >= -1021 && m <= 1023) return res * sgd_u2d((uint64_t)(m + 1023) << 52); \
  if (m > 1023) return res * sgd_u2d((uint64_t)(m - 1 + 1023) << 52) * 2.0; \
  return res * sgd_u2d((uint64_t)(m + 1000 + 1023) << 52) * sgd_u2d((uint64_t)(1023 - 1000) << 52);   /* subnormal range */ \
}
SGD_DEFINE_EXP(sgd_exp_from, const double*)

SGD_DM_TABFN double sgd_exp(double x) { return sgd_exp_from(x, SGD_EXP_TABPTR); }

/* `tab`: sgd_log_tab as 258 consecutive doubles, or a copy of it under the kernel's pointer type (see SGD_DEFINE_EXP) */
#define SGD_DEFINE_LOG(NAME, TABPTR_T) \
SGD_DM_FN double NAME(double x, TABPTR_T tab) { \
  if (x != x) return x; \
  if (x < 0.0) return sgd_u2d(0x7ff8000000000000ull);                      /* NaN */ \
  if (x == 0.0) return sgd_u2d(0xfff0000000000000ull);                     /* -inf */ \
  uint64_t u = sgd_d2u(x); \
  if (u == 0x7ff0000000000000ull) return x; \
  int m = 0; \
  if ((u >> 52) == 0) {                                                    /* subnormal: scale up */ \
    x *= 18014398509481984.0;                                              /* 2^54 */ \
    u = sgd_d2u(x); \
    m = -54; \
  } \
  m += (int)(u >> 52) - 1023; \
  const double Y = sgd_u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);   /* [1, 2) */ \
  const int j = (int)((Y - 1.0) * 128.0 + 0.5);                            /* 0..128 */ \
  const double F = 1.0 + (double)j * 0.0078125; \
  const double f = Y - F;                                                  /* exact */ \
  const double uu = (f + f) / (Y + F); \
  const double v = uu * uu; \
  const double q = uu * v * (0.08333333333333333 + v * (0.0125 + v * 0.002232142857142857)); \
  const double md = (double)m; \
  return (md * SGD_LN2_HI + tab[2 * j]) + (uu + (q + (md * SGD_LN2_LO + tab[2 * j + 1]))); \
}
SGD_DEFINE_LOG(sgd_log_from, const double*)

SGD_DM_TABFN double sgd_log(double x) { return sgd_log_from(x, SGD_LOG_TABPTR); }


#endif
