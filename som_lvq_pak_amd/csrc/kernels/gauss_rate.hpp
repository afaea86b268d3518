// gauss_rate.hpp -- the gaussian neighbourhood's rate alpha * (float) exp((double) (-dd * dd / (2.0 * radius * radius)))
// (gaussian_adapt, som_rout.c:539-542) at a tenth of the instructions of the chain written out with fp64 library calls,
// and the SAME float.
//
// The chain per (unit, sample) is: dd = (float) sqrt((double) lattice_sq); the float product dd * dd; a double
// division by den = 2 radius^2; a double exp; the rounding of that to float; a float product with alpha.  Written that
// way (common.hpp gaussian_alpha) it costs a quarter-rate v_rsq_f64 with its Newton steps, a double division
// (v_div_scale / v_rcp_f64 / v_div_fmas / v_div_fixup and four refinements) and ocml's exp: as much as updating 300
// dims of a row.  Here:
//   * the square root in fp32, correctly rounded (sqrtf): rounding sqrt to 53 bits and then to 24 gives the
//     correctly rounded 24-bit root (53 >= 2 * 24 + 2);
//   * the division as Markstein's three operations with rcp = RN(1 / den), which the caller computes ONCE per sample
//     (q = n * rcp, r = fma(-q, den, n) exactly, y = fma(r, rcp, q)): the correctly rounded quotient, i.e. the
//     reference's y bit for bit;
//   * exp(y) = 2^k e^f with k = rint(y log2 e), f = y - k ln2 (two fmas, ln2 in two parts), e^f as the Taylor
//     polynomial of degree 12 (|f| <= 0.3466: remainder < 2^-52, Horner's roundings < 2^-50), scaled by v_ldexp_f64:
//     a double within 2^-46 of e^y, relative (measured against expl over 5 10^7 arguments: 2^-51.3).  The float
//     nearest to it is the float nearest to e^y UNLESS e^y lies within that error of the midpoint of two floats: the low 29 bits of the double's mantissa say how far the
//     midpoint is, and a value closer than 2^-40 (relative) to one takes the library chain instead -- one lane in
//     65 000;
//   * y <= -104: e^y < 2^-150, the float is +0 (as the reference's cast of libm's tiny double); -104 < y < -87 (the
//     float would be subnormal: the rounding point moves), a NaN or a den that is not a normal number: library chain.
// Host-compilable (tests/helpers/gauss_rate_check.c runs it against glibc over 10^8 arguments).
#pragma once
#if defined(__HIPCC__)
#define GR_FN __host__ __device__ __forceinline__
#else
#define GR_FN static inline
#endif
// (sqrtf: hipcc rounds it correctly by default, -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is the 1-ulp
// v_sqrt_f32 in this ROCm's headers.  The products are plain: the build has -ffp-contract=off, host check included.)
#define GR_SQRTF(x) __builtin_sqrtf(x)
#define GR_FMUL(a, b) ((a) * (b))
// the polynomial's coefficients are made where they are used (two s_mov each): left to itself the compiler keeps all
// twelve doubles in vector registers across the caller's main loop -- 24 VGPRs, a third of a wave's budget at 8 waves
#if defined(__HIP_DEVICE_COMPILE__)
#define GR_COEF(name, value) double name = (value); asm volatile("" : "+s"(name))
#else
#define GR_COEF(name, value) const double name = (value)
#endif

// rcp for gauss_rate_fast: RN(1 / (2 radius^2)), or a NaN where the fast path has no say (den zero, subnormal, huge)
GR_FN void gauss_rate_den(float radius, double *den, double *rcp) {
  double d = 2.0 * (double)radius;
  d = d * (double)radius;                          // exact: 48 significant bits
  *den = d;
  *rcp = (d >= 1e-200 && d <= 1e200) ? 1.0 / d : __builtin_nan("");
}

// 1: *h = (float) exp((double) (-dd * dd) / den) as the reference computes it; 0: not decided here (library chain)
GR_FN int gauss_rate_fast(float lat_sq, double den, double rcp, float *h) {
  const float dd = GR_SQRTF(lat_sq);
  const float neg = -GR_FMUL(dd, dd);
  const double n = (double)neg;
  const double q = n * rcp;
  const double r = __builtin_fma(-q, den, n);
  const double y = __builtin_fma(r, rcp, q);
  if (y <= -104.0) { *h = 0.0f; return 1; }
  if (!(y >= -87.0)) return 0;
  GR_COEF(log2e, 1.4426950408889634);
  GR_COEF(ln2hi, 6.93147180369123816490e-01);
  GR_COEF(ln2lo, 1.90821492927058770002e-10);
  GR_COEF(c12, 1.0 / 479001600.0);
  GR_COEF(c11, 1.0 / 39916800.0);
  GR_COEF(c10, 1.0 / 3628800.0);
  GR_COEF(c9, 1.0 / 362880.0);
  GR_COEF(c8, 1.0 / 40320.0);
  GR_COEF(c7, 1.0 / 5040.0);
  GR_COEF(c6, 1.0 / 720.0);
  GR_COEF(c5, 1.0 / 120.0);
  GR_COEF(c4, 1.0 / 24.0);
  GR_COEF(c3, 1.0 / 6.0);
  const double k = __builtin_rint(y * log2e);
  double f = __builtin_fma(-k, ln2hi, y);
  f = __builtin_fma(-k, ln2lo, f);
  double p = c12;
  p = __builtin_fma(p, f, c11);
  p = __builtin_fma(p, f, c10);
  p = __builtin_fma(p, f, c9);
  p = __builtin_fma(p, f, c8);
  p = __builtin_fma(p, f, c7);
  p = __builtin_fma(p, f, c6);
  p = __builtin_fma(p, f, c5);
  p = __builtin_fma(p, f, c4);
  p = __builtin_fma(p, f, c3);
  p = __builtin_fma(p, f, 0.5);
  p = __builtin_fma(p, f, 1.0);
  p = __builtin_fma(p, f, 1.0);
  const double hd = __builtin_ldexp(p, (int)k);
  unsigned long long bits;
  __builtin_memcpy(&bits, &hd, 8);
  const unsigned int m = (unsigned int)bits & 0x1FFFFFFFu;          // below the float's last place; the midpoint is 2^28
  const unsigned int off = m > 0x10000000u ? m - 0x10000000u : 0x10000000u - m;
  if (off <= 0x1000u) return 0;                                      // 2^12 of 2^52: 2^-40 relative
  *h = (float)hd;
  return 1;
}
