// base ** expo for the GPD inverse-CDF sampler, evaluated in fp64 and rounded once to fp32.
//
// The sampler scale / shape * ((1 - u) ** (-shape) - 1) (gpd-lstm-DE.py:12, mixture-vae-DE.py:17)
// subtracts 1 from a power that sits a few fp32 ulps above 1 when shape is at its 1e-6 clamp, so
// the power must be the correctly rounded fp32 value.  A general fp64 pow costs ~250 instructions
// per element, which turns the HBM-bound sampling kernels compute-bound; this one needs ~50:
//   log(x) = k*ln2 + 2*atanh((m-1)/(m+1)),  x = m * 2^k,  m in [sqrt(1/2), sqrt(2))
//   exp(t) = 2^n * P(r),                     t = n*ln2 + r, |r| <= ln2/2
// Both series are cut below 1e-15 relative, far under the fp32 half-ulp (6e-8), so the final
// rounding differs from an exact evaluation only within ~1e-8 ulp of a rounding tie.
// Plain C so that the host test can compile the same text with gcc.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VTS_HD __host__ __device__ __forceinline__
#else
#define VTS_HD static inline
#endif

VTS_HD double vts_log_pos(double x) {   // x finite and > 0
  int k;
  double m = frexp(x, &k);              // m in [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
  const double num = m - 1.0, den = m + 1.0;
  double y = 1.0 / den;
  double s = num * y;
  s = fma(fma(-s, den, num), y, s);     // one residual step: s = num/den to the last bit or two
  const double s2 = s * s;
  double p = 1.0 / 21.0;
  p = fma(p, s2, 1.0 / 19.0);
  p = fma(p, s2, 1.0 / 17.0);
  p = fma(p, s2, 1.0 / 15.0);
  p = fma(p, s2, 1.0 / 13.0);
  p = fma(p, s2, 1.0 / 11.0);
  p = fma(p, s2, 1.0 / 9.0);
  p = fma(p, s2, 1.0 / 7.0);
  p = fma(p, s2, 1.0 / 5.0);
  p = fma(p, s2, 1.0 / 3.0);
  const double lm = fma(2.0 * s * s2, p, 2.0 * s);
  const double kd = (double)k;
  return fma(kd, 6.93147180369123816490e-01, fma(kd, 1.90821492927058770002e-10, lm));
}

VTS_HD double vts_exp_d(double t) {
  if (t > 709.0) return INFINITY;
  if (t < -745.0) return 0.0;
  const double n = rint(t * 1.44269504088896338700e+00);
  double r = fma(-n, 6.93147180369123816490e-01, t);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;                    // 1/13!
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// Correctly rounded fp32 power for the sampler's domain (0 < base < inf); anything else takes
// the general routine so that NaN / inf / zero / negative-base results keep libm's semantics.
VTS_HD float vts_gpd_pow(float base, float expo) {
  if (!(base > 0.0f) || !(base < INFINITY) || !(fabsf(expo) < INFINITY))
    return (float)pow((double)base, (double)expo);
  return (float)vts_exp_d((double)expo * vts_log_pos((double)base));
}
