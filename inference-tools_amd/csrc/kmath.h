// exp and log1p for the covariance kernels (kbuild.hip): K = a^2 exp(-s) (SquaredExponential, covariance.py:254)
// and K = a^2 (1 + s / kappa)^-kappa = a^2 exp(-kappa log1p(s / kappa)) (RationalQuadratic, covariance.py:348), evaluated
// for the N^2 elements of every covariance build.
//
// Why not the library's exp / pow: inlined once per element they re-materialise their 64-bit constants with two
// v_mov_b32 each at every use (323 v_mov_b32 for the 16 elements of a thread - as many VALU slots as the arithmetic),
// and pow() is ~300 instructions per element.  Here a thread's elements are evaluated in LOCKSTEP (N at a time, the
// coefficient loop outermost), so a coefficient is fetched once per N FMAs, and the power is one log1p + one exp.
//
// Accuracy (tools/kmath_check.cpp, against long double on the host, 10^7 arguments each): see the numbers that program
// prints - exp_neg and log1p_pos within 1 ulp, the RationalQuadratic power within a few 1e-16 x (1 + kappa log1p(s / kappa))
// relative; the tests hold K to 1e-13 of the reference's NumPy values.
// Plain fma / rint / ldexp / frexp arithmetic: the same source compiles for the host check.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define KMATH_HD __host__ __device__ __forceinline__
#else
#define KMATH_HD inline
#endif

namespace kmath {

// seed of a reciprocal (v_rcp_f64: ~2^-26 relative; the host check takes the exact quotient - the Newton steps behind it
// make the result insensitive to the seed)
KMATH_HD double rcp_seed(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(d);
#else
  return 1.0 / d;
#endif
}
// a / d to within an ulp without the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup: 12 instructions
// and three scalar-mask dependencies per quotient): two Newton steps on the reciprocal, one correction of the quotient.
// d in [1.7, 2.5] here, a of moderate size: no scaling needed.
KMATH_HD double div_newton(double a, double d) {
  double y = rcp_seed(d);
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  const double q = a * y;
  return fma(fma(-d, q, a), y, q);
}

// exp(x) for x <= 0 (any finite x; x < -745.2 gives 0): x = n ln2 + r, |r| <= ln2 / 2, Taylor polynomial of degree 13
// in Horner form on r, result scaled by 2^n.
template <int N>
KMATH_HD void exp_neg(const double (&x)[N], double (&out)[N]) {
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const double C[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
                        1.0 / 40320.0,      1.0 / 5040.0,      1.0 / 720.0,      1.0 / 120.0,     1.0 / 24.0,
                        1.0 / 6.0,          0.5,               1.0,              1.0};
  double n[N], r[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double xc = x[i] < -1000.0 ? -1000.0 : x[i];
    n[i] = rint(xc * LOG2E);
    r[i] = fma(-n[i], LN2_HI, xc);
    r[i] = fma(-n[i], LN2_LO, r[i]);
    p[i] = C[0];
  }
#pragma unroll
  for (int k = 1; k < 14; ++k) {
    const double c = C[k];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fma(p[i], r[i], c);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = ldexp(p[i], (int)n[i]);
}

// log1p(z) for z >= 0:  u = fl(1 + z), c = (1 + z) - u exactly; log1p(z) = log(u) + c / u.
// log(u): u = 2^e m, m in [sqrt(1/2), sqrt(2)); f = m - 1, t = f / (2 + f), log(m) = f - (hfsq - t (hfsq + R)),
// hfsq = f^2 / 2, R = t^2 (L1 + t^2 (L2 + ...)) - the classic fdlibm arrangement.
template <int N>
KMATH_HD void log1p_pos(const double (&z)[N], double (&out)[N]) {
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const double L[7] = {1.479819860511658591e-01, 1.531383769920937332e-01, 1.818357216161805012e-01,
                       2.222219843214978396e-01, 2.857142874366239149e-01, 3.999999999940941908e-01,
                       6.666666666666735130e-01};
  double f[N], t[N], w[N], R[N], corr[N], ke[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double u = 1.0 + z[i];
    // rounding error of 1 + z (Fast2Sum with the larger operand first)
    const double c = z[i] < 1.0 ? z[i] - (u - 1.0) : 1.0 - (u - z[i]);
    int e;
    double m = frexp(u, &e);  // m in [1/2, 1)
    if (m < 0.70710678118654752440) {
      m *= 2.0;
      e -= 1;
    }
    ke[i] = (double)e;
    f[i] = m - 1.0;
    t[i] = div_newton(f[i], 2.0 + f[i]);
    corr[i] = c * rcp_seed(u);  // a term of at most half an ulp of u: the seed's 26 bits are plenty
    w[i] = t[i] * t[i];
    R[i] = L[0];
  }
#pragma unroll
  for (int k = 1; k < 7; ++k) {
    const double c = L[k];
#pragma unroll
    for (int i = 0; i < N; ++i) R[i] = fma(R[i], w[i], c);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double hfsq = 0.5 * f[i] * f[i];
    const double Rw = R[i] * w[i];
    // log(u) + c / u = e ln2_hi + (f - (hfsq - (t (hfsq + R) + (e ln2_lo + c / u))))
    const double lo = fma(t[i], hfsq + Rw, fma(ke[i], LN2_LO, corr[i]));
    out[i] = fma(ke[i], LN2_HI, f[i] - (hfsq - lo));
  }
}

}  // namespace kmath
