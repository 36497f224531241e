// rtd_dd.h -- double-double helpers (host and device) for the ONE place where the path needs more than float64: moving the
// origin of the thermal source polynomials from tau = 0 to the top of their own layer.
//
// The reference gives the isotropic source of layer l as a polynomial in the ABSOLUTE optical depth, s_l(tau) = sum_j a_j tau^j
// (pydisort.py:316-338 -> scaled_s_poly_coeffs; subroutines.py:746-862 builds the particular solution sum_q b_q(K) tau^q from
// it).  Deep in an atmosphere every float64 evaluation of such a polynomial cancels (a_0 and a_1 tau are both ~ tau_top / dtau
// times the source): the particular solution at the two sides of an interface, in the rows of the boundary-condition system and
// in the evaluators is then consistent to eps x tau_top / dtau only (test 8ARTS_A: 3e-5 pointwise at intensities 1e-6 of the
// largest).  The device therefore keeps the coefficients about the top of the layer: s_l(tau) = sum_i b_i (tau - tau_top)^i, and
// so the particular solution (its construction is translation invariant).  b = Taylor shift of a, done ONCE in double-double --
// exact to the last bit of the result for the coefficients as given -- instead of implicitly, in float64, at every evaluation.
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define RTD_HD __host__ __device__ __forceinline__
#else
#define RTD_HD inline
#endif
#include <cmath>

struct rtd_dd {
  double hi, lo;
};
RTD_HD rtd_dd rtd_two_sum(const double a, const double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
RTD_HD rtd_dd rtd_two_prod(const double a, const double b) {
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
RTD_HD rtd_dd rtd_dd_add(const rtd_dd x, const rtd_dd y) {
  rtd_dd s = rtd_two_sum(x.hi, y.hi);
  s.lo += x.lo + y.lo;
  return rtd_two_sum(s.hi, s.lo);
}
RTD_HD rtd_dd rtd_dd_mul_d(const rtd_dd x, const double t) {  // x * t
  rtd_dd p = rtd_two_prod(x.hi, t);
  p.lo = fma(x.lo, t, p.lo);
  return rtd_two_sum(p.hi, p.lo);
}
// coefficients c[0..n) of p(x) = sum_j c_j x^j  ->  coefficients of the same polynomial in (x - t), in place.
// Horner / Ruffini scheme: n (n - 1) / 2 multiply-adds, in double-double for n <= 16 (every practical source polynomial; the
// work array is 16 double-doubles of a thread's stack); longer polynomials -- the reference puts no limit on Nscoeffs -- take
// the same scheme in plain float64, in place.
RTD_HD void rtd_taylor_shift(double* c, const int n, const double t) {
  if (n > 16) {
    for (int k = 0; k < n - 1; ++k)
      for (int j = n - 2; j >= k; --j) c[j] = fma(c[j + 1], t, c[j]);
    return;
  }
  rtd_dd w[16];
  for (int j = 0; j < n; ++j) w[j] = {c[j], 0.0};
  for (int k = 0; k < n - 1; ++k)
    for (int j = n - 2; j >= k; --j) w[j] = rtd_dd_add(w[j], rtd_dd_mul_d(w[j + 1], t));
  for (int j = 0; j < n; ++j) c[j] = w[j].hi + w[j].lo;
}
