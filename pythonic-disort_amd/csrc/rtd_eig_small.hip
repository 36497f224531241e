// rtd_eig_small.hip -- the eigen stage of the 2 ... 8-stream path (NP = 4) in ONE-LANE-PER-PROBLEM form (round 4).
//
// Replaces _solve_for_gen_and_part_sols (src/PythonicDISORT/_solve_for_gen_and_part_sols.py:5-243) for these stream counts: the
// same mathematics and the same outputs as rtd_eigen_kernel<NP, 2> (rtd_eig.hip, whose header has the algorithm: symmetrised
// problem, two Cholesky factors, one-sided Jacobi on F = L^T R, Y = L^-T Z, A = L Z, beam and thermal particular solutions).
//
// Why another form.  rtd_eigen_kernel spreads a problem over NP lanes (one column per lane, then the pair layout for the
// sweeps): at 32 streams that is what keeps an eigenproblem in registers.  At NP = 8 the per-step overhead that does not
// shrink with NP -- the rotation parameters (~28 instructions, computed by both lanes of a pair), the cross-lane moves, the
// convergence votes -- is most of a step: 3 100 vector instructions per wavefront of 8 problems, VALU issue 86-88 % of the SIMD
// cycles (profiles/archive/r04_small_stream_path.json): issue-bound at 387 instructions per problem.  Here a lane owns a whole
// problem: an 8 x 8 matrix is 64 registers, every loop is compile-time, nothing crosses lanes, symmetric matrices are packed
// (36 entries), a rotation's parameters are computed once per pair, and the table rows Ybar_l^m(mu_i) are wave-uniform scalar
// loads (a wavefront takes ONE Fourier mode and 64 (column, layer) pairs; that also keeps the spread of the sweep counts inside
// a wavefront small: they fall with m).  ~150 instructions per problem.  One wavefront per SIMD (the working set is ~200
// doubles per lane at NP = 8): nothing to hide -- the loads are issued up front and the stores leave at the end.
//
// Measured (cfg3, 1 024 columns, one MI355X): NP = 4 (8 streams): 27 us against 37 us for rtd_eigen_kernel<4, 2> -- this kernel is
// the 2 ... 8-stream path.  NP = 8 (10 ... 16 streams): 196 us against 134 us -- the working set of an 8 x 8 problem (Z 64 + packed L 36
// + the vectors of the particular solutions) needs all 512 registers of a lane and 116 spilled ones, i.e. ONE wavefront per SIMD
// with nothing to hide the latency of its dependent FP64 chains, and 2 048 wavefronts are two rounds of that: the per-problem
// instruction count fell (13 600 per 64 problems against 3 100 per 8) but the kernel left the issue-bound regime.  The NP = 8
// instance is therefore not built into the library (RTD_EIG_LANE8 in the launcher's place would be one line); 10 ... 16 streams
// stay on rtd_eigen_kernel<8, 2>.
//
// rtd_eigen_kernel<4, 2> stays selectable (RTD_EIG_SMALL_V1=1: A/B runs, and the suite passes under it).
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

__device__ __forceinline__ double lane_rsqrt(double x) {  // 1/sqrt(x), two Newton steps from the hardware seed
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
__device__ __forceinline__ double lane_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
__device__ __forceinline__ double lane_rsqrt1(double x) {  // one Newton step (~4e-15): quantities that only scale a rotation
  const double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}

#ifndef RTD_JAC_TOL
#define RTD_JAC_TOL 1e-14  /* as rtd_eig.hip: a sweep is the last one when every pair had cos^2 <= this before its rotation */
#endif

// round-robin ("circle") ordering of the NP (NP - 1) / 2 column pairs: NP - 1 rounds of NP / 2 disjoint pairs (the pairs of
// a round are independent: instruction-level parallelism for the single wavefront of a SIMD)
template <int NP>
struct Pairs {
  int p[NP * (NP - 1) / 2], q[NP * (NP - 1) / 2];
  constexpr Pairs() : p{}, q{} {
    int n = 0;
    for (int r = 0; r < NP - 1; ++r) {
      for (int k = 0; k < NP / 2; ++k) {
        int a = k == 0 ? NP - 1 : (r + k) % (NP - 1);
        int b = k == 0 ? r : (r - k + (NP - 1)) % (NP - 1);
        p[n] = a < b ? a : b;
        q[n] = a < b ? b : a;
        ++n;
      }
    }
  }
};

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

__host__ __device__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }  // packed lower triangle, r >= c

template <int NP>
__global__ __launch_bounds__(64, 1) void rtd_eigen_lane_kernel(RtdDev d) {
  constexpr int TRI = NP * (NP + 1) / 2, NN = NP * NP, Q = 2 * NP, NPAIR = NP * (NP - 1) / 2;
  // a wavefront = one Fourier mode m (wave-uniform: scalar table loads) and 64 (column, layer) pairs
  const int M = d.M, L = d.L, P = d.P;
  const int m = (int)(blockIdx.x % M);
  const long chunk = blockIdx.x / M;
  const long ncl = (long)d.C * d.ln;  // layer shards (rtd_plan_solve_layers) decompose the layers [l0, l0 + ln) only
  long idx = chunk * 64 + threadIdx.x;
  const bool valid = idx < ncl;
  if (!valid) idx = ncl - 1;  // (a lane without a problem redoes the last one and skips the stores)
  const int c = (int)(idx / d.ln), l = d.l0 + (int)(idx % d.ln);
  const int mg = d.m0 + d.mstep * m;
  const long cl = (long)c * L + l;
  const long pid = ((long)c * M + m) * L + l;
  const bool beam = d.beam != 0;

  // ---- inputs: everything this lane needs from memory is requested here
  const double om = d.omega[cl];
  double wlr[2 * NP], y0r[2 * NP];  // the layer's moments and Ybar_l^m(-mu0) for l = mg ... mg + 2 NP - 1 (clamped: P <= 2 NP)
  {
    const double* wl = d.wleg + cl * P;
    const double* Y0b = d.Y0 + ((long)c * M + m) * P;
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t) {
      const int ell = min(mg + t, P - 1);
      wlr[t] = wl[ell];
      y0r[t] = beam ? Y0b[ell] : 0.0;
    }
  }
  double invmu[NP], Sv[NP], Tv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {  // (wave-uniform addresses: scalar loads)
    invmu[i] = d.invmu[i];
    Sv[i] = d.S[i];
    Tv[i] = d.T[i];
  }
  const double* Ym = d.Y + (long)m * P * NP;  // [P][NP], wave-uniform

  // ---- assembly (:123-135): Pm = M^-1 - S (2 sum_{l - m even} c_l Y_l Y_l^T) S, Qm over the odd terms; packed lower triangles.
  //      The beam source sums (:143-152) fall out of the same loop.
  double pm[TRI], qm[TRI], xe[NP], xo[NP];
#pragma unroll
  for (int t = 0; t < TRI; ++t) pm[t] = qm[t] = 0.0;
#pragma unroll
  for (int i = 0; i < NP; ++i) xe[i] = xo[i] = 0.0;
  double cmax = 0.0;
#pragma unroll
  for (int t = 0; t < 2 * NP; ++t) {
    if (mg + t < P) {  // (wave-uniform)
      const double* Yr = Ym + (long)(mg + t) * NP;
      double yr[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) yr[i] = Yr[i];
      const double clv = 0.5 * om * wlr[t];
      cmax = fmax(cmax, fabs(clv));
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const double coef = 2.0 * clv * yr[j];
        if (t % 2 == 0) {
          xe[j] = fma(coef, y0r[t], xe[j]);
#pragma unroll
          for (int i = j; i < NP; ++i) pm[tri(i, j)] = fma(coef, yr[i], pm[tri(i, j)]);
        } else {
          xo[j] = fma(coef, y0r[t], xo[j]);
#pragma unroll
          for (int i = j; i < NP; ++i) qm[tri(i, j)] = fma(coef, yr[i], qm[tri(i, j)]);
        }
      }
    }
  }
  // "shortcut" of the reference when multiple scattering is insignificant (:119, :162-168): the layer is treated as
  // non-scattering; the general path then gives G = [[0, D], [D, 0]], k = 1/mu, B = 0
  const double live = (cmax > 1e-8) ? 1.0 : 0.0;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    xe[j] *= live;
    xo[j] *= live;
#pragma unroll
    for (int i = j; i < NP; ++i) {
      const double dg = (i == j) ? invmu[j] : 0.0;
      pm[tri(i, j)] = dg - Sv[i] * (live * pm[tri(i, j)]) * Sv[j];
      qm[tri(i, j)] = dg - Sv[i] * (live * qm[tri(i, j)]) * Sv[j];
    }
  }

  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- Cholesky factors Pm = L L^T, Qm = R R^T, in place (a non-positive pivot -- phase function not positive definite after
  //      delta-M scaling -- leaves NaN, reported through the eigenvalue check below)
  double dinv[NP];
  auto cholesky = [&](double (&a)[TRI], double* inv_diag) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const double r = lane_rsqrt(a[tri(k, k)]);
      if (inv_diag) inv_diag[k] = r;
      a[tri(k, k)] *= r;
#pragma unroll
      for (int i = k + 1; i < NP; ++i) a[tri(i, k)] *= r;
#pragma unroll
      for (int j = k + 1; j < NP; ++j)
#pragma unroll
        for (int i = j; i < NP; ++i) a[tri(i, j)] = fma(-a[tri(i, k)], a[tri(j, k)], a[tri(i, j)]);
    }
  };
  cholesky(pm, dinv);
  cholesky(qm, nullptr);

  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- F = L^T R, column j in f[j][.]:  F[i][j] = sum_{r >= max(i, j)} L[r][i] R[r][j]
  double f[NP][NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double a = 0.0;
#pragma unroll
      for (int r = (i > j ? i : j); r < NP; ++r) a = fma(pm[tri(r, i)], qm[tri(r, j)], a);
      f[j][i] = a;
    }

  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- one-sided (Hestenes) Jacobi on the columns of F: H = F F^T = L^T Qm L has the eigenvalues k^2 and the rotated columns
  //      converge to k_j z_j.  Cyclic sweeps in the round-robin order; a sweep is the last one when every pair it met had
  //      cos^2 <= RTD_JAC_TOL before its rotation; the wavefront sweeps until its slowest problem is done (one mode per
  //      wavefront keeps them close).  The angle comes from float arithmetic (it only steers the iteration), c and s are double.
  //      A lane that has had its last sweep is FROZEN (angle 0: the rotation is the identity, bit for bit) while the wavefront
  //      goes on for its slower problems: a problem's result must not depend on which problems share its wavefront -- a windowed
  //      plan groups them differently and has to return the same bits.
  int nsweep = 0;
  bool lane_done = false;
  {
    constexpr Pairs<NP> pr{};
    bool done = false;
    for (int sweep = 0; sweep < 40; ++sweep) {
      double nrm[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < NP; ++i) a = fma(f[j][i], f[j][i], a);
        nrm[j] = a;
      }
      int notconv = 0;
      static_for<0, NPAIR>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int pp = pr.p[t], qq = pr.q[t];
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i < NP; ++i) gamma = fma(f[pp][i], f[qq][i], gamma);
        const double ax = nrm[pp], ay = nrm[qq];
        // tan(2 theta) = 2 gamma / (|y|^2 - |x|^2);  t = tan(theta) without cancellation; the tiny term keeps gamma = 0 at t = 0
        const float df = (float)(ay - ax), gf = (float)(2.0 * gamma);
        const float r2f = fmaf(df, df, fmaf(gf, gf, 1e-36f));
        const float rhof = r2f * __builtin_amdgcn_rsqf(r2f);
        const float denf = df + copysignf(rhof, df);
        const double tt = done ? 0.0 : (double)(gf * __builtin_amdgcn_rcpf(denf));
        const double cs = lane_rsqrt1(fma(tt, tt, 1.0));
        const double sn = tt * cs;
        notconv |= (gamma * gamma > RTD_JAC_TOL * ax * ay) ? 1 : 0;
        nrm[pp] = fma(-tt, gamma, ax);  // |c x - s y|^2
        nrm[qq] = fma(tt, gamma, ay);   // |s x + c y|^2
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const double xi = f[pp][i], yi = f[qq][i];
          f[pp][i] = fma(-sn, yi, cs * xi);
          f[qq][i] = fma(cs, yi, sn * xi);
        }
      });
      ++nsweep;
      done = done || !notconv;
      if (!__any(!done)) break;
    }
    lane_done = done;
  }
  if (threadIdx.x == 0 && nsweep > *(volatile int*)d.sweeps) atomicMax(d.sweeps, nsweep);
  if (!lane_done && valid) rtd_raise(d, RTD_ST_JACOBI, mg, c);  // this lane's problem hit the sweep limit (its own: per lane)
  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- eigenvalues and Z (in place over F)
  double k2[NP], kj[NP], rk[NP];
  bool okk = true;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) a = fma(f[j][i], f[j][i], a);
    k2[j] = a;
    rk[j] = lane_rsqrt(a);
    kj[j] = a * rk[j];
    okk = okk && (a > 0.0 && a < 1e300);
#pragma unroll
    for (int i = 0; i < NP; ++i) f[j][i] *= rk[j];
  }
  // a non-positive Cholesky pivot or an overflow shows up as a non-finite or non-positive eigenvalue: the reference's eig / sqrt
  // would return NaN here (:186)
  if (valid && !okk) rtd_raise(d, RTD_ST_CHOL, mg, c);

  // From here on only Z (in f) and L (in pm, with 1 / L[i][i] in dinv) are kept: Y = L^-T Z and A = L Z are what the later stages
  // read, but holding them beside Z (192 doubles) is more than a lane has; every product with Y is a product with Z followed by
  // a triangular solve with L^T (Y v = L^-T (Z v), Y^T v = Z^T (L^-1 v)), and Y, A themselves are formed row by row at the end
  // (A from Z, then Y in place over Z) and leave as whole rows.
  auto solve_L = [&](double (&v)[NP]) {  // v <- L^-1 v
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double a = v[i];
#pragma unroll
      for (int r = 0; r < i; ++r) a = fma(-pm[tri(i, r)], v[r], a);
      v[i] = a * dinv[i];
    }
  };
  auto solve_Lt = [&](double (&v)[NP]) {  // v <- L^-T v
#pragma unroll
    for (int i = NP - 1; i >= 0; --i) {
      double a = v[i];
#pragma unroll
      for (int r = i + 1; r < NP; ++r) a = fma(-pm[tri(r, i)], v[r], a);
      v[i] = a * dinv[i];
    }
  };
  auto mul_L = [&](const double (&v)[NP], double (&o)[NP]) {  // o = L v
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double a = 0.0;
#pragma unroll
      for (int r = 0; r <= i; ++r) a = fma(pm[tri(i, r)], v[r], a);
      o[i] = a;
    }
  };
  auto mul_Z = [&](const double (&v)[NP], double (&o)[NP]) {  // o = Z v  (sum over the eigen-index)
#pragma unroll
    for (int i = 0; i < NP; ++i) o[i] = 0.0;
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int i = 0; i < NP; ++i) o[i] = fma(f[e][i], v[e], o[i]);
  };
  auto mul_Zt = [&](const double (&v)[NP], double (&o)[NP]) {  // o = Z^T v
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) a = fma(f[e][i], v[i], a);
      o[e] = a;
    }
  };
  if (valid) {
    const double dtau = d.taus0[(long)c * (L + 1) + l + 1] - d.taus0[(long)c * (L + 1) + l];
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      d.kk[pid * NP + e] = kj[e];
      d.Ek[pid * NP + e] = exp(-kj[e] * dtau);
    }
  }

  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- beam particular solution (:143-152, :226-231) through the spectral decomposition (rtd_eig.hip):
  //  s = B+ + B-, dd = B+ - B- ;  (I/mu0^2 - Qm Pm) T s = T(x+ + x-)/mu0 - Qm T (x+ - x-),  T dd = mu0 [T (x+ - x-) - Pm T s],
  //  Qm = Y k^2 Y^T,  Qm Pm = L^-T Z k^2 Z^T L^T
  if (beam) {
    const double mu0 = d.mu0[c], rmu0 = lane_rcp(mu0);
    const double fac = d.I0[c] * (0.25 / M_PI) * (mg == 0 ? 1.0 : 2.0);
    double v0[NP], rhat[NP], t1[NP], t2[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v0[i] = 2.0 * Tv[i] * (fac * xe[i]) * invmu[i];  // T (x+ - x-)
    // Qm v0 = Y k^2 Y^T v0 = L^-T Z k^2 Z^T L^-1 v0
#pragma unroll
    for (int i = 0; i < NP; ++i) t1[i] = v0[i];
    solve_L(t1);
    mul_Zt(t1, t2);
#pragma unroll
    for (int e = 0; e < NP; ++e) t2[e] *= k2[e];
    mul_Z(t2, t1);
    solve_Lt(t1);
#pragma unroll
    for (int i = 0; i < NP; ++i) rhat[i] = 2.0 * Tv[i] * (fac * xo[i]) * invmu[i] * rmu0 - t1[i];
#pragma unroll
    for (int j = 0; j < NP; ++j) {  // g = L^T rhat
      double a = 0.0;
#pragma unroll
      for (int r = j; r < NP; ++r) a = fma(pm[tri(r, j)], rhat[r], a);
      t1[j] = a;
    }
    mul_Zt(t1, t2);  // h = Z^T g / (1/mu0^2 - k^2)
#pragma unroll
    for (int e = 0; e < NP; ++e) t2[e] *= lane_rcp(rmu0 * rmu0 - k2[e]);
    double ev[NP], sh[NP], ps[NP];
    mul_Z(t2, ev);  // t = L^T shat = Z h
#pragma unroll
    for (int i = 0; i < NP; ++i) sh[i] = ev[i];
    solve_Lt(sh);   // shat = Y h = L^-T Z h
    mul_L(ev, ps);  // Pm shat = L (L^T shat) = L t
    bool okb = true;
    double bp[NP], bm[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const double rT = lane_rcp(Tv[i]);
      const double s_i = sh[i] * rT, d_i = mu0 * (v0[i] - ps[i]) * rT;
      bp[i] = 0.5 * (s_i + d_i);
      bm[i] = 0.5 * (s_i - d_i);
      okb = okb && (fabs(s_i) + fabs(d_i) < 1e300);
    }
    if (valid) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        d.Bv[pid * Q + i] = bp[i];
        d.Bv[pid * Q + NP + i] = bm[i];
      }
      // 1/mu0 on an eigenvalue: the reference's solve (:226-231) meets a singular matrix
      if (!okb) rtd_raise(d, RTD_ST_BEAM, mg, c);
    }
  }

  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- isotropic (thermal) source, Fourier mode 0 only (subroutines.py:746-862, _assemble.py:124); wave-uniform branch
  if (d.Ns > 0 && mg == 0) {
    // zneg_e = -k_e/2 [Z^T L^-1 (T/mu)]_e
    double zn[NP];
    {
      double tm[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) tm[i] = Tv[i] * invmu[i];
      solve_L(tm);
      mul_Zt(tm, zn);
#pragma unroll
      for (int e = 0; e < NP; ++e) zn[e] *= -0.5 * kj[e];
    }
    if (valid)
#pragma unroll
      for (int e = 0; e < NP; ++e) d.zneg[cl * NP + e] = zn[e];
    const double* sp = d.spoly + cl * d.Ns;
    // (the coefficients sp are about the layer's top, rtd_dd.h: the top is x = 0, the bottom x = the layer's scaled thickness)
    const double ts_top = 0.0, ts_bot = d.taus0[(long)c * (L + 1) + l + 1] - d.taus0[(long)c * (L + 1) + l];
    double vtu[NP], vtd[NP], vbu[NP], vbd[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) vtu[i] = vtd[i] = vbu[i] = vbd[i] = 0.0;
    double tpt = 1.0, tpb = 1.0;
    for (int q = 0; q < d.Ns; ++q) {
      double sab[NP], dab[NP];
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        // b_q(K) = sum_{jj >= q} jj!/q! a_jj K^-(jj - q + 1), K = -k (first N eigen-columns) and +k
        double bneg = 0.0, bpos = 0.0, ratio = 1.0, pw_pos = rk[e], pw_neg = -rk[e];
        for (int jj = q; jj < d.Ns; ++jj) {
          bpos += ratio * sp[jj] * pw_pos;
          bneg += ratio * sp[jj] * pw_neg;
          ratio *= (double)(jj + 1);
          pw_pos *= rk[e];
          pw_neg *= -rk[e];
        }
        const double a = zn[e] * bneg, b = -zn[e] * bpos;
        sab[e] = a + b;
        dab[e] = (a - b) * rk[e];
      }
      // up = [Y (a + b) - A (a - b)/k]/T, down = [Y (a + b) + A (a - b)/k]/T with Y v = L^-T (Z v), A v = L (Z v)
      double py[NP], pa[NP], tz[NP];
      mul_Z(sab, py);
      solve_Lt(py);
      mul_Z(dab, tz);
      mul_L(tz, pa);
      double up[NP], dn[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const double rT = 1.0 / Tv[i];
        up[i] = (py[i] - pa[i]) * rT;
        dn[i] = (py[i] + pa[i]) * rT;
        vtu[i] += up[i] * tpt;
        vtd[i] += dn[i] * tpt;
        vbu[i] += up[i] * tpb;
        vbd[i] += dn[i] * tpb;
      }
      if (valid) {
        double* dq = d.dq + (cl * d.Ns + q) * Q;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          dq[i] = up[i];
          dq[NP + i] = dn[i];
        }
      }
      tpt *= ts_top;
      tpb *= ts_bot;
    }
    if (valid) {  // v_l at the layer's own boundaries (what rtd_bc_small_kernel reads instead of the polynomials)
      double* vb = d.vb + cl * 4 * NP;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        vb[i] = vtu[i];
        vb[NP + i] = vtd[i];
        vb[2 * NP + i] = vbu[i];
        vb[3 * NP + i] = vbd[i];
      }
    }
  }

  __builtin_amdgcn_sched_barrier(0);  // (stage boundary: the scheduler must not interleave independent stages -- registers)
  // ---- eigenvector blocks (:190-198): stored are A = L Z and Y = L^-T Z, [stream][eigen-index], whole rows at a time
  {
    double* Ao = d.Am + pid * NN;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double row[NP];
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r <= i; ++r) a = fma(pm[tri(i, r)], f[e][r], a);
        row[e] = a;
      }
      if (valid)
#pragma unroll
        for (int e = 0; e < NP; ++e) Ao[i * NP + e] = row[e];
    }
    double* Yo = d.Ym + pid * NN;
#pragma unroll
    for (int i = NP - 1; i >= 0; --i) {  // in place over Z, from the last row up
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        double a = f[e][i];
#pragma unroll
        for (int r = i + 1; r < NP; ++r) a = fma(-pm[tri(r, i)], f[e][r], a);
        f[e][i] = a * dinv[i];
      }
      if (valid)
#pragma unroll
        for (int e = 0; e < NP; ++e) Yo[i * NP + e] = f[e][i];
    }
  }
}

}  // namespace

void rtd_launch_eig_small(const RtdDev& d, hipStream_t s) {
  const long ncl = (long)d.C * d.ln;
  const dim3 grid((unsigned)(((ncl + 63) / 64) * d.M));
  hipLaunchKernelGGL(rtd_eigen_lane_kernel<4>, grid, dim3(64), 0, s, d);  // (NP = 4 only: see the header)
}
