// rtd_eig.hip -- per (column, Fourier mode, layer) eigen stage on gfx950.
//
// Replaces _solve_for_gen_and_part_sols (src/PythonicDISORT/_solve_for_gen_and_part_sols.py:5-243):
//   Legendre tables (:96-109)  -> rtd_tables_* kernels (normalised three-term recurrences)
//   D+/D-, alpha, beta (:123-135), eig((alpha-beta)(alpha+beta)) (:179-183), G blocks (:186-198),
//   beam particular solution (:143-152, :209-231), G^-1 [1/mu;-1/mu] (:203-205 + _assemble.py:124)
//   and the isotropic-source particular solution coefficients (subroutines.py:746-862)
//   -> rtd_eigen_kernel<NP> (one fused kernel; rtd_asm / rtd_jacobi / rtd_post_kernel are its earlier three-kernel form,
//      kept for NQuad > 32 behind RTD_EIG32_SPLIT).
//
// Algorithm (own design, not the reference's LAPACK calls): with T = diag(sqrt(mu w)) the matrices
// -(T(alpha+beta)T^-1) = Pm and -(T(alpha-beta)T^-1) = Qm are symmetric positive definite, so with
// the Cholesky factor Pm = L L^T the non-symmetric problem (alpha-beta)(alpha+beta) v = k^2 v becomes
// the symmetric H z = k^2 z, H = L^T Qm L = F F^T with F = L^T R (Qm = R R^T), solved by a one-sided (Hestenes)
// parallel-order (XOR round-robin) cyclic Jacobi iteration on the columns of F.  One problem occupies NP lanes of a
// wavefront (64/NP problems per wave); lane j owns column j of every matrix in registers; columns are exchanged with
// DPP / ds_swizzle cross-lane moves, small vectors and the Cholesky factor go through LDS.
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

// compiler-only barrier: keeps the scheduler from hoisting a whole unrolled loop's LDS loads
#define RTD_FENCE() asm volatile("" ::: "memory")
#ifndef RTD_XOR_DPP
#define RTD_XOR_DPP 0x818E  /* bit set of the xor masks done by ONE DPP move per dword (1, 2, 3, 7, 8, 15) instead of ds_swizzle:
                               A/B on one box 6.03 -> 5.94 ms (with bound_ctrl moves; with the old copy-then-move form it lost 2 %) */
#endif

// DPP control of a lane permutation "lane ^ MASK" inside a 16-lane row that one DPP move can express, else -1:
// quad permutes for 1, 2, 3; row_half_mirror = ^7; row_ror:8 = ^8; row_mirror = ^15
constexpr __host__ __device__ int dpp_xor_ctrl(int mask) {
  return mask == 1 ? 0xB1 : mask == 2 ? 0x4E : mask == 3 ? 0x1B : mask == 7 ? 0x141 : mask == 8 ? 0x128 : mask == 15 ? 0x140 : -1;
}

template <int MASK>
__device__ __forceinline__ double xor_lane(double v) {
  // value of lane (lane ^ MASK); MASK < 32
  int lo = __double2loint(v), hi = __double2hiint(v);
  constexpr int ctrl = dpp_xor_ctrl(MASK);
  if constexpr (ctrl >= 0 && ((RTD_XOR_DPP >> MASK) & 1)) {  // one VALU move per dword, no LDS crossbar (RTD_XOR_DPP = bit set of masks)
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
  } else {  // ds_swizzle bit-mode: and = 0x1f, or = 0, xor = MASK
    constexpr int pat = (MASK << 10) | 0x1F;
    lo = __builtin_amdgcn_ds_swizzle(lo, pat);
    hi = __builtin_amdgcn_ds_swizzle(hi, pat);
  }
  return __hiloint2double(hi, lo);
}

constexpr __host__ __device__ int high_bit(int t) {
  int b = 1;
  while ((b << 1) <= t) b <<= 1;
  return b;
}

// 1/sqrt(x) and 1/x to full double precision from the hardware seeds (x > 0, normal range)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
// one Newton step from the ~5e-8 hardware seeds: ~4e-15 relative (tools/hiptests/seed_accuracy.hip), enough for
// quantities that only steer a rotation angle
__device__ __forceinline__ double approx_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}
__device__ __forceinline__ double approx_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return y * (2.0 - x * y);
}

#ifndef RTD_EIGEN_WAVES
#define RTD_EIGEN_WAVES 3  /* waves per SIMD the fused eigen kernel is compiled for (LDS: 10 KB per wave) */
#endif

// a sweep is the last one when every pair it met had cos^2(angle) <= RTD_JAC_TOL before its rotation (quadratic
// convergence squares the residual angle during that sweep).  Measured on the benchmark columns against the CPU
// oracle: 1e-16 .. 1e-11 give the same max |dI| (1.6e-11 abs, 4.1e-10 rel: other roundoff dominates), 1e-9 gives
// 1.4e-8 rel, 1e-7 gives 5e-7; 1e-11 saves a third of a sweep on average.
#ifndef RTD_JAC_TOL
#define RTD_JAC_TOL 1e-11
#endif

// One parallel step of the one-sided (Hestenes) Jacobi iteration on the columns of W (H = W W^T at the
// start): lane j holds column j, all pairs (j, j^T) are orthogonalised at once.  On convergence the columns
// are k_j z_j (singular values x left singular vectors = sqrt(eigenvalues) x eigenvectors of H).
template <int NP, int T>
struct JacobiStep {
  // alpha = |w|^2 of this lane's column, maintained across steps (it only steers the rotation angles, so the
  // slow drift of the recurrence is harmless; it is recomputed from the column at every sweep start)
  static __device__ __forceinline__ void run(double (&w)[NP], double& alpha, const int j, int& notconv) {
    double pw[NP];
    double gamma = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      pw[i] = xor_lane<T>(w[i]);
      gamma += w[i] * pw[i];
    }
    const double beta = xor_lane<T>(alpha);
    const bool lo = (j & high_bit(T)) == 0;  // j < j^T
    // tan(2 theta) = 2 gamma / (a_hi - a_lo);  t = tan(theta) without cancellation.  Branch-free: the tiny term keeps
    // gamma = 0 (decoupled or already orthogonal columns, also with equal norms) at t = 0, c = 1 without a 0/0.
    const double delta = lo ? (beta - alpha) : (alpha - beta);
    const double g2 = 2.0 * gamma;
    const double r2 = delta * delta + (g2 * g2 + 1e-280);
    const double rho = r2 * approx_rsqrt(r2);
    const double den = delta + copysign(rho, delta);
    const double tt = g2 * approx_rcp(den);
    // c from ONE Newton step (4e-15): the error scales BOTH columns of the pair by the same 1 + eps, so orthogonality and
    // the directions z are untouched; only the norms k drift, by ~50 rotations x 4e-15 (parity unchanged)
    const double c = approx_rsqrt(1.0 + tt * tt);
    const double tsg = lo ? -tt : tt;
    const double sg = tsg * c;
    notconv |= (gamma * gamma > RTD_JAC_TOL * alpha * beta) ? 1 : 0;
    alpha = fma(tsg, gamma, alpha);  // |c w + sg pw|^2 = alpha -+ t gamma for the rotation that annihilates gamma
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = c * w[i] + sg * pw[i];
    JacobiStep<NP, T + 1>::run(w, alpha, j, notconv);
  }
};
template <int NP>
struct JacobiStep<NP, NP> {
  static __device__ __forceinline__ void run(double (&)[NP], double&, const int, int&) {}
};

// Transposed reduction: every lane enters with NP terms v[0..NP) (term i belongs to row i) and leaves with the sum of
// row `j` over the NP lanes of its group -- NP-1 swizzle-adds in registers, no LDS memory.
template <int NP, int O>
struct TransposeStep {
  static __device__ __forceinline__ void run(double (&v)[NP], const int j) {
    const bool hi = (j & O) != 0;
#pragma unroll
    for (int i = 0; i < O; ++i) {
      const double keep = hi ? v[i + O] : v[i];
      const double send = hi ? v[i] : v[i + O];
      v[i] = keep + xor_lane<O>(send);
    }
    TransposeStep<NP, O / 2>::run(v, j);
  }
};
template <int NP>
struct TransposeStep<NP, 0> {
  static __device__ __forceinline__ void run(double (&)[NP], const int) {}
};
template <int NP>
__device__ __forceinline__ double transpose_reduce(double (&v)[NP], const int j) {
  TransposeStep<NP, NP / 2>::run(v, j);
  return v[0];
}

// sum over the NP lanes of a group (result in every lane)
template <int NP>
__device__ __forceinline__ double group_sum(double v) {
  if (NP > 1) v += xor_lane<1>(v);
  if (NP > 2) v += xor_lane<2>(v);
  if (NP > 4) v += xor_lane<4>(v);
  if (NP > 8) v += xor_lane<8>(v);
  if (NP > 16) v += xor_lane<16>(v);
  return v;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// value of lane K of this lane's NP-group, K a compile-time constant: a DPP row broadcast (VALU, no LDS crossbar)
// when the group is one 16-lane DPP row, a ds_bpermute otherwise
template <int NP, int K>
__device__ __forceinline__ double bcast_lane(double v) {
  if constexpr (NP == 16) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0xF, true);  // row_newbcast:K; bound_ctrl + full masks: no `old` operand, no copy
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else {
    return __shfl(v, K, NP);
  }
}

// In-register Cholesky of a symmetric positive definite matrix held one column per lane (col[i] = A[i][j]);
// on exit col[i] = L[i][j] for i >= j and 0 above the diagonal; returns 1 / L[j][j].  The trailing matrix is kept
// symmetric so that A^(K)[j][K] is available in the lane's own registers, and the scaling of a finished column by
// 1/sqrt(pivot) is deferred to the end: a step is one broadcast and one FMA per element,
//   col_j[i] -= A^(K)[i][K] * A^(K)[j][K] / A^(K)[K][K]   for j > K (factor 0 for the finished columns j <= K).
// the same broadcast through the LDS crossbar (ds_swizzle bit mode: lane' = (lane & ~(NP-1) & 0x1f) | K inside each half
// wavefront): same move count as DPP, other pipe.  RTD_CHOL_SWZ selects it for the trailing updates of the Cholesky steps.
#ifndef RTD_CHOL_SWZ
#define RTD_CHOL_SWZ 0  /* A/B: 2.5 % slower than DPP */
#endif
template <int NP, int K>
__device__ __forceinline__ double bcast_lane_lds(double v) {
  constexpr int pat = (K << 5) | (0x1F & ~(NP - 1));
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
  return __hiloint2double(hi, lo);
}

template <int NP, int K>
struct CholStep {
  static __device__ __forceinline__ void run(double (&col)[NP], double& diag, const int j) {
    const double akk = bcast_lane<NP, K>(col[K]);
    const double f = (j > K) ? col[K] * fast_rcp(akk) : 0.0;
    diag = (j == K) ? akk : diag;
#pragma unroll
    for (int i = K + 1; i < NP; ++i) {
      if constexpr (RTD_CHOL_SWZ && NP <= 16)
        col[i] = fma(-bcast_lane_lds<NP, K>(col[i]), f, col[i]);
      else
        col[i] = fma(-bcast_lane<NP, K>(col[i]), f, col[i]);
    }
    CholStep<NP, K + 1>::run(col, diag, j);
  }
};
template <int NP>
struct CholStep<NP, NP> {
  static __device__ __forceinline__ void run(double (&)[NP], double&, const int) {}
};
template <int NP>
__device__ __forceinline__ double cholesky_columns(double (&col)[NP], const int j) {
  double diag = 1.0;
  CholStep<NP, 0>::run(col, diag, j);
  const double rinv = fast_rsqrt(diag);
#pragma unroll
  for (int i = 0; i < NP; ++i) col[i] = (i >= j) ? col[i] * rinv : 0.0;
  return rinv;
}

// problem index of this lane's group; invalid groups redo the last layer and skip their stores
struct ProbId {
  long pid;
  int c, m, l;
  int mg;  // the Fourier mode the local index m stands for (mode shards: m0 + mstep * m)
  bool valid;
};
template <int NP>
__device__ __forceinline__ ProbId locate(const RtdDev& d, const int tx = threadIdx.x) {
  // One wavefront = the 64/NP consecutive layers of ONE (column, mode): c and m depend on blockIdx only, so
  // they are wave-uniform and the Legendre-table reads (indexed by m and l only) become scalar loads.
  constexpr int GPW = 64 / NP;
  const int nchunk = (d.L + GPW - 1) / GPW;
  const long cmi = (long)blockIdx.x / nchunk;
  const int chunk = (int)((long)blockIdx.x % nchunk);
  ProbId p;
  p.m = (int)(cmi % d.M);
  p.mg = d.m0 + d.mstep * p.m;
  p.c = (int)(cmi / d.M);
  const int slot = chunk * GPW + tx / NP;
  p.valid = slot < d.L;
  p.l = d.lperm[(long)p.c * d.L + (p.valid ? slot : d.L - 1)];  // invalid groups redo the last slot and skip the stores
  p.pid = cmi * d.L + p.l;
  return p;
}

// ------------------------------------------------------------------------------------------------
// Stage 1: assemble Pm, Qm (symmetrised alpha+-beta), Cholesky Pm = L L^T, Qm = R R^T, F = L^T R (H = F F^T).
// Workspace written: Lw [prob][NP][NP] (row-major L), Qw [prob][NP][NP] (Qm),
//                    F (aliases Ym) [prob][i][j].
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64, (NP <= 16 ? 3 : 1)) void rtd_asm_kernel(RtdDev d) {
  constexpr int GPW = 64 / NP;
  constexpr int LD = NP + 1;
  __shared__ double sL[GPW][NP * LD];
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const ProbId id = locate<NP>(d);
  const int P = d.P, m = id.m;
  double* L_ = sL[grp];
  const double* wl = d.wleg + ((long)id.c * d.L + id.l) * P;
  const double om = d.omega[(long)id.c * d.L + id.l];
  const double* Ym = d.Y + (long)m * P * NP;

  // D+/D- split by parity of (l - m): Ae = 2 sum_even c_l Y_l Y_l^T, Ao likewise (:123-125)
  double acc_e[NP], acc_o[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) acc_e[i] = acc_o[i] = 0.0;
  double cmax = 0.0;
  for (int ell = id.mg; ell < P; ell += 2) {
    {
      const double cl = 0.5 * om * wl[ell];
      cmax = fmax(cmax, fabs(cl));
      const double* Yr = Ym + (long)ell * NP;
      const double coef = 2.0 * cl * Yr[j];
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_e[i] += coef * Yr[i];
    }
    if (ell + 1 < P) {
      const double cl = 0.5 * om * wl[ell + 1];
      cmax = fmax(cmax, fabs(cl));
      const double* Yr = Ym + (long)(ell + 1) * NP;
      const double coef = 2.0 * cl * Yr[j];
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_o[i] += coef * Yr[i];
    }
  }
  // "shortcut" of the reference when multiple scattering is insignificant (:119, :162-168): the layer
  // is treated as non-scattering; the general path then gives G = [[0,D],[D,0]], k = 1/mu, B = 0.
  if (!(cmax > 1e-8)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) acc_e[i] = acc_o[i] = 0.0;
  }
  const double invmu_j = d.invmu[j], S_j = d.S[j];
  double pcol[NP], qcol[NP];
  double* Qw = d.Qw + id.pid * NP * NP;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double Si = d.S[i];
    pcol[i] = (i == j ? invmu_j : 0.0) - Si * acc_e[i] * S_j;  // Pm = M^-1 - S Ae S
    qcol[i] = (i == j ? invmu_j : 0.0) - Si * acc_o[i] * S_j;  // Qm = M^-1 - S Ao S
    if (id.valid) Qw[i * NP + j] = qcol[i];
  }
  cholesky_columns<NP>(pcol, j);  // Pm = L L^T
  cholesky_columns<NP>(qcol, j);  // Qm = R R^T
  double* Lw = d.Lw + id.pid * NP * NP;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    L_[i * LD + j] = pcol[i];
    if (id.valid) Lw[i * NP + j] = pcol[i];
  }
  __syncthreads();
  // H = L^T Qm L = F F^T with F = L^T R;  lane j: F[i][j] = sum_r L[r][i] R[r][j]
  double* Fw = d.Ym + id.pid * NP * NP;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double a = 0.0;
#pragma unroll
    for (int r = i; r < NP; ++r) a += L_[r * LD + i] * qcol[r];
    if (id.valid) Fw[i * NP + j] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 2: one-sided cyclic Jacobi (XOR round-robin ordering) on the columns of F; k^2 -> kk (fixed up in
// stage 3), k_j z_j -> Zw (aliases Am) [prob][i][j].  Registers and cross-lane swizzles only.
// ------------------------------------------------------------------------------------------------
#ifndef RTD_JAC_WAVES
#define RTD_JAC_WAVES 2
#endif
template <int NP>
__global__ __launch_bounds__(64, (NP <= 8 ? 4 : (NP == 16 ? RTD_JAC_WAVES : 1))) void rtd_jacobi_kernel(RtdDev d) {
  const int j = threadIdx.x % NP;
  const ProbId id = locate<NP>(d);
  const double* Fw = d.Ym + id.pid * NP * NP;
  double w[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) w[i] = Fw[i * NP + j];
  int nsweep = 0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int notconv = 0;
    double alpha = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) alpha += w[i] * w[i];
    JacobiStep<NP, 1>::run(w, alpha, j, notconv);
    ++nsweep;
    // every pair met during this sweep was already orthogonal to ~1e-10: the sweep just done finished the job
    if (!__any(notconv)) break;
  }
  if (threadIdx.x == 0 && nsweep > *(volatile int*)d.sweeps) atomicMax(d.sweeps, nsweep);
  if (id.valid) {
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) n2 += w[i] * w[i];
    d.kk[id.pid * NP + j] = n2;  // k^2 (the post kernel takes the root and normalises the column)
    double* Zw = d.Am + id.pid * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) Zw[i * NP + j] = w[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 3: eigenvector blocks Gp/Gm, k, isotropic-source coefficients, beam particular solution.
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64, (NP <= 16 ? 3 : 1)) void rtd_post_kernel(RtdDev d) {
  constexpr int GPW = 64 / NP;
  constexpr int LD = NP + 1;
  __shared__ double sL[GPW][NP * LD];
  __shared__ double sQ[GPW][NP * LD];
  __shared__ double sR[GPW][NP * LD];  // scratch for cross-lane reductions (Qm must survive until the beam stage)
  __shared__ double sV[GPW][3][NP];
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const ProbId id = locate<NP>(d);
  const int P = d.P, m = id.m, c = id.c, l = id.l;
  const bool valid = id.valid;
  double* L_ = sL[grp];
  double* Q_ = sQ[grp];
  double* R_ = sR[grp];
  double* v0 = sV[grp][0];
  double* v1 = sV[grp][1];
  double* v2 = sV[grp][2];
  const long base = id.pid;
  double zc[NP];
  {
    const double* Lw = d.Lw + base * NP * NP;
    const double* Qw = d.Qw + base * NP * NP;
    const double* Zw = d.Am + base * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      L_[i * LD + j] = Lw[i * NP + j];
      Q_[i * LD + j] = Qw[i * NP + j];
      zc[i] = Zw[i * NP + j];
    }
  }
  const double k2 = d.kk[base * NP + j];
  const double kj = sqrt(k2);
  {
    const double rk0 = 1.0 / kj;  // columns arrive as k_j z_j
#pragma unroll
    for (int i = 0; i < NP; ++i) zc[i] *= rk0;
  }
  const double invmu_j = d.invmu[j], T_j = d.T[j];
  __syncthreads();

  // eigenvector blocks (:190-198): V = T^-1 L^-T Z, U = (alpha+beta) V / k = -T^-1 L Z / k; stored as
  // Y = L^-T Z and A = L Z, from which Gp = (Y - A/k)/T, Gm = (Y + A/k)/T, V^-1 = A^T T, U^-1 = -k Y^T T
  double ya[NP], aa[NP];
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double a = zc[i];
#pragma unroll
    for (int r = i + 1; r < NP; ++r) a -= L_[r * LD + i] * ya[r];
    ya[i] = a / L_[i * LD + i];
    RTD_FENCE();
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r <= i; ++r) a += L_[i * LD + r] * zc[r];
    aa[i] = a;
    RTD_FENCE();
  }
  if (valid) {
    double* Ym = d.Ym + base * NP * NP;
    double* Am = d.Am + base * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      Ym[i * NP + j] = ya[i];
      Am[i * NP + j] = aa[i];
    }
    d.kk[base * NP + j] = kj;
    const double* ts0 = d.taus0 + (long)c * (d.L + 1);
    d.Ek[base * NP + j] = exp(-kj * (ts0[l + 1] - ts0[l]));
  }

  // isotropic (thermal) source, Fourier mode 0 only (subroutines.py:746-862, _assemble.py:124).
  // Every group runs the barriers below; only groups with m == 0 store.
  if (d.Ns > 0) {
    const bool act = (id.mg == 0);
    // q = L^-1 (T / mu) by forward substitution distributed over the lanes
    double cur = T_j * invmu_j, q_j = 0.0;
    static_for<0, NP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const double qi = bcast_lane<NP, i>(cur) / L_[i * LD + i];
      q_j = (j == i) ? qi : q_j;
      cur -= (j > i) ? L_[j * LD + i] * qi : 0.0;
    });
    v0[j] = q_j;
    __syncthreads();
    double zn = 0.0;  // zneg_j = -k_j/2 sum_i Z[i][j] q[i]
#pragma unroll
    for (int i = 0; i < NP; ++i) zn += zc[i] * v0[i];
    zn *= -0.5 * kj;
    if (valid && act) d.zneg[((long)c * d.L + l) * NP + j] = zn;
    const double* sp = d.spoly + ((long)c * d.L + l) * d.Ns;
    const double rk = 1.0 / kj;
    for (int q = 0; q < d.Ns; ++q) {
      // b_q(K) = sum_{jj>=q} jj!/q! a_jj K^-(jj-q+1), K = -k (first N eigen-columns) and +k
      double bneg = 0.0, bpos = 0.0, ratio = 1.0, pw_pos = rk, pw_neg = -rk;
      for (int jj = q; jj < d.Ns; ++jj) {
        bpos += ratio * sp[jj] * pw_pos;
        bneg += ratio * sp[jj] * pw_neg;
        ratio *= (double)(jj + 1);
        pw_pos *= rk;
        pw_neg *= -rk;
      }
      const double a = zn * bneg, b = -zn * bpos;
      // up-streams: Gp a + Gm b ; down-streams: Gm a + Gp b  (sum over eigen-index = lanes), with
      // Gp = (Y - A/k)/T, Gm = (Y + A/k)/T:  up = [Y (a+b) - A (a-b)/k]/T, down = [Y (a+b) + A (a-b)/k]/T
      const double sab = a + b, dab = (a - b) * rk;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NP; ++i) R_[i * LD + j] = (ya[i] * sab - aa[i] * dab) / d.T[i];
      __syncthreads();
      double up = 0.0;
#pragma unroll
      for (int r = 0; r < NP; ++r) up += R_[j * LD + r];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NP; ++i) R_[i * LD + j] = (ya[i] * sab + aa[i] * dab) / d.T[i];
      __syncthreads();
      double dn = 0.0;
#pragma unroll
      for (int r = 0; r < NP; ++r) dn += R_[j * LD + r];
      if (valid && act) {
        double* dq = d.dq + (((long)c * d.L + l) * d.Ns + q) * 2 * NP;
        dq[j] = up;
        dq[NP + j] = dn;
      }
    }
    __syncthreads();
  }

  // beam particular solution (:143-152, :226-231) through the spectral decomposition:
  //  s = B+ + B-, dd = B+ - B- ;  (I/mu0^2 - Qm Pm) T s = T(x+ + x-)/mu0 - Qm T (x+ - x-)
  //  T dd = mu0 [ T (x+ - x-) - Pm T s ],  Qm Pm = L^-T Z k^2 Z^T L^T
  if (d.beam) {
    const double mu0 = d.mu0[c];
    const double om = d.omega[(long)c * d.L + l];
    const double* wl = d.wleg + ((long)c * d.L + l) * P;
    const double* Ym = d.Y + (long)m * P * NP;
    const double fac = d.I0[c] / (4.0 * M_PI) * (id.mg == 0 ? 1.0 : 2.0) * om;
    const double* Y0 = d.Y0 + ((long)c * d.M + m) * P;
    double xe = 0.0, xo = 0.0, cmax = 0.0;  // X^e_j, X^o_j of this lane's stream
    for (int ell = id.mg; ell < P; ell += 2) {
      cmax = fmax(cmax, fabs(0.5 * om * wl[ell]));
      xe += fac * wl[ell] * Y0[ell] * Ym[(long)ell * NP + j];
      if (ell + 1 < P) {
        cmax = fmax(cmax, fabs(0.5 * om * wl[ell + 1]));
        xo += fac * wl[ell + 1] * Y0[ell + 1] * Ym[(long)(ell + 1) * NP + j];
      }
    }
    if (!(cmax > 1e-8)) xe = xo = 0.0;
    const double txd = 2.0 * T_j * xe * invmu_j;  // T (x+ - x-)
    v0[j] = txd;
    __syncthreads();
    double qv = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) qv += Q_[i * LD + j] * v0[i];
    const double rhat = 2.0 * T_j * xo * invmu_j / mu0 - qv;
    v1[j] = rhat;
    __syncthreads();
    double g = 0.0;  // g = L^T rhat
#pragma unroll
    for (int r = 0; r < NP; ++r) g += L_[r * LD + j] * v1[r];
    v2[j] = g;
    __syncthreads();
    double h = 0.0;  // h = Z^T g / (1/mu0^2 - k^2)
#pragma unroll
    for (int i = 0; i < NP; ++i) h += zc[i] * v2[i];
    h /= (1.0 / (mu0 * mu0) - k2);
    // e = Z h (cross-lane sum through LDS)
#pragma unroll
    for (int i = 0; i < NP; ++i) R_[i * LD + j] = zc[i] * h;
    __syncthreads();
    double e = 0.0;
#pragma unroll
    for (int r = 0; r < NP; ++r) e += R_[j * LD + r];
    // shat = L^-T e by back substitution distributed over the lanes
    double sh = 0.0;
    static_for<0, NP>([&](auto ic) {
      constexpr int i = NP - 1 - decltype(ic)::value;
      const double si = bcast_lane<NP, i>(e) / L_[i * LD + i];
      sh = (j == i) ? si : sh;
      e -= (j < i) ? L_[i * LD + j] * si : 0.0;
    });
    v1[j] = sh;
    __syncthreads();
    double tv = 0.0;  // t = L^T shat
#pragma unroll
    for (int r = 0; r < NP; ++r) tv += L_[r * LD + j] * v1[r];
    v2[j] = tv;
    __syncthreads();
    double ps = 0.0;  // Pm shat = L t
#pragma unroll
    for (int r = 0; r < NP; ++r) ps += L_[j * LD + r] * v2[r];
    const double rT = 1.0 / T_j;
    const double s_j = sh * rT;
    const double d_j = mu0 * (txd - ps) * rT;
    if (valid) {
      d.Bv[base * 2 * NP + j] = 0.5 * (s_j + d_j);
      d.Bv[base * 2 * NP + NP + j] = 0.5 * (s_j - d_j);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused eigen stage: assembly, Cholesky factors, one-sided Jacobi and the eigenvector /
// particular-solution stage in ONE kernel per (c, m, l).  F, L, Qm and k Z never leave the CU (registers + LDS):
// compared with the three-kernel pipeline this removes 10.6 KB of HBM traffic per problem (a third of the path's
// total) and the Lw / Qw workspaces.
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64, (NP <= 16 ? RTD_EIGEN_WAVES : 1)) void rtd_eigen_kernel(RtdDev d) {
  constexpr int GPW = 64 / NP;
  constexpr int LD = NP + 1;
  __shared__ double sL[GPW][NP * LD];  // Cholesky factor L of Pm
  __shared__ double sV[GPW][4][NP];
  double w[NP];  // column j of F = L^T R, then of k Z
  {  // ---- stage 1: assembly, Cholesky factors, F, Jacobi.  Only w (and the LDS tile of L) leaves this block.
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const ProbId id = locate<NP>(d);
  const int P = d.P, m = id.m, c = id.c, l = id.l;
  double* L_ = sL[grp];
  double* dinv = sV[grp][3];  // 1 / L[i][i]
  const double* wl = d.wleg + ((long)c * d.L + l) * P;
  const double om = d.omega[(long)c * d.L + l];
  const double* Ym = d.Y + (long)m * P * NP;

  // D+/D- split by parity of (l - m): Ae = 2 sum_even c_l Y_l Y_l^T, Ao likewise (:123-125)
  double acc_e[NP], acc_o[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) acc_e[i] = acc_o[i] = 0.0;
  double cmax = 0.0;
  for (int ell = id.mg; ell < P; ell += 2) {
    {
      const double cl = 0.5 * om * wl[ell];
      cmax = fmax(cmax, fabs(cl));
      const double* Yr = Ym + (long)ell * NP;
      const double coef = 2.0 * cl * Yr[j];
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_e[i] += coef * Yr[i];
    }
    if (ell + 1 < P) {
      const double cl = 0.5 * om * wl[ell + 1];
      cmax = fmax(cmax, fabs(cl));
      const double* Yr = Ym + (long)(ell + 1) * NP;
      const double coef = 2.0 * cl * Yr[j];
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_o[i] += coef * Yr[i];
    }
  }
  // "shortcut" of the reference when multiple scattering is insignificant (:119, :162-168): the layer
  // is treated as non-scattering; the general path then gives G = [[0,D],[D,0]], k = 1/mu, B = 0.
  if (!(cmax > 1e-8)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) acc_e[i] = acc_o[i] = 0.0;
  }
  const double invmu_j = d.invmu[j], S_j = d.S[j];
  {
    double pcol[NP], qcol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const double Si = d.S[i];
      pcol[i] = (i == j ? invmu_j : 0.0) - Si * acc_e[i] * S_j;  // Pm = M^-1 - S Ae S
      qcol[i] = (i == j ? invmu_j : 0.0) - Si * acc_o[i] * S_j;  // Qm = M^-1 - S Ao S
    }
    dinv[j] = cholesky_columns<NP>(pcol, j);  // Pm = L L^T
    cholesky_columns<NP>(qcol, j);  // Qm = R R^T
#pragma unroll
    for (int i = 0; i < NP; ++i) L_[i * LD + j] = pcol[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double a = 0.0;
#pragma unroll
      for (int r = i; r < NP; ++r) a += L_[r * LD + i] * qcol[r];
      w[i] = a;
    }
  }
  // one-sided Jacobi on the columns of F
  int nsweep = 0;
  bool converged = false;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int notconv = 0;
    double alpha = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) alpha += w[i] * w[i];
    JacobiStep<NP, 1>::run(w, alpha, j, notconv);
    ++nsweep;
    if (!__any(notconv)) {
      converged = true;
      break;
    }
  }
  if (threadIdx.x == 0 && nsweep > *(volatile int*)d.sweeps) atomicMax(d.sweeps, nsweep);
  if (!converged && threadIdx.x == 0) atomicOr(d.status, RTD_ST_JACOBI);  // NaN input (failed Cholesky) also ends here
  }
  // ---- stage 2: eigenvector blocks and particular solutions.  The lane's identifiers are rebuilt from an opaque copy
  //      of the lane index, so that none of them is kept (and spilled) across the Jacobi loop.
  int tx = threadIdx.x;
  asm volatile("" : "+v"(tx));
  const int grp = tx / NP, j = tx % NP;
  const ProbId id = locate<NP>(d, tx);
  const int P = d.P, m = id.m, c = id.c, l = id.l;
  const bool valid = id.valid;
  const long base = id.pid;
  double* L_ = sL[grp];
  double* v0 = sV[grp][0];
  double* v1 = sV[grp][1];
  double* v2 = sV[grp][2];
  double* dinv = sV[grp][3];  // 1 / L[i][i]
  const double* wl = d.wleg + ((long)c * d.L + l) * P;
  const double om = d.omega[(long)c * d.L + l];
  const double* Ym = d.Y + (long)m * P * NP;
  const double invmu_j = d.invmu[j], T_j = d.T[j];
  double k2 = 0.0;
#pragma unroll
  for (int i = 0; i < NP; ++i) k2 += w[i] * w[i];
  const double rk0 = fast_rsqrt(k2), kj = k2 * rk0;
  // a non-positive Cholesky pivot (phase function not positive definite after delta-M scaling) or an overflow shows up
  // as a non-finite or non-positive eigenvalue: the reference's eig / sqrt would return NaN here (:186)
  if (valid && !(k2 > 0.0 && k2 < 1e300)) atomicOr(d.status, RTD_ST_CHOL);
  double zc[NP];
  {
#pragma unroll
    for (int i = 0; i < NP; ++i) zc[i] = w[i] * rk0;
  }

  // eigenvector blocks (:190-198): V = T^-1 L^-T Z, U = (alpha+beta) V / k = -T^-1 L Z / k; stored as
  // Y = L^-T Z and A = L Z, from which Gp = (Y - A/k)/T, Gm = (Y + A/k)/T, V^-1 = A^T T, U^-1 = -k Y^T T
  double ya[NP];
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double a = zc[i];
#pragma unroll
    for (int r = i + 1; r < NP; ++r) a -= L_[r * LD + i] * ya[r];
    ya[i] = a * dinv[i];
    RTD_FENCE();
  }
  if (valid) {
    double* Ym = d.Ym + base * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) Ym[i * NP + j] = ya[i];
    d.kk[base * NP + j] = kj;
    const double* ts0 = d.taus0 + (long)c * (d.L + 1);
    d.Ek[base * NP + j] = exp(-kj * (ts0[l + 1] - ts0[l]));
  }

  // beam particular solution (:143-152, :226-231) through the spectral decomposition:
  //  s = B+ + B-, dd = B+ - B- ;  (I/mu0^2 - Qm Pm) T s = T(x+ + x-)/mu0 - Qm T (x+ - x-)
  //  T dd = mu0 [ T (x+ - x-) - Pm T s ],  Qm Pm = L^-T Z k^2 Z^T L^T
  if (d.beam) {
    const double mu0 = d.mu0[c];
    const double fac = d.I0[c] * (0.25 / M_PI) * (id.mg == 0 ? 1.0 : 2.0) * om;
    const double* Y0 = d.Y0 + ((long)c * d.M + m) * P;
    double xe = 0.0, xo = 0.0, cmax = 0.0;  // X^e_j, X^o_j of this lane's stream
    for (int ell = id.mg; ell < P; ell += 2) {
      cmax = fmax(cmax, fabs(0.5 * om * wl[ell]));
      xe += fac * wl[ell] * Y0[ell] * Ym[(long)ell * NP + j];
      if (ell + 1 < P) {
        cmax = fmax(cmax, fabs(0.5 * om * wl[ell + 1]));
        xo += fac * wl[ell + 1] * Y0[ell + 1] * Ym[(long)(ell + 1) * NP + j];
      }
    }
    if (!(cmax > 1e-8)) xe = xo = 0.0;
    const double txd = 2.0 * T_j * xe * invmu_j;  // T (x+ - x-)
    v0[j] = txd;
    __syncthreads();
    // Qm v0 through Qm = L^-T H L^-1 = Y k^2 Y^T (exact for the rotated columns whatever their convergence):
    // lane e forms k_e^2 (Y^T v0)_e, the sum over the eigen-index is a transposed reduction in registers
    double qv;
    {
      double tq = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) tq += ya[i] * v0[i];
      tq *= k2;
      double x[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = ya[i] * tq;
      qv = transpose_reduce<NP>(x, j);
    }
    const double rmu0 = fast_rcp(mu0);
    const double rhat = 2.0 * T_j * xo * invmu_j * rmu0 - qv;
    v1[j] = rhat;
    __syncthreads();
    double g = 0.0;  // g = L^T rhat
#pragma unroll
    for (int r = 0; r < NP; ++r) g += L_[r * LD + j] * v1[r];
    v2[j] = g;
    __syncthreads();
    double h = 0.0;  // h = Z^T g / (1/mu0^2 - k^2)
#pragma unroll
    for (int i = 0; i < NP; ++i) h += zc[i] * v2[i];
    h *= fast_rcp(rmu0 * rmu0 - k2);
    // shat = L^-T Z h = Y h  and  t = L^T shat = Z h  (sums over the eigen-index = lanes): no triangular solve
    double sh, e;
    {
      double x[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = ya[i] * h;
      sh = transpose_reduce<NP>(x, j);
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = zc[i] * h;
      e = transpose_reduce<NP>(x, j);
    }
    __syncthreads();
    v2[j] = e;
    __syncthreads();
    double ps = 0.0;  // Pm shat = L (L^T shat) = L t
#pragma unroll
    for (int r = 0; r < NP; ++r) ps += L_[j * LD + r] * v2[r];
    const double rT = fast_rcp(T_j);
    const double s_j = sh * rT;
    const double d_j = mu0 * (txd - ps) * rT;
    if (valid) {
      d.Bv[base * 2 * NP + j] = 0.5 * (s_j + d_j);
      d.Bv[base * 2 * NP + NP + j] = 0.5 * (s_j - d_j);
      // 1/mu0 on an eigenvalue: the reference's solve (:226-231) meets a singular matrix
      if (!(fabs(s_j) + fabs(d_j) < 1e300)) atomicOr(d.status, RTD_ST_BEAM);
    }
  }
  __syncthreads();
  // A = L Z after the beam stage: its 2 NP registers are not live while that stage runs
  double aa[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r <= i; ++r) a += L_[i * LD + r] * zc[r];
    aa[i] = a;
    RTD_FENCE();
  }
  if (valid) {
    double* Am = d.Am + base * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) Am[i * NP + j] = aa[i];
  }

  // isotropic (thermal) source, Fourier mode 0 only (subroutines.py:746-862, _assemble.py:124).
  // A wave holds layers of ONE (c, m), so the branch is wave-uniform.
  if (d.Ns > 0 && id.mg == 0) {
    const bool act = true;
    // zneg_j = -k_j/2 [Z^T L^-1 (T/mu)]_j = -k_j/2 sum_i Y[i][j] T_i/mu_i   (Y = L^-T Z: no triangular solve)
    double zn = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) zn += ya[i] * (d.T[i] * d.invmu[i]);
    zn *= -0.5 * kj;
    if (valid && act) d.zneg[((long)c * d.L + l) * NP + j] = zn;
    const double* sp = d.spoly + ((long)c * d.L + l) * d.Ns;
    const double rk = rk0;
    for (int q = 0; q < d.Ns; ++q) {
      // b_q(K) = sum_{jj>=q} jj!/q! a_jj K^-(jj-q+1), K = -k (first N eigen-columns) and +k
      double bneg = 0.0, bpos = 0.0, ratio = 1.0, pw_pos = rk, pw_neg = -rk;
      for (int jj = q; jj < d.Ns; ++jj) {
        bpos += ratio * sp[jj] * pw_pos;
        bneg += ratio * sp[jj] * pw_neg;
        ratio *= (double)(jj + 1);
        pw_pos *= rk;
        pw_neg *= -rk;
      }
      const double a = zn * bneg, b = -zn * bpos;
      // up-streams: Gp a + Gm b ; down-streams: Gm a + Gp b  (sum over eigen-index = lanes), with
      // Gp = (Y - A/k)/T, Gm = (Y + A/k)/T:  up = [Y (a+b) - A (a-b)/k]/T, down = [Y (a+b) + A (a-b)/k]/T
      const double sab = a + b, dab = (a - b) * rk;
      double xu[NP], xd[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const double py = ya[i] * sab, pa = aa[i] * dab;
        xu[i] = py - pa;
        xd[i] = py + pa;
      }
      const double rT = 1.0 / T_j;
      const double up = transpose_reduce<NP>(xu, j) * rT;
      const double dn = transpose_reduce<NP>(xd, j) * rT;
      if (valid && act) {
        double* dq = d.dq + (((long)c * d.L + l) * d.Ns + q) * 2 * NP;
        dq[j] = up;
        dq[NP + j] = dn;
      }
    }
    __syncthreads();
  }

}

// ---- Legendre tables: Ybar_l^m(x) = sqrt((l-m)!/(l+m)!) P_l^m(x) without the Condon-Shortley sign
//      (it cancels in every product the path forms); replaces scipy.special.lpmv/poch (:96-109).
__device__ __forceinline__ void ybar_column(int m, int P, double x, double* out, long stride) {
  double v = 1.0;
  const double sx = sqrt(fmax(0.0, 1.0 - x * x));
  for (int jj = 1; jj <= m; ++jj) v *= sqrt((2.0 * jj - 1.0) / (2.0 * jj)) * sx;
  for (int ell = 0; ell < m && ell < P; ++ell) out[ell * stride] = 0.0;
  if (m >= P) return;
  double ym1 = 0.0, y = v;
  out[m * stride] = y;
  for (int ell = m; ell + 1 < P; ++ell) {
    const double yn = ((2.0 * ell + 1.0) * x * y - sqrt((double)(ell + m) * (double)(ell - m)) * ym1) /
                      sqrt((double)(ell + 1 - m) * (double)(ell + 1 + m));
    ym1 = y;
    y = yn;
    out[(ell + 1) * stride] = y;
  }
}

__global__ void rtd_tables_quad_kernel(RtdDev d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // (m, i)
  if (t >= d.M * d.NP) return;
  const int m = t / d.NP, i = t % d.NP;
  double* out = d.Y + (long)m * d.P * d.NP + i;
  if (i >= d.N) {
    for (int ell = 0; ell < d.P; ++ell) out[(long)ell * d.NP] = 0.0;
    return;
  }
  ybar_column(d.m0 + d.mstep * m, d.P, d.mu[i], out, d.NP);
}

__global__ void rtd_tables_mu0_kernel(RtdDev d) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (c, m)
  if (t >= (long)d.C * d.M) return;
  const int c = (int)(t / d.M), m = (int)(t % d.M);
  ybar_column(d.m0 + d.mstep * m, d.P, -d.mu0[c], d.Y0 + t * d.P, 1);
  if (m == 0) {  // beam attenuation at the scaled layer boundaries, used by the boundary-condition kernel
    const double rmu0 = 1.0 / d.mu0[c];
    for (int l = 0; l <= d.L; ++l) d.att[(long)c * (d.L + 1) + l] = exp(-d.taus0[(long)c * (d.L + 1) + l] * rmu0);
  }
}

}  // namespace

void rtd_launch_tables(const RtdDev& d, hipStream_t s, bool with_quad) {
  if (with_quad) {
    const int n = d.M * d.NP;
    hipLaunchKernelGGL(rtd_tables_quad_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d);
  }
  if (d.beam) {
    const long n = (long)d.C * d.M;
    hipLaunchKernelGGL(rtd_tables_mu0_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, d);
  }
}

void rtd_launch_eig(const RtdDev& d, hipStream_t s, int part) {
  // part 0: assembly + Cholesky + F, 1: Jacobi, 2: eigenvector blocks / particular solutions
  const int gpw = 64 / d.NP;
  const dim3 grid((unsigned)((long)d.C * d.M * ((d.L + gpw - 1) / gpw)));
  // One fused kernel (launched as part 1).  RTD_EIG32_SPLIT=1 selects, for NP = 32, the earlier three-kernel pipeline
  // that exchanges F, L, Qm, k Z through the Ym / Am / Lw / Qw buffers (A/B: 25.6 ms against 17.1 ms fused on cfg5).
#define RTD_EIG_FUSED(NPV)                                                                      \
  case NPV:                                                                                     \
    if (part == 1) hipLaunchKernelGGL(rtd_eigen_kernel<NPV>, grid, dim3(64), 0, s, d);          \
    break;
  switch (d.NP) {
    RTD_EIG_FUSED(4)
    RTD_EIG_FUSED(8)
    RTD_EIG_FUSED(16)
    case 32:
      if (d.Lw == nullptr) {  // default: fused (the plan did not allocate the exchange buffers)
        if (part == 1) hipLaunchKernelGGL(rtd_eigen_kernel<32>, grid, dim3(64), 0, s, d);
        break;
      }
      if (part == 0) hipLaunchKernelGGL(rtd_asm_kernel<32>, grid, dim3(64), 0, s, d);
      if (part == 1) hipLaunchKernelGGL(rtd_jacobi_kernel<32>, grid, dim3(64), 0, s, d);
      if (part == 2) hipLaunchKernelGGL(rtd_post_kernel<32>, grid, dim3(64), 0, s, d);
      break;
    default: break;
  }
#undef RTD_EIG_FUSED
}
