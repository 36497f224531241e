// rtd_bc.hip -- boundary-condition solve across layers, one workgroup per (column, Fourier mode).
//
// Replaces _solve_for_coeffs (src/PythonicDISORT/_solve_for_coeffs.py:8-390): RHS assembly (:142-254),
// LHS assembly in banded/dense form (:276-323 / :337-380), scipy.linalg.solve_banded / np.linalg.solve
// (:326-333 / :383).  The matrix and its Stamnes-Conklin scaling are the reference's; the solver is a
// block elimination designed for a wavefront:
//
//   unknowns x_l = [C-_l ; C+_l] (Q = 2 NP per layer).  The rows that involve x_l are the NP "carry"
//   rows left over from the layers above (initially the top boundary condition) and the Q continuity
//   rows of interface l: a [3NP x (2Q+1)] panel  [carry 0 | rhs ; P_l  -Q_{l+1} | rhs].
//   Each lane owns one panel row in registers.  Gauss-Jordan elimination of the Q columns of x_l with
//   partial pivoting over the rows not yet used as pivots -- the same pivot candidates dgbsv sees,
//   because only these 3NP rows are non-zero in those columns -- leaves
//        x_l = y_l - F_l x_{l+1}      (Q pivot rows, stored to HBM, F column-major)
//   and NP rows that involve x_{l+1} only: the carry of the next panel.  The last panel (carry +
//   bottom boundary condition) gives x_{L-1}; a backward sweep x_l = y_l - F_l x_{l+1} finishes.
//   Pivot rows are broadcast with v_readlane (single-wave panels, NP <= 16) or through LDS (NP = 32).
#include "rtd_device.h"

namespace {

template <int NP>
struct BcCfg {
  static constexpr int Q = 2 * NP;
  static constexpr int R = 3 * NP;
  static constexpr int NC = 2 * Q + 1;
  static constexpr int T = (R + 63) / 64 * 64;
  static constexpr int NW = T / 64;
};

__device__ __forceinline__ double bcast_lane(double v, int src) {
  // src is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

template <int NP>
struct BcShared {
  double prow[2][BcCfg<NP>::NC];
  double xs[BcCfg<NP>::Q];
  double wmax[4];
  int wlane[4];
  int flags[BcCfg<NP>::T];
};

// Elimination of column K of the panel.
template <int NP, int K>
struct ElimStep {
  static __device__ __forceinline__ void run(double (&row)[BcCfg<NP>::NC], bool alive, int& pivcol, double& mypiv,
                                             BcShared<NP>& sh) {
    using C = BcCfg<NP>;
    const int tid = threadIdx.x;
    const bool cand = alive && pivcol < 0;
    const double val = cand ? fabs(row[K]) : -1.0;
    double f = 0.0;
    if constexpr (C::NW == 1) {
      const double vmax = wave_max(val);
      const unsigned long long bal = __ballot(val == vmax);
      const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)bal) - 1);
      const double piv = bcast_lane(row[K], src);
      const double rp = 1.0 / piv;
      const bool isp = (tid == src);
      if (isp) {
        pivcol = K;
        mypiv = row[K];
      }
      f = (alive && !isp) ? row[K] * rp : 0.0;
#pragma unroll
      for (int c = K + 1; c < C::NC; ++c) {
        const double pv = bcast_lane(row[c], src);
        row[c] -= f * pv;
      }
      if (alive && !isp) row[K] = 0.0;
    } else {
      const int wave = tid >> 6, lane = tid & 63;
      const double vmax = wave_max(val);
      const unsigned long long bal = __ballot(val == vmax);
      if (lane == 0) {
        sh.wmax[wave] = vmax;
        sh.wlane[wave] = (wave << 6) + __ffsll((long long)bal) - 1;
      }
      __syncthreads();
      int src = sh.wlane[0];
      double best = sh.wmax[0];
#pragma unroll
      for (int w = 1; w < C::NW; ++w)
        if (sh.wmax[w] > best) {
          best = sh.wmax[w];
          src = sh.wlane[w];
        }
      const bool isp = (tid == src);
      double* pr = sh.prow[K & 1];
      if (isp) {
        pivcol = K;
        mypiv = row[K];
#pragma unroll
        for (int c = K; c < C::NC; ++c) pr[c] = row[c];
      }
      __syncthreads();
      const double rp = 1.0 / pr[K];
      f = (alive && !isp) ? row[K] * rp : 0.0;
#pragma unroll
      for (int c = K + 1; c < C::NC; ++c) row[c] -= f * pr[c];
      if (alive && !isp) row[K] = 0.0;
    }
    ElimStep<NP, K + 1>::run(row, alive, pivcol, mypiv, sh);
  }
};
template <int NP>
struct ElimStep<NP, 2 * NP> {
  static __device__ __forceinline__ void run(double (&)[BcCfg<NP>::NC], bool, int&, double&, BcShared<NP>&) {}
};

// rank of this thread among the threads with flag set (block-wide)
template <int NP>
__device__ __forceinline__ int rank_of(bool flag, BcShared<NP>& sh) {
  using C = BcCfg<NP>;
  if constexpr (C::NW == 1) {
    const unsigned long long bal = __ballot(flag);
    const unsigned long long lt = (threadIdx.x == 0) ? 0ull : (~0ull >> (64 - threadIdx.x));
    return __popcll(bal & lt);
  } else {
    sh.flags[threadIdx.x] = flag ? 1 : 0;
    __syncthreads();
    int r = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) r += sh.flags[t];
    __syncthreads();
    return r;
  }
}

template <int NP>
__global__ __launch_bounds__(BcCfg<NP>::T) void rtd_bc_kernel(RtdDev d) {
  using C = BcCfg<NP>;
  constexpr int Q = C::Q, NC = C::NC;
  __shared__ BcShared<NP> sh;
  const int tid = threadIdx.x;
  const int c = blockIdx.x / d.M, m = blockIdx.x % d.M;
  const int L = d.L;
  const long cm = (long)c * d.M + m;
  const double* Gp = d.Gp + cm * L * NP * NP;
  const double* Gm = d.Gm + cm * L * NP * NP;
  const double* kk = d.kk + cm * L * NP;
  const double* Bv = d.Bv + cm * L * Q;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  const double* dq = d.dq + (long)c * L * d.Ns * Q;
  const bool iso = (d.Ns > 0) && (m == 0);
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  double* Fws = d.Fws + cm * (L - 1) * Q * Q;
  double* yws = d.yws + cm * (L - 1) * Q;
  double* coef = d.coef + cm * L * Q;

  double row[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) row[cc] = 0.0;
  bool alive = tid < C::R;
  int pivcol = -1;
  double mypiv = 1.0;

  // isotropic-source particular solution v_l(tau)[idx] = sum_q dq[l][q][idx] tau^q
  auto vpoly = [&](int l, double t, int idx) {
    double a = 0.0, tp = 1.0;
    for (int q = 0; q < d.Ns; ++q) {
      a += dq[((long)l * d.Ns + q) * Q + idx] * tp;
      tp *= t;
    }
    return a;
  };
  // continuity rows of interface l (between layers l and l+1): [P_l | -Q_{l+1} | rhs]  (:296-323, :184-205)
  auto load_interface = [&](int l, int ir) {
    const bool up = ir < NP;
    const int i = up ? ir : ir - NP;
    const double* A0 = (up ? Gp : Gm) + ((long)l * NP + i) * NP;
    const double* B0 = (up ? Gm : Gp) + ((long)l * NP + i) * NP;
    const double* A1 = (up ? Gp : Gm) + ((long)(l + 1) * NP + i) * NP;
    const double* B1 = (up ? Gm : Gp) + ((long)(l + 1) * NP + i) * NP;
    const double dt0 = ts0[l + 1] - ts0[l], dt1 = ts0[l + 2] - ts0[l + 1];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
      const double e0 = exp(-kk[l * NP + jj] * dt0), e1 = exp(-kk[(l + 1) * NP + jj] * dt1);
      row[jj] = A0[jj] * e0;
      row[NP + jj] = B0[jj];
      row[Q + jj] = -A1[jj];
      row[Q + NP + jj] = -B1[jj] * e1;
    }
    const double tb = ts0[l + 1];
    double r = 0.0;
    if (beam) r = (Bv[(l + 1) * Q + ir] - Bv[l * Q + ir]) * exp(-tb / mu0);
    if (iso) r += vpoly(l + 1, tb, ir) - vpoly(l, tb, ir);
    row[2 * Q] = r;
  };
  // bottom boundary condition row i (up-stream i at tau_L)  (:208-232, :248-254, :288-293)
  auto load_bottom = [&](int i) {
    const int l = L - 1;
    const double dt = ts0[L] - ts0[L - 1];
    const double att = beam ? exp(-ts0[L] / mu0) : 0.0;
    const double* gp = Gp + (long)l * NP * NP;
    const double* gm = Gm + (long)l * NP * NP;
    double r = d.bpos[cm * NP + i];
    if (m < d.NBDRF) {
      const double delta = (m == 0) ? 2.0 : 1.0;
      const double* qt = d.bdrfq + (((long)c * d.NBDRF + m) * NP + i) * NP;
      double acc_a[NP], acc_b[NP];
#pragma unroll
      for (int jj = 0; jj < NP; ++jj) {
        acc_a[jj] = gp[i * NP + jj];
        acc_b[jj] = gm[i * NP + jj];
      }
      double rb = 0.0, rv = 0.0;
      for (int j2 = 0; j2 < NP; ++j2) {
        const double Rij = delta * qt[j2] * d.mu[j2] * d.w[j2];  // R = (1+delta_m0) q (mu w)
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
          acc_a[jj] -= Rij * gm[j2 * NP + jj];
          acc_b[jj] -= Rij * gp[j2 * NP + jj];
        }
        if (beam) rb += Rij * Bv[l * Q + NP + j2];
        if (iso) rv += Rij * vpoly(l, ts0[L], NP + j2);
      }
#pragma unroll
      for (int jj = 0; jj < NP; ++jj) {
        row[jj] = acc_a[jj] * exp(-kk[l * NP + jj] * dt);
        row[NP + jj] = acc_b[jj];
      }
      if (beam) {
        const double Xs = mu0 * d.I0[c] / M_PI * d.bdrfq0[((long)c * d.NBDRF + m) * NP + i];
        r += (Xs + rb - Bv[l * Q + i]) * att;
      }
      if (iso) r += rv - vpoly(l, ts0[L], i);
    } else {
#pragma unroll
      for (int jj = 0; jj < NP; ++jj) {
        row[jj] = gp[i * NP + jj] * exp(-kk[l * NP + jj] * dt);
        row[NP + jj] = gm[i * NP + jj];
      }
      if (beam) r -= Bv[l * Q + i] * att;
      if (iso) r -= vpoly(l, ts0[L], i);
    }
#pragma unroll
    for (int jj = 0; jj < Q; ++jj) row[Q + jj] = 0.0;
    row[2 * Q] = r;
  };

  // ---- first panel: top boundary condition (down-streams at tau = 0)  (:161-179, :238, :284-285)
  if (tid < NP) {
    const int i = tid;
    const double dt = ts0[1] - ts0[0];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
      row[jj] = Gm[i * NP + jj];
      row[NP + jj] = Gp[i * NP + jj] * exp(-kk[jj] * dt);
    }
    double r = d.bneg[cm * NP + i];
    if (beam) r -= Bv[NP + i];
    if (iso) r -= dq[NP + i];
    row[2 * Q] = r;
  } else if (tid < C::R) {
    if (L > 1)
      load_interface(0, tid - NP);
    else if (tid < 2 * NP)
      load_bottom(tid - NP);
    else
      alive = false;
  }

  for (int l = 0; l < L; ++l) {
    ElimStep<NP, 0>::run(row, alive, pivcol, mypiv, sh);
    const double inv = 1.0 / mypiv;
    if (l < L - 1) {
      // pivot rows: x_l[pivcol] = y - F x_{l+1}; store F column-major so both sweeps are coalesced
      if (alive && pivcol >= 0) {
        double* F = Fws + (long)l * Q * Q;
#pragma unroll
        for (int cc = 0; cc < Q; ++cc) F[cc * Q + pivcol] = row[Q + cc] * inv;
        yws[(long)l * Q + pivcol] = row[2 * Q] * inv;
      }
      // next panel: un-pivoted rows become the carry, pivoted rows are reloaded
      const bool freed = alive && pivcol >= 0;
      const int rk = rank_of<NP>(freed, sh);
      if (alive && !freed) {
#pragma unroll
        for (int cc = 0; cc < Q; ++cc) {
          row[cc] = row[Q + cc];
          row[Q + cc] = 0.0;
        }
      } else if (freed) {
        if (l + 1 < L - 1) {
          load_interface(l + 1, rk);
        } else if (rk < NP) {
          load_bottom(rk);
        } else {
          alive = false;
        }
      }
      pivcol = -1;
      mypiv = 1.0;
    } else {
      if (alive && pivcol >= 0) {
        const double x = row[2 * Q] * inv;
        sh.xs[pivcol] = x;
        coef[(long)l * Q + pivcol] = x;
      }
    }
  }
  __syncthreads();
  // ---- backward sweep: x_l = y_l - F_l x_{l+1}
  for (int l = L - 2; l >= 0; --l) {
    double x = 0.0;
    if (tid < Q) {
      const double* F = Fws + (long)l * Q * Q;
      x = yws[(long)l * Q + tid];
      for (int cc = 0; cc < Q; ++cc) x -= F[cc * Q + tid] * sh.xs[cc];
    }
    __syncthreads();
    if (tid < Q) {
      sh.xs[tid] = x;
      coef[(long)l * Q + tid] = x;
    }
    __syncthreads();
  }
}

}  // namespace

void rtd_launch_bc(const RtdDev& d, hipStream_t s) {
  const dim3 grid((unsigned)(d.C * d.M));
  switch (d.NP) {
    case 4: hipLaunchKernelGGL(rtd_bc_kernel<4>, grid, dim3(BcCfg<4>::T), 0, s, d); break;
    case 8: hipLaunchKernelGGL(rtd_bc_kernel<8>, grid, dim3(BcCfg<8>::T), 0, s, d); break;
    case 16: hipLaunchKernelGGL(rtd_bc_kernel<16>, grid, dim3(BcCfg<16>::T), 0, s, d); break;
    case 32: hipLaunchKernelGGL(rtd_bc_kernel<32>, grid, dim3(BcCfg<32>::T), 0, s, d); break;
    default: break;
  }
}
