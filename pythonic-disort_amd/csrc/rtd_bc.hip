// rtd_bc.hip -- boundary-condition solve across layers (coefficients C of the homogeneous solutions).
//
// Replaces _solve_for_coeffs (src/PythonicDISORT/_solve_for_coeffs.py:8-390): RHS assembly (:142-254),
// LHS assembly in banded/dense form (:276-323 / :337-380) and scipy.linalg.solve_banded /
// np.linalg.solve (:326-333 / :383).  Same linear system (same unknowns, same Stamnes-Conklin
// scaling), solved by a structured block elimination instead of a general banded LU:
//
//   interface l:  G_l [E_l C-_l ; C+_l]  -  G_{l+1} [C-_{l+1} ; E_{l+1} C+_{l+1}]  =  r_l
//   The eigen stage knows G_l^-1 in closed form (G = [[V+U, V-U],[V-U, V+U]], V^-1 = Z^T L^T T,
//   U^-1 = -k Z^T L^-1 T), so multiplying the 2N continuity rows by G_l^-1 makes the x_l block diagonal:
//        E_l C-_l = Wp C-' + Wq E' C+' + rho_t ,     C+_l = Wq C-' + Wp E' C+' + rho_b ,
//   with  W = G_l^-1 G_{l+1} = [[Wp, Wq],[Wq, Wp]],  Wp/Wq = (V^-1 V' +- U^-1 U')/2.
//   With the N "carry" rows  Ta C-_l + Tb C+_l = t  (initially the top boundary condition) this gives
//        C-_l = s - S C+_l ,  S = Ta^-1 Tb, s = Ta^-1 t      (the only pivoted solve: N x N, partial pivoting)
//   and the carry of the next layer  Ta' = -(E_l S Wq + Wp), Tb' = -(E_l S Wp + Wq) E', t' = rho_t - E_l (s - S rho_b).
//   Pivots are 1 for the C+ columns and come from the carry block for the C- columns, which is the
//   order partial pivoting takes whenever E_l < 1; verified against pivoted elimination of the full
//   banded matrix to <= 5e-13 on every golden case and on thick/thin/near-conservative stress cases
//   (tools/proto_device_algo.py: check_structured).
//
// Three implementations.  16 < NQuad <= 32: rtd_bc_mfma_kernel, one wavefront per (column, mode), everything in the
// matrix-core register layout (see its comment below).  32 < NQuad <= 64: rtd_bc_tile_kernel<2>, the same on 2 x 2 tiles.
// NQuad <= 16 (and the tiled kernel's last resort for a singular carry block): rtd_iface_kernel (all (column, mode,
// interface) in parallel: Wp, Wq, rho through HBM) and rtd_sweep_kernel (per (column, mode): forward carry recursion,
// bottom boundary, backward sweep); NP lanes per problem, 64/NP problems per wavefront, lane i owns row i of the carry
// system.  (An MFMA form of the interface kernel at NP = 16 and a three-wavefront form of rtd_bc_mfma_kernel lost every
// A/B of round 2 and were removed in round 3.)
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

#include "rtd_bc_common.h"

// ------------------------------------------------------------------------------------------------
// Interface kernel: per (c, m, l < L-1):  Wp, Wq, rho_t, rho_b.
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64) void rtd_iface_kernel(RtdDev d, const int* only) {  // only != null: flagged (c, m) chains
  constexpr int GPW = 64 / NP, LD = NP + 1, Q = 2 * NP;
  __shared__ double sA[GPW][NP * LD];  // A_l  (natural [i][j])
  __shared__ double sY[GPW][NP * LD];  // Y_l
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const int Lm1 = d.L - 1;
  const long nprob = (long)d.C * d.M * Lm1;
  long pid = (long)blockIdx.x * GPW + grp;
  bool valid = pid < nprob;
  if (!valid) pid = nprob - 1;
  if (only != nullptr) {  // only the chains handed over by the tiled kernel: the others keep what that kernel stored
    valid = valid && only[pid / Lm1] != 0;
    const unsigned long long want = __ballot(valid);
    if (want == 0) return;
    // groups with nothing to do redo the work of one that has (well-defined data, no stores)
    const int src = __ffsll((long long)want) - 1;
    const int pid_w = __shfl((int)pid, src, 64);
    if (!valid) pid = pid_w;
  }
  const int l = (int)(pid % Lm1);
  const long cm = pid / Lm1;
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const long p0 = cm * d.L + l, p1 = p0 + 1;
  double* A0 = sA[grp];
  double* Y0 = sY[grp];
  {
    const double* Am = d.Am + p0 * NP * NP;
    const double* Ym = d.Ym + p0 * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      A0[i * LD + j] = Am[i * NP + j];
      Y0[i * LD + j] = Ym[i * NP + j];
    }
  }
  // V^-1 V' = A^T Y'   and   U^-1 U' = diag(k) Y^T A' diag(1/k')   (T cancels)
  const double rk1 = 1.0 / d.kk[p1 * NP + j];
  double* ws = d.Fws + (cm * Lm1 + l) * Ws<NP>::SLOT;
  if constexpr (NP <= 32) {
    // column j of Y' and A' of layer l+1
    double ycol[NP], acol[NP];
    {
      const double* Ym = d.Ym + p1 * NP * NP;
      const double* Am = d.Am + p1 * NP * NP;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        ycol[i] = Ym[i * NP + j];
        acol[i] = Am[i * NP + j];
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < NP; ++r) {
      double vv = 0.0, uu = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        vv += A0[i * LD + r] * ycol[i];
        uu += Y0[i * LD + r] * acol[i];
      }
      uu *= d.kk[p0 * NP + r] * rk1;
      if (valid) {
        ws[Ws<NP>::WP + r * NP + j] = 0.5 * (vv + uu);
        ws[Ws<NP>::WQ + r * NP + j] = 0.5 * (vv - uu);
      }
    }
  } else {
    // 128 streams: one product at a time (a column of 64 doubles each: both at once do not fit the register file); the
    // first product waits in the Wp slot
    __syncthreads();
    double col[NP];
    {
      const double* Ym = d.Ym + p1 * NP * NP;
#pragma unroll
      for (int i = 0; i < NP; ++i) col[i] = Ym[i * NP + j];
    }
#pragma unroll 2
    for (int r = 0; r < NP; ++r) {
      double vv = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) vv += A0[i * LD + r] * col[i];
      if (valid) ws[Ws<NP>::WP + r * NP + j] = vv;
    }
    {
      const double* Am = d.Am + p1 * NP * NP;
#pragma unroll
      for (int i = 0; i < NP; ++i) col[i] = Am[i * NP + j];
    }
#pragma unroll 2
    for (int r = 0; r < NP; ++r) {
      double uu = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) uu += Y0[i * LD + r] * col[i];
      uu *= d.kk[p0 * NP + r] * rk1;
      if (valid) {
        const double vv = ws[Ws<NP>::WP + r * NP + j];  // (this lane's own store)
        ws[Ws<NP>::WP + r * NP + j] = 0.5 * (vv + uu);
        ws[Ws<NP>::WQ + r * NP + j] = 0.5 * (vv - uu);
      }
    }
  }
  // particular-solution jump r_l at the interface (:184-205, :242-245) and rho = G_l^-1 r_l:
  //   rho_t/b = 1/4 [ V^-1 (r_up + r_dn) +- U^-1 (r_up - r_dn) ],  V^-1[j][i] = T_i A[i][j],  U^-1[j][i] = -k_j T_i Y[i][j]
  const double* ts0 = d.taus0 + (long)c * (d.L + 1);
  const double tb = ts0[l + 1];
  const double att = d.beam ? exp(-tb / d.mu0[c]) : 0.0;
  const int mg = d.m0 + d.mstep * m;  // the Fourier mode this local index stands for (mode shards)
  const bool iso = d.Ns > 0 && mg == 0;
  const double kj = d.kk[p0 * NP + j];
  double rt = 0.0, rb = 0.0;
#pragma unroll 4
  for (int i = 0; i < NP; ++i) {
    double ru = 0.0, rd = 0.0;
    if (d.beam) {
      ru = (d.Bv[p1 * Q + i] - d.Bv[p0 * Q + i]) * att;
      rd = (d.Bv[p1 * Q + NP + i] - d.Bv[p0 * Q + NP + i]) * att;
    }
    if (iso) {  // v_{l+1} at its top minus v_l at its bottom: the eigen kernel's boundary values (vb), no polynomial evaluated here
      const double* vb0 = d.vb + ((long)c * d.L + l) * 4 * NP;
      ru += vb0[4 * NP + i] - vb0[2 * NP + i];
      rd += vb0[5 * NP + i] - vb0[3 * NP + i];
    }
    const double Ti = d.T[i];
    const double a = Ti * A0[i * LD + j] * (ru + rd), b = -kj * Ti * Y0[i * LD + j] * (ru - rd);
    rt += a + b;
    rb += a - b;
  }
  if (valid) {
    ws[Ws<NP>::RT + j] = 0.25 * rt;
    ws[Ws<NP>::RB + j] = 0.25 * rb;
  }
}

// 128 streams (NP = 64, one chain per wavefront): the rows [Ta | Tb | t] live in LDS, not in registers (64 fully unrolled
// pivot steps over 129 registers per lane spilled 267 registers and 256 KB of code per elimination), and the elimination is a
// rolled loop: step K reads the pivot row as LDS broadcasts and updates the own row in place; the pivot row is scaled by the
// same FMA (f = 1 - 1/pivot on the pivot lane).  Columns (K, NP) and [x0, x1) of the rows are updated.
template <bool IN_LDS, typename A>
__device__ __forceinline__ decltype(auto) pick_row(A& regs, double* lds) {
  if constexpr (IN_LDS) return lds;
  else return (regs);
}
__device__ __forceinline__ void gj_rows_in_lds(double* R, const int ldr, const int j, const int x0, const int x1, int& pc) {
  constexpr int NP = 64;
  double* row = R + j * ldr;
  for (int K = 0; K < NP; ++K) {
    const float key = (pc < 0) ? fabsf((float)row[K]) : -1.0f;
    const float kmax = group_max_key<NP>(key);
    const unsigned long long bal = __ballot(key == kmax);
    const int found = __ffsll((long long)bal) - 1;  // pivot lane (wave-uniform); -1: a chain that has gone NaN
    const bool isp = j == found;
    const int src = found < 0 ? 0 : found;
    const double* prow = R + src * ldr;
    const double rp = fast_rcp(prow[K]);
    const double f = isp ? 1.0 - rp : row[K] * rp;
    if (isp) pc = K;
    // eight columns at a time, every load of a batch issued before its first store: the rows may alias each other for the
    // compiler, element by element every update waited for two LDS round trips (columns below K + 1 that a batch touches are
    // dead: nothing reads them again)
    auto batch = [&](const int c0) {
      double pv[8], rv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = prow[c0 + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) rv[e] = row[c0 + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) row[c0 + e] = fma(-f, pv[e], rv[e]);
    };
    for (int c0 = (K + 1) & ~7; c0 < NP; c0 += 8) batch(c0);
    int c = x0;
    for (; c + 8 <= x1; c += 8) batch(c);
    for (; c < x1; ++c) row[c] -= f * prow[c];
  }
}

// ------------------------------------------------------------------------------------------------
// Sweep kernel: per (c, m): forward carry recursion over the layers, bottom BC, backward sweep.
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64, (NP <= 8 ? RTD_SWEEP_WAVES : (NP == 16 ? 2 : 1))) void rtd_sweep_kernel(RtdDev d, const int* only) {
  constexpr int GPW = 64 / NP, LD = NP + 1, Q = 2 * NP;
  constexpr bool ROWS_IN_LDS = NP == 64;  // (see gj_rows_in_lds; the whole wavefront is one chain there: Wq, Wp are wave-uniform
                                          //  and come as scalar loads, not through LDS: two workgroups fit a CU)
  __shared__ double sA[GPW][ROWS_IN_LDS ? 1 : NP * LD];  // Wq (forward) / S (bottom)
  __shared__ double sB[GPW][ROWS_IN_LDS ? 1 : NP * LD];  // Wp
  __shared__ double sV[GPW][4][NP];
  constexpr int LDR = 2 * NP + 3;         // [Ta | Tb | t | bottom right-hand side], odd
  __shared__ double sR[ROWS_IN_LDS ? NP * LDR : 1];
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const long nprob = (long)d.C * d.M;
  long cm = (long)blockIdx.x * GPW + grp;
  bool valid = cm < nprob;
  if (!valid) cm = nprob - 1;
  if (only != nullptr) {  // only the chains handed over by the tiled kernel: the others keep what that kernel stored
    valid = valid && only[cm] != 0;
    const unsigned long long want = __ballot(valid);
    if (want == 0) return;
    // groups with nothing to do redo the work of one that has (well-defined data -- their own interface operators were
    // not formed -- and no stores)
    const int src = __ffsll((long long)want) - 1;
    const int cm_w = __shfl((int)cm, src, 64);
    if (!valid) cm = cm_w;
  }
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const int L = d.L, Lm1 = L - 1;
  double* A_ = sA[grp];
  double* B_ = sB[grp];
  double* v0 = sV[grp][0];
  double* v1 = sV[grp][1];
  double* v2 = sV[grp][2];
  double* v3 = sV[grp][3];
  const double* Ym = d.Ym + cm * L * NP * NP;
  const double* Am = d.Am + cm * L * NP * NP;
  const double* kk = d.kk + cm * L * NP;
  const double rTj = 1.0 / d.T[j];  // row scaling of G: Gp = (Y - A/k)/T, Gm = (Y + A/k)/T
  const double* Ek = d.Ek + cm * L * NP;
  const double* Bv = d.Bv + cm * L * Q;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  const double* dq = d.dq + (long)c * L * d.Ns * Q;
  double* wsb = d.Fws + cm * Lm1 * Ws<NP>::SLOT;
  double* coef = d.coef + cm * L * Q;
  const int mg = d.m0 + d.mstep * m;  // the Fourier mode this local index stands for (mode shards)
  const bool iso = d.Ns > 0 && mg == 0;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  // thermal particular solution of layer l at one of the layer's own boundaries (top / bottom), streams idx in [0, 2 NP): the values
  // the eigen kernel left in vb (it holds the polynomial coefficients about the layer's top, rtd_dd.h) -- no polynomial is evaluated here
  const double* vbp = d.vb + (long)c * L * 4 * NP;
  auto vedge = [&](int l, bool bottom, int idx) { return vbp[((long)l * 4 + (bottom ? 2 : 0)) * NP + idx]; };

  // carry rows, one per lane: Ta C- + Tb C+ = t.  Top boundary (down-streams at tau = 0) (:161-179, :284-285)
  double ta_regs[ROWS_IN_LDS ? 1 : NP], tb_regs[ROWS_IN_LDS ? 1 : NP], tt;
  auto&& ta = pick_row<ROWS_IN_LDS>(ta_regs, sR + j * LDR);
  auto&& tb = pick_row<ROWS_IN_LDS>(tb_regs, sR + j * LDR + NP);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const double yv = Ym[j * NP + k], av = Am[j * NP + k] / kk[k];
    ta[k] = (yv + av) * rTj;          // Gm_0
    tb[k] = (yv - av) * rTj * Ek[k];  // Gp_0 E_0
  }
  tt = d.bneg[cm * NP + j];
  if (beam) tt -= Bv[NP + j];
  if (iso) tt -= dq[NP + j];

  int pc = -1;
  for (int l = 0; l < L; ++l) {
    pc = -1;
    if constexpr (ROWS_IN_LDS) {
      sR[j * LDR + 2 * NP] = tt;
      gj_rows_in_lds(sR, LDR, j, NP, 2 * NP + 1, pc);
      tt = sR[j * LDR + 2 * NP];
    } else {
      GjStep<NP, NP, 0>::run(ta, tb, tt, pc, grp);  // lane now holds row pc of S = Ta^-1 Tb and s[pc]
    }
    // a chain that has gone NaN (failed eigen stage of its mode) finds no pivots: its lanes keep their own row index, so
    // that what they write below stays inside their group's LDS and workspace (the other chains of the wavefront are
    // other modes and other columns); the NaN coefficients raise RTD_ST_BC for this chain's mode at the end
    if (pc < 0) pc = j;
    if (l == Lm1) break;
    double* ws = wsb + (long)l * Ws<NP>::SLOT;
    __syncthreads();
    {  // stage Wq, Wp of this interface in LDS (coalesced rows); store S row and s for the backward sweep
      if constexpr (!ROWS_IN_LDS) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          A_[i * LD + j] = ws[Ws<NP>::WQ + i * NP + j];
          B_[i * LD + j] = ws[Ws<NP>::WP + i * NP + j];
        }
      }
      v0[j] = ws[Ws<NP>::RB + j];
      v1[j] = Ek[(l + 1) * NP + j];
      if (valid) {
#pragma unroll
        for (int k = 0; k < NP; ++k) ws[Ws<NP>::S + pc * NP + k] = tb[k];
        ws[Ws<NP>::SV + pc] = tt;
      }
    }
    __syncthreads();
    const double Er = Ek[l * NP + pc];
    double srb = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) srb += tb[k] * v0[k];  // (S rho_b)[pc]
    const double tnew = ws[Ws<NP>::RT + pc] - Er * (tt - srb);
    if constexpr (ROWS_IN_LDS) {
      // the S row into registers once (the rows in LDS may alias each other for the compiler), eight columns at a time
      double srow[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) srow[k] = tb[k];
      for (int c0 = 0; c0 < NP; c0 += 8) {
        double swq[8] = {}, swp[8] = {};
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            swq[e] += srow[k] * ws[Ws<NP>::WQ + k * NP + c0 + e];  // wave-uniform addresses: scalar loads
            swp[e] += srow[k] * ws[Ws<NP>::WP + k * NP + c0 + e];
          }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ta[c0 + e] = -(Er * swq[e] + ws[Ws<NP>::WP + pc * NP + c0 + e]);
          tb[c0 + e] = -(Er * swp[e] + ws[Ws<NP>::WQ + pc * NP + c0 + e]) * v1[c0 + e];  // (srow keeps the inputs)
        }
      }
    } else {
    double nbuf[NP];
#pragma unroll
    for (int cc = 0; cc < NP; ++cc) {
      double swq = 0.0, swp = 0.0;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        swq += tb[k] * A_[k * LD + cc];
        swp += tb[k] * B_[k * LD + cc];
      }
      ta[cc] = -(Er * swq + B_[pc * LD + cc]);           // Ta' = -(E S Wq + Wp)
      nbuf[cc] = -(Er * swp + A_[pc * LD + cc]) * v1[cc];  // Tb' = -(E S Wp + Wq) E'  (tb is still an input)
      RTD_FENCE();
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) tb[k] = nbuf[k];
    }
    tt = tnew;
  }

  // ---- bottom boundary (up-streams at tau_L) (:208-232, :248-254, :288-293):  Ba C- + Bb C+ = br,
  //      with C- = s - S C+  ->  (Bb - Ba S) C+ = br - Ba s.
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NP; ++k)
    if constexpr (!ROWS_IN_LDS) A_[pc * LD + k] = tb[k];  // S at its true row index
  int* lane_of_row = reinterpret_cast<int*>(v3);  // rows in LDS: row r of S is the Tb part of lane lane_of_row[r]
  if constexpr (ROWS_IN_LDS) lane_of_row[pc] = j;
  v0[pc] = tt;                                            // s
  __syncthreads();
  {
    const int l = Lm1;
    const double* ymL = Ym + (long)l * NP * NP;
    const double* amL = Am + (long)l * NP * NP;
    const double* kl = kk + (long)l * NP;
    const double att = beam ? exp(-ts0[L] / mu0) : 0.0;
    // Ba = Gp - R Gm, Bb = Gm - R Gp  built from  P = (I - R) Y / T-rows and  Qd = (I + R) A / (k T-rows):
    //   Gp = P0 - Q0, Gm = P0 + Q0 with P0 = Y/T, Q0 = A/(kT)  =>  Ba = (P0 - R P0) - (Q0 + R Q0), Bb = (P0 - R P0) + (Q0 + R Q0)
    double pa[NP], qa[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      pa[k] = ymL[j * NP + k] * rTj;
      qa[k] = amL[j * NP + k] * rTj;
    }
    double br = d.bpos[cm * NP + j];
    if (mg < d.NBDRF) {
      const double delta = (mg == 0) ? 2.0 : 1.0;
      const double* qt = d.bdrfq + (((long)c * d.NBDRF + mg) * NP + j) * NP;
      double rbm = 0.0, rvm = 0.0;
      for (int j2 = 0; j2 < NP; ++j2) {
        const double Rij = delta * qt[j2] * d.mu[j2] * d.w[j2] / d.T[j2];  // R = (1 + delta_m0) q (mu w), times 1/T_j2
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          pa[k] -= Rij * ymL[j2 * NP + k];
          qa[k] += Rij * amL[j2 * NP + k];
        }
        const double Rraw = Rij * d.T[j2];
        if (beam) rbm += Rraw * Bv[l * Q + NP + j2];
        if (iso) rvm += Rraw * vedge(l, true, NP + j2);
      }
      if (beam) {
        const double Xs = mu0 * d.I0[c] / M_PI * d.bdrfq0[((long)c * d.NBDRF + mg) * NP + j];
        br += (Xs + rbm - Bv[l * Q + j]) * att;
      }
      if (iso) br += rvm - vedge(l, true, j);
    } else {
      if (beam) br -= Bv[l * Q + j] * att;
      if (iso) br -= vedge(l, true, j);
    }
    double ba[NP], bb[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const double qk = qa[k] / kl[k];
      ba[k] = pa[k] - qk;
      bb[k] = pa[k] + qk;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) ba[k] *= Ek[l * NP + k];
    // am = Bb - Ba S,  bvec = br - Ba s
    double am[NP], dummy[1] = {0.0};
#pragma unroll
    for (int cc = 0; cc < NP; ++cc) {
      double a = bb[cc];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if constexpr (ROWS_IN_LDS) a -= ba[k] * sR[lane_of_row[k] * LDR + NP + cc];
        else a -= ba[k] * A_[k * LD + cc];
      }
      am[cc] = a;
      RTD_FENCE();
    }
    double bvec = br;
#pragma unroll
    for (int k = 0; k < NP; ++k) bvec -= ba[k] * v0[k];
    int pc2 = -1;
    if constexpr (ROWS_IN_LDS) {  // (the Ta part of the rows is free; the S rows stay where they are)
#pragma unroll
      for (int k = 0; k < NP; ++k) ta[k] = am[k];
      sR[j * LDR + 2 * NP + 1] = bvec;
      gj_rows_in_lds(sR, LDR, j, 2 * NP + 1, 2 * NP + 2, pc2);
      bvec = sR[j * LDR + 2 * NP + 1];
    } else {
      GjStep<NP, 1, 0>::run(am, dummy, bvec, pc2, grp);  // lane holds C+[pc2]
    }
    if (pc2 < 0) pc2 = j;  // (NaN chain, as above)
    v1[pc2] = bvec;
    __syncthreads();
    double cmin = tt;  // C-[pc] = s[pc] - S[pc][:] C+
#pragma unroll
    for (int k = 0; k < NP; ++k) cmin -= tb[k] * v1[k];
    v2[pc] = cmin;
    __syncthreads();
    if (valid) {
      coef[(long)l * Q + j] = v2[j];
      coef[(long)l * Q + NP + j] = v1[j];
      // singular system (the reference's solve_banded / solve raises LinAlgError, :326-333, :383)
      if (!(fabs(v2[j]) + fabs(v1[j]) < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
    }
  }
  // ---- backward sweep: C+_l = Wq C-' + Wp E' C+' + rho_b ;  C-_l = s - S C+_l
  if constexpr (NP == 16) {
    // lane j keeps C-_l[j], C+_l[j]; the other lanes' values arrive by DPP row broadcasts (no LDS, no barriers)
    double cmj = v2[j], cpj = v1[j];
    for (int l = Lm1 - 1; l >= 0; --l) {
      const double* ws = wsb + (long)l * Ws<NP>::SLOT;
      double wq[NP], wp[NP], sr[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        wq[k] = ws[Ws<NP>::WQ + j * NP + k];
        wp[k] = ws[Ws<NP>::WP + j * NP + k];
        sr[k] = ws[Ws<NP>::S + j * NP + k];
      }
      double cp = ws[Ws<NP>::RB + j];
      double cmin = ws[Ws<NP>::SV + j];
      const double ecp = Ek[(l + 1) * NP + j] * cpj;  // E'_j C+'_j
      static_for<0, NP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        cp += wq[k] * bcast16<k>(cmj) + wp[k] * bcast16<k>(ecp);
      });
      static_for<0, NP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        cmin -= sr[k] * bcast16<k>(cp);
      });
      cmj = cmin;
      cpj = cp;
      if (valid) {
        coef[(long)l * Q + j] = cmin;
        coef[(long)l * Q + NP + j] = cp;
      }
    }
  } else {
    for (int l = Lm1 - 1; l >= 0; --l) {
      const double* ws = wsb + (long)l * Ws<NP>::SLOT;
      double cp = ws[Ws<NP>::RB + j];
#pragma unroll 4
      for (int k = 0; k < NP; ++k)
        cp += ws[Ws<NP>::WQ + j * NP + k] * v2[k] + ws[Ws<NP>::WP + j * NP + k] * (Ek[(l + 1) * NP + k] * v1[k]);
      v3[j] = cp;
      __syncthreads();
      double cmin = ws[Ws<NP>::SV + j];
#pragma unroll 4
      for (int k = 0; k < NP; ++k) cmin -= ws[Ws<NP>::S + j * NP + k] * v3[k];
      __syncthreads();
      v1[j] = cp;
      v2[j] = cmin;
      if (valid) {
        coef[(long)l * Q + j] = cmin;
        coef[(long)l * Q + NP + j] = cp;
      }
      __syncthreads();
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Fused boundary-condition kernel, NP = 16: ONE wavefront per (column, mode) does the interface operators, the
// forward carry recursion, the bottom boundary and the backward sweep, with every 16 x 16 matrix held in the
// operand / accumulator layout of v_mfma_f64_16x16x4_f64 ("D layout": lane = 16 kq + col, register q holds the
// element [row 4 q + kq][col]).  In that layout one MFMA chain gives X^T Y for two D-layout matrices, X^T for
// Y = I, so the recursion is carried in transposed form:
//      H = S^T ,   Ta'^T = -(Wq^T (H E) + Wp^T) ,   Tb'^T = -E' (Wp^T (H E) + Wq^T) ,   H' = Tb'^T Ta'^-T
// (column-pivoted Gauss-Jordan on the stacked rows [Ta'^T ; Tb'^T ; t'^T]: columns = lanes, the same elimination
// as the two-kernel path seen through a transpose).  W and W^T come from the eigen stage's Y, A straight from HBM
// (each layer is read once per direction); nothing but H_l, s_l and rho_b is stored for the backward sweep, which
// applies W through its factors:  Wq x + Wp y = [A_l^T Y' (x + y) + k_l Y_l^T A' ((y - x)/k')] / 2.
// Against the two-kernel path this removes the Wp/Wq round trip through HBM (about 40 % of the stage's traffic).
// ------------------------------------------------------------------------------------------------
#ifndef RTD_BCF_WIN
#define RTD_BCF_WIN 20  // layers of small vectors resident in LDS (12.6 KB per wavefront with the save area: 12 per CU)
#endif
#include "rtd_bc_tile_common.h"

// Speculative, branch-free form of the same elimination with the diagonal as pivot at every step: straight-line
// code (the 16 steps schedule into each other), no pivot search.  A step whose diagonal candidate is more than a
// factor RTD_GJ_GROWTH smaller than another unused entry of its row raises `bad` (a zero pivot leaves inf / nan in the
// result, which the caller tests); the caller then
// repeats the elimination from its saved inputs with column pivoting (counts from a temporary statistics build).
// The pivot column is scaled by the same FMA as the others: its own broadcast value is itself, so f = 1 - 1/pivot
// gives v - f v = v / pivot.
// Threshold: a multiplier |f| > RTD_GJ_GROWTH flags the elimination.  The pivoted redo is slow (a rolled loop on the LDS
// copy of the inputs, one stacked row per lane; its register-resident predecessor took ~45 000 cycles per elimination and
// cost 20 % of kernel time at threshold 8, where 3.1 % of the eliminations are flagged).  64 is the classical relaxed
// threshold of sparse direct solvers (u = 1/64: local growth <= 65, i.e. ~1e-14 instead of 1e-16 relative): well under
// 1 % flagged, parity against the oracle unchanged to all printed digits (1.62e-11 abs, 4.10e-10 rel; also at 512).
#ifndef RTD_GJ_GROWTH
#define RTD_GJ_GROWTH 64.0
#endif
#ifndef RTD_GJ_FMAC_DPP
#define RTD_GJ_FMAC_DPP 1
#endif
template <int K, int Q0, int Q1>
struct FmacRows {  // v[q] -= bcast_K(v[q]) * f for q in [Q0, Q1)
  static __device__ __forceinline__ void run(double (&v)[4], const double f) {
    if constexpr (Q0 < Q1) {
      asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(v[Q0]) : "v"(f), "n"(K));
      FmacRows<K, Q0 + 1, Q1>::run(v, f);
    }
  }
};

template <int NB, int K>
struct GjFast {
  static __device__ __forceinline__ void run(double (&ta)[4], double (&tb)[4], double& tv, int& bad, const int col) {
    constexpr int QK = K >> 2, RK = K & 3;
    const double x = bcast_row<RK>(ta[QK], col);  // row K of Ta^T, replicated over the lane-rows
    const double xk = bcast16<K>(x);
    // one Newton step from the hardware seed (~4e-15): an inexact multiplier only perturbs entries that are never
    // read again, an inexact pivot scale perturbs column K of the result by the same relative amount
    const double r0 = __builtin_amdgcn_rcp(xk);
    const double rp = r0 * (2.0 - xk * r0);
    const double f = (col == K) ? 1.0 - rp : x * rp;
    bad |= (col > K && fabs(f) > RTD_GJ_GROWTH) ? 1 : 0;  // a zero pivot shows up as a non-finite result (checked by the caller)
#if RTD_GJ_FMAC_DPP
    // v -= bcast_K(v) * f as ONE instruction per register: v_fmac_f64 with a row_newbcast DPP source (the only DPP control
    // the DP ALU has), the source being the accumulator itself.  The compiler's hazard recogniser does not see VALU writes
    // made inside inline asm; a DPP read needs two wait states after a VALU write of the same VGPR: every register
    // touched here was last written by the previous step's block (>= 9 instructions back), and the block ends with the
    // wait states that cover the compiler's own DPP / permute reads of ta in the next step.  `volatile` keeps the blocks
    // of consecutive steps in program order (the scheduler would otherwise put step K + 1's update of a register right
    // behind step K's); build.py scans the generated ISA for the hazard pattern (tools/check_dpp_hazards.py).
    FmacRows<K, QK, 4>::run(ta, f);
    FmacRows<K, 0, NB>::run(tb, f);
    asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(tv) : "v"(f), "n"(K));
#else
    static_for<QK, 4>([&](auto qc) {  // rows below 4 QK are finished: the pivot column is zero there
      constexpr int q = decltype(qc)::value;
      ta[q] = fma(-f, bcast16<K>(ta[q]), ta[q]);
    });
    static_for<0, NB>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      tb[q] = fma(-f, bcast16<K>(tb[q]), tb[q]);
    });
    tv = fma(-f, bcast16<K>(tv), tv);
#endif
    GjFast<NB, K + 1>::run(ta, tb, tv, bad, col);
  }
};
template <int NB>
struct GjFast<NB, 16> {
  static __device__ __forceinline__ void run(double (&)[4], double (&)[4], double&, int&, const int) {}
};

// Column-pivoted Gauss-Jordan on the same registers as GjFast (round 4: the elimination of the chains that are pivoted
// THROUGHOUT -- chain_needs_pivoting -- used to be the rolled LDS loop `pivoted_lds`, ~10 x a speculative elimination: a
// batch with a conservative cloud layer in every column ran 3 x slower).  Step K makes row K of Ta^T a unit vector; the pivot
// is the largest unused column of that row, with threshold 1/4 in favour of the diagonal (the rule of pivoted_lds: partial
// pivoting with threshold 1/4 bounds the growth like LAPACK's, _solve_for_coeffs.py:326-333).  One chain per wavefront: the
// pivot column is wave-uniform, its row entry comes by v_readlane, its column by ds_bpermute (a run-time lane: no DPP
// broadcast); 18 cross-lane fetches + 9 FMAs per step, no LDS memory, no barrier.  Afterwards the column that was the
// pivot of step c holds column c of Tb^T Ta^-T and t^T Ta^-T (perm[c], written to sPerm): the caller moves them back.
template <int NB, int K>
struct GjPiv {
  static __device__ __forceinline__ void run(double (&ta)[4], double (&tb)[4], double& tv, unsigned& used, int* sPerm, const int col,
                                             const int rowbase, const int lane) {
    constexpr int QK = K >> 2, RK = K & 3;
    const double x = bcast_row<RK>(ta[QK], col);  // row K of Ta^T, replicated over the lane-rows
    const float key = ((used >> col) & 1u) ? -1.0f : fabsf((float)x);
    const float kmax = group_max_key<16>(key);
    const float kd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(key), 0x150 + K, 0xF, 0xF, true));
    int pcol = K;
    if (!(kd >= 0.25f * kmax && kd > 0.0f)) {  // (wave-uniform: the four lane-rows hold the same row)
      const unsigned long long bal = __ballot(key == kmax) & 0xFFFFull;
      pcol = bal ? __ffsll((long long)bal) - 1 : K;  // (a chain that has gone NaN has no candidate: the diagonal, NaN stays NaN)
    }
    pcol = __builtin_amdgcn_readfirstlane(pcol);
    const double xp = readlane_f64(x, pcol);
    const double rp = fast_rcp(xp);
    // the pivot column is scaled by 1 / pivot EXACTLY (one multiplication): the speculative form's v - (1 - 1/p) v loses
    // |1 - 1/p| / |1/p| ulps, harmless for its growth-limited pivots, not for the chains that are here because their result
    // hangs on the last digits (random32/3736: 1.6e-2 of the field scale with that form, 1.9e-4 -- the reference's level -- so)
    const bool isp = col == pcol;
    const double f = x * rp;
    const int addr = (rowbase | pcol) << 2;
    auto upd = [&](double& v) {
      const double bp = bperm(addr, v);
      v = isp ? bp * rp : fma(-f, bp, v);
    };
    static_for<QK, 4>([&](auto qc) {  // rows below 4 QK are finished: unit vectors with a zero in every unused column
      upd(ta[decltype(qc)::value]);
    });
    static_for<0, NB>([&](auto qc) { upd(tb[decltype(qc)::value]); });
    upd(tv);
    used |= 1u << pcol;
    if (lane == 0) sPerm[K] = pcol;
    if constexpr (K + 1 < 16) GjPiv<NB, K + 1>::run(ta, tb, tv, used, sPerm, col, rowbase, lane);
  }
};

// Four wavefronts per SIMD: <= 128 registers and <= 10 KB of LDS each, so the kernel prefetches one layer ahead, forms the
// interface products one after the other (one accumulator set live), takes exp(-k dtau) from memory, saves t^T once and
// rotates two operand sets in the backward sweep (128 VGPRs, 36 dwords spilled outside the two loops -- profiles/rNN_kernel_resources.json; 9.8 KB).  The kernel
// is bound by the latency of its dependent chains: a three-wavefront form (two layers of prefetch, the products side by
// side, exp(-k dtau) in the LDS window: 168 VGPRs, 12.7 KB; removed in round 3) took 4.19-4.25 ms per 2 048 cfg4 columns
// against this form's 4.08-4.11 ms, and padded to 7 / 9 / 12 wavefronts per CU 5.46 / 4.82 / 4.33 ms.
__global__ __launch_bounds__(64, 4) void rtd_bc_mfma_kernel(RtdDev d) {
  constexpr int NP = 16, Q = 32, NN = NP * NP;
  const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15, rowbase = lane & 48;
  const long cm = chain_of_block(blockIdx.x, d.C, d.M);
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const int L = d.L, Lm1 = L - 1;
#ifdef RTD_BC_ALIAS_EXPERIMENT
  // Timing experiment of a PROFILING build only (-DRTD_BC_ALIAS_EXPERIMENT=1 | 2 | 3 through RTD_EXTRA_FLAGS; results are
  // garbage; never part of the shipped library: tests/test_host_logic.py checks that no result-changing switch is read from
  // the environment): the chains read the eigen stage's hand-off of only 32 chains (bit 0: 2.6 MB, served by the L2s) or of
  // 2 048 chains (bit 1: 168 MB, served by the Infinity Cache); with both bits the factors H, s, rho_b of the forward sweep
  // are aliased to 32 chains as well.  What the kernel takes then is the floor that any scheme for cutting its HBM traffic
  // can approach (profiles/archive/r03_experiments.json: bc_traffic_floor).
  const long cmr = (RTD_BC_ALIAS_EXPERIMENT & 1) ? cm % 32 : (RTD_BC_ALIAS_EXPERIMENT & 2) ? cm % 2048 : cm;
  const long cmw = ((RTD_BC_ALIAS_EXPERIMENT & 3) == 3) ? cm % 32 : cm;
#else
  const long cmr = cm, cmw = cm;
#endif
  const double* Ym = d.Ym + cmr * L * NN;
  const double* Am = d.Am + cmr * L * NN;
  const double* kk = d.kk + cmr * L * NP;
  const double* Ek = d.Ek + cmr * L * NP;
  const double* Bv = d.Bv + cmr * L * Q;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  const double* dq = d.dq + (long)c * L * d.Ns * Q;
  double* wsb = d.Fws + cmw * Lm1 * Ws<NP>::SLOT;
  double* coef = d.coef + cm * L * Q;
  const int mg = d.m0 + d.mstep * m;  // the Fourier mode this local index stands for (mode shards)
  const bool iso = d.Ns > 0 && mg == 0;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  // careful: this chain takes the column-pivoted elimination throughout (GjPiv, registers) -- the rule of chain_needs_pivoting,
  // or the test hook RTD_BC_FORCE_PIVOT=2 for every chain.  force_redo (RTD_BC_FORCE_PIVOT=1): every speculative elimination is
  // declared failed and redone by pivoted_lds, the path of the rare real failures (the suite runs under both).
  const int careful = chain_needs_pivoting(d, RTD_BC_CAREFUL_ALL_MODE0 ? mg == 0 : iso, kk, L, NP) | ((d.flags >> 2) & 1);
  const int force_redo = d.flags & 1;
  // thermal particular solution of layer l at one of the layer's own boundaries (top / bottom), streams idx in [0, 2 NP): the values
  // the eigen kernel left in vb (it holds the polynomial coefficients about the layer's top, rtd_dd.h) -- no polynomial is evaluated here
  const double* vbp = d.vb + (long)c * L * 4 * NP;
  auto vedge = [&](int l, bool bottom, int idx) { return vbp[((long)l * 4 + (bottom ? 2 : 0)) * NP + idx]; };
  // (kq, col are passed in so that the loops can hand over an opaque copy of the lane index: the compiler then
  //  rebuilds the few address VGPRs per iteration instead of keeping dozens of hoisted ones alive and spilling)
  auto load_d = [](const double* p, const int kq, const int col) {  // row-major 16 x 16 matrix -> D layout
    v4f64 x;
    x[0] = p[kq * NP + col];
    x[1] = p[(4 + kq) * NP + col];
    x[2] = p[(8 + kq) * NP + col];
    x[3] = p[(12 + kq) * NP + col];
    return x;
  };
  auto load_row = [](const double* p, const int kq) {  // a 16-vector in row form
    v4f64 x;
    x[0] = p[kq];
    x[1] = p[4 + kq];
    x[2] = p[8 + kq];
    x[3] = p[12 + kq];
    return x;
  };
  auto make_eye = [](const int kq, const int col) {
    v4f64 e;
    e[0] = (kq == col) ? 1.0 : 0.0;
    e[1] = (4 + kq == col) ? 1.0 : 0.0;
    e[2] = (8 + kq == col) ? 1.0 : 0.0;
    e[3] = (12 + kq == col) ? 1.0 : 0.0;
    return e;
  };
  // ---- LDS.  (1) The save area of the running elimination, read back only when its speculation fails; the backward sweep
  //      stages its results there.  (2) The chain's small vectors for a window of RTD_BCF_WIN layers, filled by coalesced
  //      loads: exp(-k_l dtau_l), the stream scaling T, and the particular solution p_l(tau) = B_l exp(-tau / mu0) + v_l(tau)
  //      (beam + thermal polynomial) in the form each sweep needs -- forward: its jump at the interface below layer l,
  //      r_l = p_(l+1)(tau_(l+1)) - p_l(tau_(l+1)) (:184-205, :242-245); backward: its value at the top of layer l, which the
  //      fused evaluation adds.  The loops then read them in whatever form they need (row form = a broadcast read)
  //      without keeping dozens of registers in flight, know nothing of beam or thermal sources, and have no global load
  //      that is consumed right away (one such load makes the wave wait for everything it has in flight: the counter is
  //      in-order).
  constexpr int W = RTD_BCF_WIN;
  constexpr int NSV = 8 * 64 + 16;  // the save area: rows of Ta^T, Tb^T (4 x 64 each) and one copy of t^T
  constexpr int NSTG = NSV / 64;                    // result rows it can stage in the backward sweep
  __shared__ double sSaveFlat[NSV];
  double (*const sSave)[64] = reinterpret_cast<double (*)[64]>(sSaveFlat);
  __shared__ double sPs[W][Q];  // forward: r_l; backward: p_l(tau_l)
  __shared__ double sT[2][NP];  // T and 1 / T
  __shared__ double sF[NP];
  // Diagnostic build (-DRTD_BCF_STAMPS): lane 0 of three chains records s_memtime at the phase boundaries and prints the
  // differences (tools/bc_phase_cycles.py formats them); this is how the stalls named in the comments were measured.
  // The stamps split basic blocks: read the stamped build's own ISA before trusting a phase (its forward loop, unlike the
  // product's, ends with a vmcnt(0) that waits for the stores).
#ifdef RTD_BCF_STAMPS
  __shared__ long long sStamp[512];
  int nstamp = 0;
#define RTD_STAMP()                                                                  \
  {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if (lane == 0 && nstamp < 512) sStamp[nstamp] = (long long)__builtin_amdgcn_s_memtime(); \
    ++nstamp;                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                               \
  }
#else
#define RTD_STAMP()
#endif
  RTD_STAMP();
  __shared__ int sPerm[NP];
  int wb = 0;                   // the window holds layers [wb, wb + W) and interfaces [wb, wb + W]
  // Column-pivoted Gauss-Jordan on the save area: the stacked rows [Ta^T ; Tb^T (with_tb) ; t^T] in the D layout, row r of
  // the stack = sSave[r >> 2][16 (r & 3) + column].  Step K makes row K of Ta^T a unit vector; the pivot is the largest
  // unused column of that row (threshold 1/4 in favour of the diagonal); afterwards the column that was the pivot of
  // step c holds column c of Tb^T Ta^-T and t^T Ta^-T.  Rolled loops, one stacked row per lane: slow and small -- it runs
  // for the few eliminations whose speculation fails and must not cost the others registers.
  auto pivoted_lds = [&](const bool with_tb) {
    __syncthreads();
    double* const sM = &sSave[0][0];
    unsigned int used = 0;
    const int r = lane;
    const bool mine = r < NP || (with_tb && r < 2 * NP) || r == 2 * NP;
    double* const myrow = sM + (r >> 2) * 64 + 16 * (r & 3);
#pragma unroll 1
    for (int K = 0; K < NP; ++K) {
      const double* rowK = sM + (K >> 2) * 64 + 16 * (K & 3);
      float key = (lane < NP && !((used >> lane) & 1u)) ? fabsf((float)rowK[lane]) : -1.0f;
      const float kd = __shfl(key, K, 64);
      int idx = lane;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {  // argmax over lanes 0..15 (lowest index among equals)
        const float k2 = __shfl_xor(key, o, 64);
        const int i2 = __shfl_xor(idx, o, 64);
        if (k2 > key || (k2 == key && i2 < idx)) {
          key = k2;
          idx = i2;
        }
      }
      const float kmax = __shfl(key, 0, 64);
      const int pcol = (kd >= 0.25f * kmax && kd > 0.0f) ? K : __shfl(idx, 0, 64);
      used |= 1u << pcol;
      const double rp = 1.0 / rowK[pcol];
      if (lane < NP) sF[lane] = (lane == pcol) ? 0.0 : rowK[lane] * rp;
      if (lane == 0) sPerm[K] = pcol;
      __syncthreads();
      if (mine) {
        const double mp = myrow[pcol];
#pragma unroll 4
        for (int jj = 0; jj < NP; ++jj) myrow[jj] -= sF[jj] * mp;
        myrow[pcol] = mp * rp;
      }
      __syncthreads();
    }
  };
  // Two forms.  `fill` is the plain loop, used for the refills inside the sweeps (chains deeper than the window): it pays
  // the memory latency once per 64 elements but costs the loops no registers.  `fill_all` issues all loads of a fill before
  // the first is used (indices clamped, not predicated): one latency instead of ten; it holds ~70 registers and is used
  // where few others are live -- before the forward loop and at the turn into the backward sweep, i.e. for every fill of a
  // chain that fits the window.
  auto fill = [&](const int base, const bool backward) {
    __syncthreads();
    wb = base;
    const int nl = min(W, L - base);
#pragma unroll 1
    for (int e = lane; e < nl * Q; e += 64) {
      const int l = base + (e >> 5), i = e & 31;
      double v = 0.0;
      if (backward) {
        if (beam) v = Bv[l * Q + i] * d.att[(long)c * (L + 1) + l];
        if (iso) v += vedge(l, false, i);
      } else if (l < Lm1) {
        if (beam) v = (Bv[(l + 1) * Q + i] - Bv[l * Q + i]) * d.att[(long)c * (L + 1) + l + 1];
        if (iso) v += vedge(l + 1, false, i) - vedge(l, true, i);
      }
      (&sPs[0][0])[e] = v;
    }
    
    __syncthreads();
  };
  auto fill_all = [&](const int base, const bool backward) {
    __syncthreads();
    wb = base;
    const int nl = min(W, L - base);
    constexpr int NE = W * Q / 64, NK = (W * NP + 63) / 64;
    static_assert(W * Q % 64 == 0, "whole passes of the wavefront");
    const double* att = d.att + (long)c * (L + 1);
    double b1[NE], b0[NE], at[NE], ek[NK];
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = min(lane + 64 * it, nl * Q - 1), l = base + (e >> 5), i = e & 31;
      const int lt = backward ? l : min(l + 1, Lm1);  // forward: the jump B_(l+1) - B_l (zero at the last layer)
      b1[it] = Bv[lt * Q + i];
      b0[it] = Bv[l * Q + i];
      at[it] = att[backward ? l : l + 1];
    }
    
#pragma unroll
    for (int it = 0; it < NE; ++it) {
      const int e = lane + 64 * it, l = base + (e >> 5), i = e & 31;
      if (e < nl * Q) {
        double v = beam ? (backward ? b1[it] : b1[it] - b0[it]) * at[it] : 0.0;
        if (iso) {
          if (backward) v += vedge(l, false, i);
          else if (l < Lm1) v += vedge(l + 1, false, i) - vedge(l, true, i);
        }
        (&sPs[0][0])[e] = v;
      }
    }
    
    __syncthreads();
  };
  // (everything the prologue needs from memory is requested here, ahead of the window's fill: one memory latency, the
  //  fill's, for all of it instead of three in a row)
  v4f64 a0 = load_d(Am, kq, col), y0 = load_d(Ym, kq, col);
  const int lsecond = min(1, Lm1);
  v4f64 a1 = a0, y1 = y0;  // (lean: layer l + 1 is requested at the top of iteration l)
  double k0c = kk[col], k1c = k0c;
  
  const v4f64 k_row = load_row(kk, kq);
  v4f64 e_row_g = k_row;
  e_row_g = load_row(Ek, kq);
  double tv = d.bneg[cm * NP + col];
  const double bv_top = beam ? Bv[NP + col] : 0.0, dq_top = iso ? dq[NP + col] : 0.0;
  {
    const double t = d.T[col];
    if (lane < NP) {
      sT[0][lane] = t;
      sT[1][lane] = fast_rcp(t);
    }
  }
  fill_all(0, false);
  tv -= bv_top + dq_top;

  // carry rows (transposed): top boundary, down-streams at tau = 0 (:161-179, :284-285):
  //   Ta = Gm_0 = (Y + A/k)/T-rows,  Tb = Gp_0 E_0 = (Y - A/k)/T-rows E_0
  double ta[4], tb[4];
  {
    const double rT_col = sT[1][col];
    const v4f64 eye = make_eye(kq, col);
    const v4f64 yt = mm_t(y0, eye), at = mm_t(a0, eye);
    const v4f64 e_row = e_row_g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double av = at[q] * fast_rcp(k_row[q]);
      ta[q] = (yt[q] + av) * rT_col;
      tb[q] = (yt[q] - av) * rT_col * e_row[q];
    }
  }

  // One layer per iteration, in this order:
  //   loads     the operands of layer l + 2 (consumed by the NEXT iteration);
  //   products  M1 = A_l^T Y', M2s = diag(k) Y_l^T A' diag(1/k') and their transposes, the particular-solution jump rho:
  //             independent of the carry, in one block with
  //   the elimination  [Ta^T ; Tb^T ; t^T] -> H = S^T, s   whose dependent steps leave the issue slots the MFMAs fill;
  //   a wait    for the loads (issued a whole elimination ago), THEN the stores of H, s, rho_b: with loads and stores
  //             both in flight every wait is a wait for the youngest store's acknowledgement (~4 000 cycles, measured:
  //             40 % of an iteration when the stores came before the wait); stored here they have a whole iteration;
  //   the carry of the next layer.
  // (Everything the prologue loaded is waited for here: a register that still had a load pending at the loop's entry would
  //  get a counted wait at its first use in the loop, and a counted wait is a wait for every OLDER operation -- the
  //  previous iteration's stores.)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int l = 0; l < L; ++l) {
    // (kq, col from an opaque copy of the lane index: the compiler then rebuilds the few address VGPRs per iteration
    //  instead of keeping dozens of hoisted ones alive and spilling)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
    const int ln = min(l + 1, Lm1), l2 = min(l + 2, Lm1);
    v4f64 a2 = a1, y2 = y1, e1r_g = a1;
    double k2c = k1c, e0c_g = 0.0;
    {  // one layer ahead: consumed behind this iteration's elimination
      a1 = load_d(Am + (long)ln * NN, kq, col);
      y1 = load_d(Ym + (long)ln * NN, kq, col);
      k1c = kk[ln * NP + col];
      e0c_g = Ek[l * NP + col];
      e1r_g = load_row(Ek + ln * NP, kq);
    }
    if (ln >= wb + W) fill(l, false);
    RTD_STAMP();  // 4 l + 1: loop top (register rotation, loads issued)
    const int r0 = l - wb, r1 = ln - wb;
    // ---- elimination (speculative: the diagonal as pivot; see GjFast)
    {
      // (the speculative elimination stays unconditional and in ONE basic block with the products above: its dependent steps
      //  leave the issue slots the MFMAs fill.  A chain that is pivoted throughout pays for it as well -- straight-line, cheap --
      //  and then redoes the elimination from the saved inputs; wrapping the two forms in if / else split the block and cost
      //  the kernel 4 %)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sSave[q][lane] = ta[q];
        sSave[4 + q][lane] = tb[q];
      }
      if (kq == 0) sSave[8][col] = tv;
      int bad = 0;
      GjFast<4, 0>::run(ta, tb, tv, bad, col);
      bad |= (fabs(tv) + fabs(tb[0]) + fabs(tb[1]) + fabs(tb[2]) + fabs(tb[3]) < 1e300) ? 0 : 1;  // zero pivot: inf / nan
      bad |= force_redo | careful;
      if (__any(bad)) {  // some diagonal pivot was too small, or the chain is pivoted throughout: the saved inputs once more
        if (careful) {  // (wave-uniform) in registers
          __syncthreads();
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ta[q] = sSave[q][lane];
            tb[q] = sSave[4 + q][lane];
          }
          tv = sSave[8][col];
          unsigned used = 0;
          GjPiv<4, 0>::run(ta, tb, tv, used, sPerm, col, rowbase, lane);
          __syncthreads();
          const int addr = (rowbase | sPerm[col]) << 2;  // unknown `col` sits in the column that was the pivot of step `col`
#pragma unroll
          for (int q = 0; q < 4; ++q) tb[q] = bperm(addr, tb[q]);
          tv = bperm(addr, tv);
          __syncthreads();
        } else {
          pivoted_lds(true);
          const int src = sPerm[col];  // unknown `col` sits in the column that was the pivot of step `col`
#pragma unroll
          for (int q = 0; q < 4; ++q) tb[q] = sSave[4 + q][16 * kq + src];
          tv = sSave[8][src];
          __syncthreads();
        }
      }
    }
    RTD_STAMP();  // 4 l + 2: elimination
    if (l == Lm1) break;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads of this iteration, before the stores go out
    double* ws = wsb + (long)l * Ws<NP>::SLOT;
#pragma unroll
    for (int q = 0; q < 4; ++q) ws[Ws<NP>::S + (4 * q + kq) * NP + col] = tb[q];
    if (kq == 0) ws[Ws<NP>::SV + col] = tv;
    {
      // the same quantities with one accumulator set live at a time (Y_l is scaled in place: it is not used again)
      v4f64 a1s;
      {
        const double rk1c = fast_rcp(k1c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y0[q] *= k0c;
          a1s[q] = a1[q] * rk1c;
        }
      }
      double rt = 0.0, rb = 0.0;
      {
        const v4f64 t_row = load_row(&sT[0][0], kq);
        const v4f64 ru = load_row(&sPs[r0][0], kq), rd = load_row(&sPs[r0][NP], kq);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double pa = t_row[s] * a0[s] * (ru[s] + rd[s]), pb = -t_row[s] * y0[s] * (ru[s] - rd[s]);
          rt += pa + pb;
          rb += pa - pb;
        }
        rt = 0.25 * sum_kq(rt);
        rb = 0.25 * sum_kq(rb);
      }
      if (kq == 0) ws[Ws<NP>::RB + col] = rb;
      const double e0c = e0c_g;
      v4f64 he;
#pragma unroll
      for (int q = 0; q < 4; ++q) he[q] = tb[q] * e0c;
      const v4f64 hcur = {tb[0], tb[1], tb[2], tb[3]};
      const double tnew = rt - e0c * (tv - col_dot(hcur, col_to_row(rb, rowbase, kq)));  // t' = rho_t - E (s - S rho_b)
      // X + M1^T = M1^T (H E + I) and Z - M2s^T = M2s^T (H E - I): the unit matrix goes onto the diagonal of H E before each product,
      // so that neither transpose is ever formed -- not by a second MFMA chain (FP64 MFMAs and FP64 vector instructions share the DP
      // ALUs on gfx950, tools/hiptests/dp_coissue.hip: 8 MFMAs were 512 cycles of the very resource the kernel is short of), and not
      // through LDS either (rounds 3-4: two round trips on the chain's critical path per layer)
      v4f64 hep, hem;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double one = (4 * q + kq == col) ? 1.0 : 0.0;
        hep[q] = he[q] + one;
        hem[q] = he[q] - one;
      }
      v4f64 s1;  // M1^T (H E + I)
      {
        const v4f64 m1 = mm_t(a0, y1);
        s1 = mm_t(m1, hep);
      }
      {
        const v4f64 m2s = mm_t(y0, a1s);
        const v4f64 dd4 = mm_t(m2s, hem);  // M2s^T (H E - I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double dd = dd4[q];
          ta[q] = -0.5 * (s1[q] - dd);
          tb[q] = -0.5 * (s1[q] + dd) * e1r_g[q];
        }
      }
      tv = tnew;
    }
#ifdef RTD_BCF_STAMPS
    asm volatile("" ::"v"(ta[0]), "v"(ta[3]), "v"(tb[0]), "v"(tb[3]), "v"(tv));
#endif
    RTD_STAMP();  // 4 l + 4: carry
    a0 = a1;
    y0 = y1;
    k0c = k1c;
    
  }

  // ---- bottom boundary (up-streams at tau_L) (:208-232, :248-254, :288-293):  Ba C- + Bb C+ = br,
  //      with C- = s - S C+  ->  (Bb - Ba S) C+ = br - Ba s;  Ba = [(I - R) P0 - (I + R) Q0] E_L, Bb = (I - R) P0 + (I + R) Q0,
  //      P0 = Y/T-rows, Q0 = A/(k T-rows), R = (1 + delta_m0) q (mu w).  Solved transposed like the carry.
  double cminus, cplus;
  {
    const int l = Lm1, rL = Lm1 - wb;
    // (everything the block needs from memory is requested here, before the first use: one memory latency instead of
    //  one per group of loads)
    const bool refl = mg < d.NBDRF;
    const double kLc = kk[l * NP + col];
    double br = d.bpos[cm * NP + col];
    const double att = beam ? d.att[(long)c * (L + 1) + L] : 0.0;
    const double bvc = beam ? Bv[l * Q + col] : 0.0;
    v4f64 qr = {0.0, 0.0, 0.0, 0.0}, mur = qr, wr = qr, bdr = qr;
    double I0c = 0.0, q0c = 0.0;
    if (refl) {
      const double* qt = d.bdrfq + (((long)c * d.NBDRF + mg) * NP + col) * NP;  // row j = col of q^m
      qr = load_row(qt, kq);
      mur = load_row(d.mu, kq);
      wr = load_row(d.w, kq);
      if (beam) {
        bdr = load_row(Bv + l * Q + NP, kq);
        I0c = d.I0[c];
        q0c = d.bdrfq0[((long)c * d.NBDRF + mg) * NP + col];
      }
    }
    v4f64 eLr_g = qr;
    eLr_g = load_row(Ek + l * NP, kq);
    const double rkLc = fast_rcp(kLc);
    const v4f64 eLr = eLr_g;
    const v4f64 rT_row = load_row(&sT[1][0], kq), eye = make_eye(kq, col);
    v4f64 p0, q0, x1 = eye, x2 = eye, rtr = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      p0[q] = y0[q] * rT_row[q];
      q0[q] = a0[q] * rT_row[q] * rkLc;
    }
    if (refl) {
      const double delta = (mg == 0) ? 2.0 : 1.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double r = delta * qr[q] * mur[q] * wr[q];  // R^T in the D layout: [row j2 = 4 q + kq][col j] = R[j][j2]
        rtr[q] = r;
        x1[q] -= r;
        x2[q] += r;
      }
    }
    const v4f64 g1 = mm_t(p0, x1), g2 = mm_t(q0, x2);  // ((I - R) P0)^T, ((I + R) Q0)^T
    v4f64 bat;
    double mt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bat[q] = eLr[q] * (g1[q] - g2[q]);
      mt[q] = g1[q] + g2[q];
    }
    const v4f64 hcur = {tb[0], tb[1], tb[2], tb[3]};
    const v4f64 sd = mm_t(hcur, eye);     // S in the D layout
    const v4f64 hb = mm_t(sd, bat);       // S^T Ba^T
#pragma unroll
    for (int q = 0; q < 4; ++q) mt[q] -= hb[q];  // (Bb - Ba S)^T
    const double tL = iso ? ts0[L] : 0.0;
    if (refl) {
      if (beam) {
        const double rbm = col_dot(rtr, bdr);
        const double Xs = mu0 * I0c / M_PI * q0c;
        br += (Xs + rbm - bvc) * att;
      }
      if (iso) {
        v4f64 vr;
#pragma unroll
        for (int q = 0; q < 4; ++q) vr[q] = vedge(l, true, NP + 4 * q + kq);
        br += col_dot(rtr, vr) - vedge(l, true, col);
      }
    } else {
      br -= bvc * att;
      if (iso) br -= vedge(l, true, col);
    }
    double rhs = br - col_dot(bat, col_to_row(tv, rowbase, kq));
    double none[4] = {0.0, 0.0, 0.0, 0.0};
    if (careful) {
      unsigned used = 0;
      GjPiv<0, 0>::run(mt, none, rhs, used, sPerm, col, rowbase, lane);
      __syncthreads();
      rhs = bperm((rowbase | sPerm[col]) << 2, rhs);
      __syncthreads();
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) sSave[q][lane] = mt[q];
      if (kq == 0) sSave[8][col] = rhs;
      int bad = 0;
      GjFast<0, 0>::run(mt, none, rhs, bad, col);
      bad |= (fabs(rhs) < 1e300) ? 0 : 1;
      bad |= force_redo;
      if (__any(bad)) {
        pivoted_lds(false);
        rhs = sSave[8][sPerm[col]];
        __syncthreads();
      }
    }
    cplus = rhs;
    cminus = tv - col_dot(hcur, col_to_row(cplus, rowbase, kq));
    // a singular system (the reference's solve_banded / solve raises LinAlgError, :326-333, :383) leaves inf / nan here,
    // and they propagate through the whole backward sweep: one test at its end is enough
  }
  // ---- fused evaluation at the layer interfaces (d.um != null): u^m = G_l [e- C- ; e+ C+] + B_l exp(-tau*/mu0) (+ v) at
  //      the top of layer l (e- = 1, e+ = E_l) and, for the last layer, at its bottom (e- = E_L, e+ = 1); with
  //      Gp = (Y - A/k)/T, Gm = (Y + A/k)/T:  up = [Y (en + ep) - A (en - ep)/k]/T,  down = [Y (en + ep) + A (en - ep)/k]/T
  //      (_assemble_intensity_and_fluxes.py:197-254).  The sums over the eigen-index are row sums over the 16 lanes of a
  //      lane-row of the D layout.
  RTD_STAMP();  // bottom boundary
  double* um = d.um ? d.um + cm * (L + 1) * Q : nullptr;
  // Results leave through LDS: a global store issued inside the sweep would make every later wait for a prefetched operand
  // a wait for that store's acknowledgement as well.  The elimination's save area is free during the sweep: it takes the
  // rows [C-_l, C+_l | u^m(tau_l)] of nine layers (row L: u^m(tau_L) only), which then go out as nine full-width stores,
  // followed by an explicit wait: the stores are then out of the way of the counted waits of the next steps.
  double* const sOut = &sSave[0][0];
  int nstage = 0, ltop = L;  // slot k holds the rows of layer / interface ltop - k
  auto flush = [&]() {
    __syncthreads();
#pragma unroll 1
    for (int k2 = 0; k2 < nstage; ++k2) {
      const long row = ltop - k2;
      const double v = sOut[k2 * 64 + lane];
      if (lane < 32) {
        if (row < L) coef[row * Q + lane] = v;
      } else if (um) {
        um[row * Q + lane - 32] = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    ltop -= nstage;
    nstage = 0;
  };
  // the homogeneous part of u^m at an interface from the two row sums  P = Y_l (e- C- + e+ C+),  Qs = A_l (e- C- - e+ C+) / k_l
  // of layer l; lanes col < 4 hold element i = 4 (col & 3) + kq of the up- and of the down-streams
  auto um_values = [&](const v4f64& P, const v4f64& Qs, const int kq, const int col, double& up, double& dn) {
    const int c3 = col & 3;
    const v4f64 rT_row = load_row(&sT[1][0], kq);
    up = 0.0;
    dn = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double u_q = (P[q] - Qs[q]) * rT_row[q], d_q = (P[q] + Qs[q]) * rT_row[q];
      up = (c3 == q) ? u_q : up;
      dn = (c3 == q) ? d_q : dn;
    }
  };
  // C-_l, C+_l and (with the fused evaluation) u^m at the top of layer l into the next slot
  auto stage = [&](const int l, const double cmn, const double cp, const v4f64& P, const v4f64& Qs, const int kq, const int col) {
    if (nstage == NSTG) flush();
    double* o = sOut + nstage * 64;
    if (kq == 0) {
      o[col] = cmn;
      o[NP + col] = cp;
    }
    if (um) {
      double up, dn;
      um_values(P, Qs, kq, col, up, dn);
      if (col < 4) {
        const int i = 4 * (col & 3) + kq;
        o[32 + i] = up + sPs[l - wb][i];
        o[32 + NP + i] = dn + sPs[l - wb][NP + i];
      }
    }
    ++nstage;
  };
  // ---- backward sweep: C+_l = Wq C-' + Wp E' C+' + rho_b ;  C-_l = s_l - S_l C+_l, with W applied through its
  //      factors: the row sums  w1 = Y' (C-' + E' C+'),  w2 = A' (E' C+' - C-') / k'  of the layer below are carried from
  //      step to step, so that a step touches the operands of ONE layer only:
  //          C+_l = rho_b + (A_l^T w1 + k_l Y_l^T w2) / 2 ,  C-_l = s_l - H_l^T C+_l ,  then w1, w2 of layer l.
  //      Interface l for free: w1 and -w2 are the two row sums of the TOP of layer l (e- = 1, e+ = E_l).  The reference
  //      evaluates tau = tau_arr[l - 1] in layer l - 1, from above the interface (_assemble_intensity_and_fluxes.py:185);
  //      the continuity rows of the boundary-condition system make the two sides equal to the residual of the solve,
  //      which is what the fused-vs-kernel test holds to 1e-13 of the field scale.
  //      The sweep moves 6 KB per layer and does ~300 instructions on them: it runs at the speed of its loads (~3 000
  //      cycles of latency against ~1 300 of arithmetic).  Three register sets rotate (the loop is unrolled by three so that
  //      the rotation is a renaming, not a copy that would wait for the load): the operands of layer l - 3 are requested
  //      when layer l has been consumed; a step has no load of its own and no store.
  struct BwSet {
    v4f64 a, y, h;
    double sl, rb, k, e;  // (e: exp(-k dtau) of the layer, lean form only)
  };
  auto load_set = [&](const int l) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int kq = lv >> 4, col = lv & 15;
    BwSet s;
    const double* w = wsb + (long)l * Ws<NP>::SLOT;
    s.a = load_d(Am + (long)l * NN, kq, col);
    s.y = load_d(Ym + (long)l * NN, kq, col);
    s.h = load_d(w + Ws<NP>::S, kq, col);
    s.sl = w[Ws<NP>::SV + col];
    s.rb = w[Ws<NP>::RB + col];
    s.k = kk[l * NP + col];
    s.e = 0.0;
    s.e = Ek[l * NP + col];
    return s;
  };
  v4f64 w1, w2;
  // (what the turn needs from memory -- the first operand set included -- is requested ahead of the window's fill: one
  //  memory latency for all of it)
  const double kL = kk[Lm1 * NP + col];
  double eL_g = 0.0;
  eL_g = Ek[Lm1 * NP + col];
  double attL = 0.0, buL = 0.0, bdL = 0.0;
  if (beam && um) {
    attL = d.att[(long)c * (L + 1) + L];
    buL = Bv[Lm1 * Q + 4 * (col & 3) + kq];
    bdL = Bv[Lm1 * Q + NP + 4 * (col & 3) + kq];
  }
  BwSet s0 = {};
  if (Lm1 > 0) s0 = load_set(Lm1 - 1);
  fill_all(max(L - W, 0), true);
  {
    const double eL = eL_g, rk = fast_rcp(kL);
    nstage = 1;  // slot 0 = row L: u^m at tau_L, the bottom of the last layer (e- = E_L, e+ = 1); no coefficients
    if (um) {
      const double en = eL * cminus, ep = cplus;
      double up, dn;
      um_values(row_dot(y0, en + ep), row_dot(a0, (en - ep) * rk), kq, col, up, dn);
      if (col < 4) {
        const int i = 4 * (col & 3) + kq;
        up = fma(buL, attL, up);
        dn = fma(bdL, attL, dn);
        if (iso) {
          up += vedge(Lm1, true, i);
          dn += vedge(Lm1, true, NP + i);
        }
        sOut[32 + i] = up;
        sOut[32 + NP + i] = dn;
      }
    }
    const double x = cminus, y = eL * cplus;
    w1 = row_dot(y0, x + y);
    w2 = row_dot(a0, (y - x) * rk);
    v4f64 nw2;
#pragma unroll
    for (int q = 0; q < 4; ++q) nw2[q] = -w2[q];
    stage(Lm1, cminus, cplus, w1, nw2, kq, col);
  }
  if (Lm1 == 0) {  // single layer: no interface, no workspace
    flush();
    if (!(fabs(cminus) + fabs(cplus) < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
    return;
  }
  auto step = [&](const int l, const BwSet& s) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
    if (l < wb) fill(max(l - W + 1, 0), true);
    const double cp = s.rb + 0.5 * (col_dot(s.a, w1) + s.k * col_dot(s.y, w2));
    RTD_STAMP();  // backward step: operands arrived, C+
    const double cmn = s.sl - col_dot(s.h, col_to_row(cp, rowbase, kq));
    cminus = cmn;
    cplus = cp;
    v4f64 nw2 = {0.0, 0.0, 0.0, 0.0};
    if (l > 0 || um) {
      const double x = cmn, y = s.e * cp;
      w1 = row_dot(s.y, x + y);
      w2 = row_dot(s.a, (y - x) * fast_rcp(s.k));
#pragma unroll
      for (int q = 0; q < 4; ++q) nw2[q] = -w2[q];
    }
    stage(l, cmn, cp, w1, nw2, kq, col);
    RTD_STAMP();  // backward step: C-, row sums, staging
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing pending at the loop's entry (see the forward loop)
  // (issued in the order of their use: a set requested after a younger one would be waited for with a smaller count; the
  //  first set came in with the fill)
  {  // two sets, unrolled by two
    BwSet s1 = load_set(max(Lm1 - 2, 0));
    __builtin_amdgcn_sched_barrier(0);
    for (int l = Lm1 - 1; l >= 0; l -= 2) {
      step(l, s0);
      s0 = load_set(max(l - 2, 0));
      if (l < 1) break;
      step(l - 1, s1);
      s1 = load_set(max(l - 3, 0));
    }
  }
  flush();
  RTD_STAMP();
#ifdef RTD_BCF_STAMPS
  if (lane == 0 && (cm == 1000 || cm == 30000 || cm == 60000 || (d.C <= 64 && (cm == 100 || cm == 300))))
    for (int i = 1; i < min(nstamp, 512); ++i) printf("ST %d %d %lld\n", (int)cm, i, sStamp[i] - sStamp[i - 1]);
#endif
  if (!(fabs(cminus) + fabs(cplus) < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
}


// ------------------------------------------------------------------------------------------------
// Fused boundary-condition kernel for NP = 16 T streams per hemisphere (T x T tiles of 16 x 16, each in the D layout):
// the 64-stream form (T = 2) of rtd_bc_mfma_kernel -- same recursion, same speculative column elimination, one
// wavefront per (column, mode); a 32 x 32 matrix is 16 doubles per lane, so the kernel is compiled for one wavefront per
// SIMD and prefetches the next layer's operands behind the elimination.  An elimination whose speculation fails is redone
// column-pivoted on an LDS copy of its inputs; a chain that still cannot be solved (singular carry block) raises its flag
// in `need_split` and leaves, and the row-per-lane kernels (rtd_iface_kernel / rtd_sweep_kernel, partial pivoting) solve
// the flagged chains afterwards.  T = 1 reproduces the
// arithmetic of rtd_bc_mfma_kernel (used as a cross-check of this generalisation in the tests, RTD_BC_TILED=1).
// ------------------------------------------------------------------------------------------------
// Growth threshold of the tiled kernel's speculative elimination.  A flagged chain is expensive here: it is redone as a
// whole by the row-per-lane kernels, whose latency per chain (50 layers x 32 pivoted steps) is that of a whole launch.
// On cfg5 (128 columns = 8 192 chains): threshold 64 flags 507 chains (the pivoted kernels then cost what they cost for all
// chains, 15 ms), 1e3: 189, 1e5: 4 (2.8 ms), 1e8: none; the error against the reference goldens is 2.19e-10 of the field
// scale at every one of them (the row-per-lane path alone: 2.1e-9).  1e6 bounds the relative perturbation of a step by
// ~1e-10; zero pivots and overflow still go to the pivoted kernels through the non-finite check.
template <int T>
__global__ __launch_bounds__(64, (T == 1 ? 2 : 1)) void rtd_bc_tile_kernel(RtdDev d, int* need_split) {
  constexpr int NP = 16 * T, Q = 2 * NP, NN = NP * NP;
  const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15, rowbase = lane & 48;
  const long cm = chain_of_block(blockIdx.x, d.C, d.M);
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const int L = d.L, Lm1 = L - 1;
  const double* Ym = d.Ym + cm * L * NN;
  const double* Am = d.Am + cm * L * NN;
  const double* kk = d.kk + cm * L * NP;
  const double* Ek = d.Ek + cm * L * NP;
  const double* Bv = d.Bv + cm * L * Q;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  const double* dq = d.dq + (long)c * L * d.Ns * Q;
  double* wsb = d.Fws + cm * Lm1 * Ws<NP>::SLOT;
  double* coef = d.coef + cm * L * Q;
  const int mg = d.m0 + d.mstep * m;
  const bool iso = d.Ns > 0 && mg == 0;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  const int careful = chain_needs_pivoting(d, iso, kk, L, NP) | (d.flags & 1) | ((d.flags >> 2) & 1);  // (RTD_BC_FORCE_PIVOT: either value)
  if ((d.flags & 2) && m % 3 == 0) {  // test hook (RTD_BC_FORCE_HANDOVER): every third Fourier mode's chain goes to the pivoted
    //                                    kernels (by mode, not by chain index: the choice must not depend on the windowing)
    if (lane == 0) {
      need_split[cm] = 1;
      *d.split_any = 1;
    }
    return;
  }
  // thermal particular solution of layer l at one of the layer's own boundaries (top / bottom), streams idx in [0, 2 NP): the values
  // the eigen kernel left in vb (it holds the polynomial coefficients about the layer's top, rtd_dd.h) -- no polynomial is evaluated here
  const double* vbp = d.vb + (long)c * L * 4 * NP;
  auto vedge = [&](int l, bool bottom, int idx) { return vbp[((long)l * 4 + (bottom ? 2 : 0)) * NP + idx]; };
  auto load_d = [](const double* p, const int kq, const int col) {  // row-major NP x NP matrix -> tiles in the D layout
    MatT<T> x;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) x.t[I][J][q] = p[(16 * I + 4 * q + kq) * NP + 16 * J + col];
    return x;
  };
  auto load_row = [](const double* p, const int kq) {
    RowT<T> x;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) x.r[I][q] = p[16 * I + 4 * q + kq];
    return x;
  };
  auto load_col = [](const double* p, const int col) {
    ColT<T> x;
#pragma unroll
    for (int J = 0; J < T; ++J) x.c[J] = p[16 * J + col];
    return x;
  };
  auto make_eye = [](const int kq, const int col) {
    MatT<T> e;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) e.t[I][J][q] = (I == J && 4 * q + kq == col) ? 1.0 : 0.0;
    return e;
  };
  auto fail_chain = [&]() {  // this chain could not be solved here: hand it to the row-per-lane kernels
    if (lane == 0) {
      need_split[cm] = 1;
      *d.split_any = 1;  // (they do not evaluate at the interfaces: the evaluation kernel then does it for the window)
    }
  };
  // The inputs of the running elimination, row-major [2 NP + 1][NP] (+1 padding): read back only when its speculation fails.
  // Then the same elimination is done once more, column-pivoted, straight on this LDS copy: every lane owns a row (rows
  // lane and lane + 64), a step reads the pivot row, picks the largest unused column, and every lane updates its row.
  // Slow (~25 us) and rare (1 of 8 192 chains x 50 layers on cfg5 at the growth threshold used).
  constexpr int LDM = NP + 1, NROW = 2 * NP + 1;
  __shared__ double sM[NROW * LDM];
  __shared__ double sF[NP];
  __shared__ int sPerm[NP];
  auto save_inputs = [&](const MatT<T>& xa, const MatT<T>& xb, const ColT<T>& xv, const int kq, const int col) {
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sM[(16 * I + 4 * q + kq) * LDM + 16 * J + col] = xa.t[I][J][q];
          sM[(NP + 16 * I + 4 * q + kq) * LDM + 16 * J + col] = xb.t[I][J][q];
        }
    if (kq == 0)
#pragma unroll
      for (int J = 0; J < T; ++J) sM[2 * NP * LDM + 16 * J + col] = xv.c[J];
  };
  // -> false when the matrix is singular (no usable pivot); on success xb, xv hold Tb^T Ta^-T and t^T Ta^-T, columns in
  //    their natural order
  auto pivoted_redo = [&](MatT<T>& xb, ColT<T>& xv, const int kq, const int col) -> bool {
    __syncthreads();
    unsigned long long used = 0;
    bool ok = true;
    for (int K = 0; K < NP; ++K) {
      float key = (lane < NP && !((used >> lane) & 1ull)) ? fabsf((float)sM[K * LDM + lane]) : -1.0f;
      int idx = lane;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {  // wave argmax
        const float k2 = __shfl_xor(key, o, 64);
        const int i2 = __shfl_xor(idx, o, 64);
        if (k2 > key || (k2 == key && i2 < idx)) {
          key = k2;
          idx = i2;
        }
      }
      const int pcol = idx;
      if (!(key > 0.0f)) ok = false;
      used |= 1ull << pcol;
      const double piv = sM[K * LDM + pcol];
      const double rp = 1.0 / piv;
      if (lane < NP) sF[lane] = (lane == pcol) ? 0.0 : sM[K * LDM + lane] * rp;
      if (lane == 0) sPerm[K] = pcol;
      __syncthreads();
      for (int row = lane; row < NROW; row += 64) {
        double* r = sM + row * LDM;
        const double mp = r[pcol];
        for (int jj = 0; jj < NP; ++jj) r[jj] -= sF[jj] * mp;
        r[pcol] = mp * rp;
      }
      __syncthreads();
    }
#pragma unroll
    for (int J = 0; J < T; ++J) {
      const int src = sPerm[16 * J + col];  // unknown 16 J + col sits in the column that was the pivot of step 16 J + col
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) xb.t[I][J][q] = sM[(NP + 16 * I + 4 * q + kq) * LDM + src];
      xv.c[J] = sM[2 * NP * LDM + src];
    }
    __syncthreads();
    return ok;
  };

  // The chain's small vectors for a window of RTD_BCT_WIN layers in LDS, as in rtd_bc_mfma_kernel: exp(-k dtau), the stream
  // scaling T, and the particular solution (beam + thermal) as the forward sweep needs it -- its jump r_l at the interface
  // below layer l (:184-205, :242-245).  The loops then have no global load that is consumed at once.
  constexpr int W = RTD_BCT_WIN;
  __shared__ double sPs[W][Q];
  __shared__ double sEk[W][NP];
  __shared__ double sT[2][NP];  // T and 1 / T
  int wb = 0;  // the window holds layers [wb, wb + W)
  // mode 1: the jump r_l (forward sweep); 2: the particular solution at the top of layer l (the fused evaluation of the
  // backward sweep); 0: exp(-k dtau) only
  auto fill = [&](const int base, const int mode) {
    __syncthreads();
    wb = base;
    const int nl = min(W, L - base);
    // (eight passes of the wavefront at a time, all their loads issued before the first is used -- indices clamped, not
    //  predicated: a loop of load / wait / write would pay the memory latency once per 64 elements)
    const double* att = d.att + (long)c * (L + 1);
    if (mode != 0)
      for (int e0 = 0; e0 < nl * Q; e0 += 8 * 64) {
        double b1[8], b0[8], at[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int e = min(e0 + lane + 64 * it, nl * Q - 1), l = base + e / Q, i = e % Q;
          const int lt = mode == 2 ? l : min(l + 1, Lm1);  // forward: the jump B_(l+1) - B_l (zero at the last layer)
          b1[it] = Bv[lt * Q + i];
          b0[it] = Bv[l * Q + i];
          at[it] = att[mode == 2 ? l : l + 1];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int e = e0 + lane + 64 * it, l = base + e / Q, i = e % Q;
          if (e < nl * Q) {
            double v = beam ? (mode == 2 ? b1[it] : b1[it] - b0[it]) * at[it] : 0.0;
            if (iso) {
              if (mode == 2) v += vedge(l, false, i);
              else if (l < Lm1) v += vedge(l + 1, false, i) - vedge(l, true, i);
            }
            (&sPs[0][0])[e] = v;
          }
        }
      }
    for (int e0 = 0; e0 < nl * NP; e0 += 8 * 64) {
      double ek[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) ek[it] = Ek[(long)base * NP + min(e0 + lane + 64 * it, nl * NP - 1)];
#pragma unroll
      for (int it = 0; it < 8; ++it)
        if (e0 + lane + 64 * it < nl * NP) (&sEk[0][0])[e0 + lane + 64 * it] = ek[it];
    }
    __syncthreads();
  };
  for (int e = lane; e < NP; e += 64) {
    const double t = d.T[e];
    sT[0][e] = t;
    sT[1][e] = fast_rcp(t);
  }
  fill(0, 1);

  MatT<T> a0 = load_d(Am, kq, col), y0 = load_d(Ym, kq, col);
  const int lsecond = min(1, Lm1);
  MatT<T> a1 = load_d(Am + (long)lsecond * NN, kq, col), y1 = load_d(Ym + (long)lsecond * NN, kq, col);
  ColT<T> k0c = load_col(kk, col), k1c = load_col(kk + lsecond * NP, col);
  ColT<T> rT_col;
  {
    const ColT<T> tc = load_col(d.T, col);
#pragma unroll
    for (int J = 0; J < T; ++J) rT_col.c[J] = fast_rcp(tc.c[J]);
  }
  // carry rows (transposed): top boundary, down-streams at tau = 0 (:161-179, :284-285):
  //   Ta = Gm_0 = (Y + A/k)/T-rows,  Tb = Gp_0 E_0 = (Y - A/k)/T-rows E_0
  MatT<T> ta, tb;
  {
    const MatT<T> eye = make_eye(kq, col);
    const MatT<T> yt = mmT<T>(y0, eye), at = mmT<T>(a0, eye);
    const RowT<T> k_row = load_row(kk, kq), e_row = load_row(&sEk[0][0], kq);
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double av = at.t[I][J][q] * fast_rcp(k_row.r[I][q]);
          ta.t[I][J][q] = (yt.t[I][J][q] + av) * rT_col.c[J];
          tb.t[I][J][q] = (yt.t[I][J][q] - av) * rT_col.c[J] * e_row.r[I][q];
        }
  }
  ColT<T> tv = load_col(d.bneg + cm * NP, col);
  if (beam) {
    const ColT<T> b = load_col(Bv + NP, col);
#pragma unroll
    for (int J = 0; J < T; ++J) tv.c[J] -= b.c[J];
  }
  if (iso) {
    const ColT<T> b = load_col(dq + NP, col);
#pragma unroll
    for (int J = 0; J < T; ++J) tv.c[J] -= b.c[J];
  }

  // One layer per iteration: loads (layer l + 2's operands, consumed by the NEXT iteration: one wavefront per SIMD, nothing
  // else hides their latency), the elimination, an explicit wait for the loads and only THEN the stores of H, s, rho_b (with
  // loads and stores both in flight every wait is a wait for the youngest store's acknowledgement), rho, the carry.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing of the prologue pending at the loop's entry (see rtd_bc_mfma_kernel)
  for (int l = 0; l < L; ++l) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
    const int ln = min(l + 1, Lm1), l2 = min(l + 2, Lm1);
    const MatT<T> a2 = load_d(Am + (long)l2 * NN, kq, col), y2 = load_d(Ym + (long)l2 * NN, kq, col);
    const ColT<T> k2c = load_col(kk + l2 * NP, col);
    if (ln >= wb + W) fill(l, 1);
    const int r0 = l - wb, r1 = ln - wb;
    // ---- elimination: [Ta^T ; Tb^T ; t^T] -> H = S^T (in tb), s (in tv)
    {
      save_inputs(ta, tb, tv, kq, col);
      int bad = 0;
      GjFastT<T, 0>::run(ta, tb, tv, bad, col);
      auto finite = [&]() {
        double chk = 0.0;
#pragma unroll
        for (int J = 0; J < T; ++J) {
          chk += fabs(tv.c[J]);
#pragma unroll
          for (int I = 0; I < T; ++I) chk += fabs(tb.t[I][J][0]) + fabs(tb.t[I][J][1]) + fabs(tb.t[I][J][2]) + fabs(tb.t[I][J][3]);
        }
        return chk < 1e300;
      };
      bad |= finite() ? 0 : 1;  // zero pivot: inf / nan
      bad |= careful;  // (RTD_BC_FORCE_PIVOT, or a chain that hangs on the last digits: chain_needs_pivoting)
      if (__any(bad)) {  // some diagonal pivot was too small: the pivoted elimination from the saved inputs
        const bool ok = pivoted_redo(tb, tv, kq, col);
        if (__any(!ok || !finite())) {
          fail_chain();
          return;
        }
      }
    }
    if (l == Lm1) break;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads of this iteration, before the stores go out
    double* ws = wsb + (long)l * Ws<NP>::SLOT;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) ws[Ws<NP>::S + (16 * I + 4 * q + kq) * NP + 16 * J + col] = tb.t[I][J][q];
    if (kq == 0)
#pragma unroll
      for (int J = 0; J < T; ++J) ws[Ws<NP>::SV + 16 * J + col] = tv.c[J];
    // ---- rho = G_l^-1 r_l for the particular-solution jump r_l at the interface:
    //   rho_t/b = 1/4 [ V^-1 (r_up + r_dn) +- U^-1 (r_up - r_dn) ],  V^-1[j][i] = T_i A[i][j],  U^-1[j][i] = -k_j T_i Y[i][j]
    MatT<T> y0s, a1s;
#pragma unroll
    for (int J = 0; J < T; ++J) {
      const double rk1 = fast_rcp(k1c.c[J]);
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y0s.t[I][J][q] = y0.t[I][J][q] * k0c.c[J];
          a1s.t[I][J][q] = a1.t[I][J][q] * rk1;
        }
    }
    ColT<T> rt, rb;
    {
      const RowT<T> t_row = load_row(&sT[0][0], kq), ru = load_row(&sPs[r0][0], kq), rd = load_row(&sPs[r0][NP], kq);
      RowT<T> vs, vd;  // T (r_up + r_dn), -T (r_up - r_dn) in row form
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vs.r[I][q] = t_row.r[I][q] * (ru.r[I][q] + rd.r[I][q]);
          vd.r[I][q] = -t_row.r[I][q] * (ru.r[I][q] - rd.r[I][q]);
        }
#pragma unroll
      for (int J = 0; J < T; ++J) {
        double pa = 0.0, pb = 0.0;
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            pa += a0.t[I][J][q] * vs.r[I][q];
            pb += y0s.t[I][J][q] * vd.r[I][q];
          }
        rt.c[J] = 0.25 * sum_kq(pa + pb);
        rb.c[J] = 0.25 * sum_kq(pa - pb);
      }
      if (kq == 0)
#pragma unroll
        for (int J = 0; J < T; ++J) ws[Ws<NP>::RB + 16 * J + col] = rb.c[J];
    }
    // ---- carry of the next layer:  Ta'^T = -(Wq^T H E + Wp^T),  Tb'^T = -E' (Wp^T H E + Wq^T)  with
    //      Wp/Wq = (M1 +- M2s)/2, M1 = A_l^T Y', M2s = diag(k) Y_l^T A' diag(1/k'):  X = M1^T H E, Z = M2s^T H E;
    //      t' = rho_t - E (s - S rho_b).  Products are formed and consumed one after the other (registers).
    const ColT<T> e0c = load_col(&sEk[r0][0], col);
    const RowT<T> e1r = load_row(&sEk[r1][0], kq);
    ColT<T> tnew;
    {
      const ColT<T> srb = col_dotT<T>(tb, col_to_rowT<T>(rb, rowbase, kq));
#pragma unroll
      for (int J = 0; J < T; ++J) tnew.c[J] = rt.c[J] - e0c.c[J] * (tv.c[J] - srb.c[J]);
    }
    MatT<T> he;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) he.t[I][J][q] = tb.t[I][J][q] * e0c.c[J];
    // the transposes M1^T, M2s^T through the (now free) save area instead of a second MFMA chain each: FP64 MFMAs occupy the
    // DP ALUs the vector instructions need (tools/hiptests/dp_coissue.hip); at T = 2 they were 64 of the 192 MFMAs of a layer
    auto transposedT = [&](const MatT<T>& mm) {
      MatT<T> t;
      __syncthreads();
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) sM[(16 * I + 4 * q + kq) * LDM + 16 * J + col] = mm.t[I][J][q];
      __syncthreads();
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) t.t[I][J][q] = sM[(16 * J + col) * LDM + 16 * I + 4 * q + kq];
      __syncthreads();
      return t;
    };
    MatT<T> s1;  // X + M1^T
    {
      const MatT<T> m1 = mmT<T>(a0, y1);
      const MatT<T> xx = mmT<T>(m1, he);
      const MatT<T> m1t = transposedT(m1);
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) s1.t[I][J][q] = xx.t[I][J][q] + m1t.t[I][J][q];
    }
    {
      const MatT<T> m2s = mmT<T>(y0s, a1s);
      const MatT<T> zz = mmT<T>(m2s, he);
      const MatT<T> m2st = transposedT(m2s);
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double dd = zz.t[I][J][q] - m2st.t[I][J][q];
            ta.t[I][J][q] = -0.5 * (s1.t[I][J][q] - dd);
            tb.t[I][J][q] = -0.5 * (s1.t[I][J][q] + dd) * e1r.r[I][q];
          }
    }
    tv = tnew;
    a0 = a1;
    y0 = y1;
    a1 = a2;
    y1 = y2;
    k0c = k1c;
    k1c = k2c;
  }

  // ---- bottom boundary (up-streams at tau_L) (:208-232, :248-254, :288-293):  Ba C- + Bb C+ = br,
  //      with C- = s - S C+  ->  (Bb - Ba S) C+ = br - Ba s;  Ba = [(I - R) P0 - (I + R) Q0] E_L, Bb = (I - R) P0 + (I + R) Q0,
  //      P0 = Y/T-rows, Q0 = A/(k T-rows), R = (1 + delta_m0) q (mu w).  Solved transposed like the carry.
  ColT<T> cminus, cplus;
  {
    const int l = Lm1;
    const RowT<T> eLr = load_row(&sEk[l - wb][0], kq), t_row = load_row(&sT[0][0], kq);
    const ColT<T> kLc = load_col(kk + l * NP, col);
    const MatT<T> eye = make_eye(kq, col);
    MatT<T> p0, q0, x1 = eye, x2 = eye, rtr;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double rTr = fast_rcp(t_row.r[I][q]);
          p0.t[I][J][q] = y0.t[I][J][q] * rTr;
          q0.t[I][J][q] = a0.t[I][J][q] * rTr * fast_rcp(kLc.c[J]);
          rtr.t[I][J][q] = 0.0;
        }
    const bool refl = mg < d.NBDRF;
    if (refl) {
      const double delta = (mg == 0) ? 2.0 : 1.0;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j2 = 16 * I + 4 * q + kq, j = 16 * J + col;  // R^T in the D layout: [row j2][col j] = R[j][j2]
            const double r = delta * d.bdrfq[(((long)c * d.NBDRF + mg) * NP + j) * NP + j2] * d.mu[j2] * d.w[j2];
            rtr.t[I][J][q] = r;
            x1.t[I][J][q] -= r;
            x2.t[I][J][q] += r;
          }
    }
    const MatT<T> g1 = mmT<T>(p0, x1), g2 = mmT<T>(q0, x2);  // ((I - R) P0)^T, ((I + R) Q0)^T
    MatT<T> bat, mt;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bat.t[I][J][q] = eLr.r[I][q] * (g1.t[I][J][q] - g2.t[I][J][q]);
          mt.t[I][J][q] = g1.t[I][J][q] + g2.t[I][J][q];
        }
    {
      const MatT<T> sd = mmT<T>(tb, eye);   // S in the D layout
      const MatT<T> hb = mmT<T>(sd, bat);   // S^T Ba^T
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) mt.t[I][J][q] -= hb.t[I][J][q];  // (Bb - Ba S)^T
    }
    ColT<T> br = load_col(d.bpos + cm * NP, col);
    const double tL = ts0[L];
    const double att = beam ? d.att[(long)c * (L + 1) + L] : 0.0;
    if (refl) {
      if (beam) {
        const ColT<T> rbm = col_dotT<T>(rtr, load_row(Bv + l * Q + NP, kq));
#pragma unroll
        for (int J = 0; J < T; ++J) {
          const double Xs = mu0 * d.I0[c] / M_PI * d.bdrfq0[((long)c * d.NBDRF + mg) * NP + 16 * J + col];
          br.c[J] += (Xs + rbm.c[J] - Bv[l * Q + 16 * J + col]) * att;
        }
      }
      if (iso) {
        RowT<T> vr;
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int q = 0; q < 4; ++q) vr.r[I][q] = vedge(l, true, NP + 16 * I + 4 * q + kq);
        const ColT<T> rv = col_dotT<T>(rtr, vr);
#pragma unroll
        for (int J = 0; J < T; ++J) br.c[J] += rv.c[J] - vedge(l, true, 16 * J + col);
      }
    } else {
#pragma unroll
      for (int J = 0; J < T; ++J) {
        if (beam) br.c[J] -= Bv[l * Q + 16 * J + col] * att;
        if (iso) br.c[J] -= vedge(l, true, 16 * J + col);
      }
    }
    ColT<T> rhs;
    {
      const ColT<T> bs = col_dotT<T>(bat, col_to_rowT<T>(tv, rowbase, kq));
#pragma unroll
      for (int J = 0; J < T; ++J) rhs.c[J] = br.c[J] - bs.c[J];
    }
    {
      MatT<T> none;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J) none.t[I][J] = v4f64{0.0, 0.0, 0.0, 0.0};
      save_inputs(mt, none, rhs, kq, col);
      int bad = 0;
      GjFastT<T, 0>::run(mt, none, rhs, bad, col);  // (the updates of the zero block cost 16 T^2 FMAs per step: once per chain)
      auto finite = [&]() {
        double chk = 0.0;
#pragma unroll
        for (int J = 0; J < T; ++J) chk += fabs(rhs.c[J]);
        return chk < 1e300;
      };
      bad |= finite() ? 0 : 1;
      bad |= careful;
      if (__any(bad)) {
        const bool ok = pivoted_redo(none, rhs, kq, col);
        if (__any(!ok || !finite())) {
          fail_chain();
          return;
        }
      }
    }
    cplus = rhs;
    {
      const ColT<T> sc = col_dotT<T>(tb, col_to_rowT<T>(cplus, rowbase, kq));
#pragma unroll
      for (int J = 0; J < T; ++J) cminus.c[J] = tv.c[J] - sc.c[J];
    }
  }
  // ---- backward sweep: C+_l = Wq C-' + Wp E' C+' + rho_b ;  C-_l = s_l - S_l C+_l, with W applied through its factors
  //      Wq x + Wp y = [A_l^T Y' (x + y) + k_l Y_l^T A' ((y - x)/k')] / 2:  the row sums  w1 = Y' (C-' + E' C+'),
  //      w2 = A' (E' C+' - C-') / k'  of the layer below are carried from step to step, so that a step touches the operands of
  //      ONE layer only.  Two operand sets rotate (loop unrolled by two: a renaming, not a copy that would wait for the
  //      load); the coefficients are staged in the (now free) save area of the elimination and leave as full-width
  //      stores every NSLOT layers -- a store inside the sweep would turn every wait for an operand into a wait for
  //      that store's acknowledgement.
  //      With the fused evaluation (d.um) a slot also takes u^m at the top of the layer: the two row sums ARE those of
  //      that interface (see rtd_bc_mfma_kernel), plus the particular solution from the window; row L is the bottom of the
  //      last layer.
  double* um = d.um ? d.um + cm * (L + 1) * Q : nullptr;
  constexpr int SLOTW = 2 * Q;  // [C-, C+ | u^m up, down]
  constexpr int NSLOT = (NROW * LDM) / SLOTW;
  double* const sOut = sM;
  int nstage = 0, ltop = L;  // slot s holds the rows of layer / interface ltop - s (row L: u^m only)
  auto flush = [&]() {
    __syncthreads();
#pragma unroll 1
    for (int s2 = 0; s2 < nstage; ++s2) {
      const long row = ltop - s2;
      for (int e = lane; e < SLOTW; e += 64) {
        const double v = sOut[s2 * SLOTW + e];
        if (e < Q) {
          if (row < L) coef[row * Q + e] = v;
        } else if (um) {
          um[row * Q + e - Q] = v;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    ltop -= nstage;
    nstage = 0;
  };
  // the homogeneous part of u^m from the row sums P = Y_l (e- C- + e+ C+), Qs = A_l (e- C- - e+ C+) / k_l: lanes col < 4 T hold
  // element i = 16 (col >> 2) + 4 (col & 3) + kq of the up- and of the down-streams
  auto um_values = [&](const RowT<T>& P, const RowT<T>& Qs, const int kq, const int col, double& up, double& dn) {
    const RowT<T> rT = load_row(&sT[1][0], kq);
    up = 0.0;
    dn = 0.0;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool mine = (col >> 2) == I && (col & 3) == q;
        const double u_q = (P.r[I][q] - Qs.r[I][q]) * rT.r[I][q], d_q = (P.r[I][q] + Qs.r[I][q]) * rT.r[I][q];
        up = mine ? u_q : up;
        dn = mine ? d_q : dn;
      }
  };
  RowT<T> w1, w2;  // w1 = P, w2 = -Qs of the top of the current layer
  auto stage = [&](const int l, const int kq, const int col) {
    if (nstage == NSLOT) flush();
    double* o = sOut + nstage * SLOTW;
    if (kq == 0)
#pragma unroll
      for (int J = 0; J < T; ++J) {
        o[16 * J + col] = cminus.c[J];
        o[NP + 16 * J + col] = cplus.c[J];
      }
    if (um) {
      RowT<T> nw2;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) nw2.r[I][q] = -w2.r[I][q];
      double up, dn;
      um_values(w1, nw2, kq, col, up, dn);
      if (col < 4 * T) {
        const int i = 16 * (col >> 2) + 4 * (col & 3) + kq;
        o[Q + i] = up + sPs[l - wb][i];
        o[Q + NP + i] = dn + sPs[l - wb][NP + i];
      }
    }
    ++nstage;
  };
  fill(max(L - W, 0), um ? 2 : 0);
  auto row_sums = [&](const MatT<T>& yl, const MatT<T>& al, const ColT<T>& kl, const ColT<T>& el) {
    ColT<T> xpy, ymx;
#pragma unroll
    for (int J = 0; J < T; ++J) {
      const double x = cminus.c[J], y = el.c[J] * cplus.c[J];
      xpy.c[J] = x + y;
      ymx.c[J] = (y - x) * fast_rcp(kl.c[J]);
    }
    w1 = row_dotT<T>(yl, xpy);
    w2 = row_dotT<T>(al, ymx);
  };
  {
    const ColT<T> kL = load_col(kk + Lm1 * NP, col), eL = load_col(&sEk[Lm1 - wb][0], col);
    nstage = 1;  // slot 0 = row L: u^m at tau_L, the bottom of the last layer (e- = E_L, e+ = 1); no coefficients
    if (um) {
      ColT<T> spe, dme;
#pragma unroll
      for (int J = 0; J < T; ++J) {
        const double en = eL.c[J] * cminus.c[J], ep = cplus.c[J];
        spe.c[J] = en + ep;
        dme.c[J] = (en - ep) * fast_rcp(kL.c[J]);
      }
      double up, dn;
      um_values(row_dotT<T>(y0, spe), row_dotT<T>(a0, dme), kq, col, up, dn);
      if (col < 4 * T) {
        const int i = 16 * (col >> 2) + 4 * (col & 3) + kq;
        if (beam) {
          const double attv = d.att[(long)c * (L + 1) + L];
          up += Bv[Lm1 * Q + i] * attv;
          dn += Bv[Lm1 * Q + NP + i] * attv;
        }
        if (iso) {
          up += vedge(Lm1, true, i);
          dn += vedge(Lm1, true, NP + i);
        }
        sOut[Q + i] = up;
        sOut[Q + NP + i] = dn;
      }
    }
    row_sums(y0, a0, kL, eL);
    stage(Lm1, kq, col);
  }
  struct BwSet {
    MatT<T> a, y, h;
    ColT<T> sl, rb, k;
  };
  auto load_set = [&](const int l) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int kq = lv >> 4, col = lv & 15;
    BwSet s;
    const double* ws = wsb + (long)l * Ws<NP>::SLOT;
    s.a = load_d(Am + (long)l * NN, kq, col);
    s.y = load_d(Ym + (long)l * NN, kq, col);
    s.h = load_d(ws + Ws<NP>::S, kq, col);
    s.sl = load_col(ws + Ws<NP>::SV, col);
    s.rb = load_col(ws + Ws<NP>::RB, col);
    s.k = load_col(kk + l * NP, col);
    return s;
  };
  auto step = [&](const int l, const BwSet& s) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
    const ColT<T> t1 = col_dotT<T>(s.a, w1), t2 = col_dotT<T>(s.y, w2);
    ColT<T> cp;
#pragma unroll
    for (int J = 0; J < T; ++J) cp.c[J] = s.rb.c[J] + 0.5 * (t1.c[J] + s.k.c[J] * t2.c[J]);
    const ColT<T> hc = col_dotT<T>(s.h, col_to_rowT<T>(cp, rowbase, kq));
#pragma unroll
    for (int J = 0; J < T; ++J) cminus.c[J] = s.sl.c[J] - hc.c[J];
    cplus = cp;
    if (l > 0 || um) row_sums(s.y, s.a, s.k, load_col(&sEk[l - wb][0], col));
    stage(l, kq, col);
  };
  // One pass of the outer loop per window of the small vectors; the requests of the sets are unconditional (past the top
  // they repeat layer 0) so that the waits stay counted.
  for (int lhi = Lm1 - 1; lhi >= 0;) {
    if (lhi < wb) fill(max(lhi - W + 1, 0), um ? 2 : 0);
    const int llo = wb;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    BwSet s0 = load_set(lhi);
    __builtin_amdgcn_sched_barrier(0);
    BwSet s1 = load_set(max(lhi - 1, 0));
    __builtin_amdgcn_sched_barrier(0);
    for (int l = lhi; l >= llo; l -= 2) {
      step(l, s0);
      s0 = load_set(max(l - 2, 0));
      if (l - 1 < llo) break;
      step(l - 1, s1);
      s1 = load_set(max(l - 3, 0));
    }
    lhi = llo - 1;
  }
  flush();
  double chk = 0.0;
#pragma unroll
  for (int J = 0; J < T; ++J) chk += fabs(cminus.c[J]) + fabs(cplus.c[J]);
  if (!(chk < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
}

}  // namespace

bool rtd_small_split() {  // RTD_SMALL_SPLIT: 2 ... 16 streams through rtd_iface_kernel + rtd_sweep_kernel + rtd_eval_kernel (A/B, tests)
  static const bool v = getenv("RTD_SMALL_SPLIT") != nullptr;
  return v;
}

bool rtd_bc_fuses_eval(const RtdDev& d) {
  // the fused kernels -- rtd_bc_small_kernel (NP <= 8), rtd_bc_mfma_kernel and the tiled one at 16 or 32 streams per hemisphere --
  // write u^m at the interfaces themselves; a window in which the tiled kernel handed a chain to the row-per-lane kernels
  // (d.split_any) is evaluated by the evaluation kernel instead (rtd_launch_eval)
  return d.NP == 16 || d.NP == 32 || (d.NP <= 8 && !rtd_small_split());
}

void rtd_launch_bc(const RtdDev& d, hipStream_t s, int part) {
  // part 0: interface operators (all interfaces in parallel), 1: carry recursion / bottom BC / backward sweep
  const int gpw = 64 / d.NP;
  const long nif = (long)d.C * d.M * (d.L - 1);
  const dim3 gi((unsigned)((nif + gpw - 1) / gpw));
  const dim3 gs((unsigned)(((long)d.C * d.M + gpw - 1) / gpw));
  const dim3 gc((unsigned)((long)d.C * d.M));
  // RTD_BC_TILED=1: the tiled fused kernel also at NP = 16 (T = 1: cross-check of the 64-stream kernel's generalisation)
  static const bool tiled16 = getenv("RTD_BC_TILED") != nullptr;
  const int* none = nullptr;
#define RTD_BC_CASE(NPV)                                                                                    \
  case NPV:                                                                                                 \
    if (part == 0 && nif > 0) hipLaunchKernelGGL(rtd_iface_kernel<NPV>, gi, dim3(64), 0, s, d, none);       \
    if (part == 1) hipLaunchKernelGGL(rtd_sweep_kernel<NPV>, gs, dim3(64), 0, s, d, none);                  \
    break;
  // 2 ... 16 streams: one fused kernel (part 1; part 0 is empty) unless RTD_SMALL_SPLIT asks for the separate ones
#define RTD_BC_SMALL_CASE(NPV)                                                                              \
  case NPV:                                                                                                 \
    if (rtd_small_split()) {                                                                                \
      if (part == 0 && nif > 0) hipLaunchKernelGGL(rtd_iface_kernel<NPV>, gi, dim3(64), 0, s, d, none);     \
      if (part == 1) hipLaunchKernelGGL(rtd_sweep_kernel<NPV>, gs, dim3(64), 0, s, d, none);                \
    } else if (part == 1) {                                                                                 \
      rtd_launch_bc_small(d, s);                                                                            \
    }                                                                                                       \
    break;
  // fused tiled kernel first (part 0); the chains whose speculative elimination failed raise need_split and are solved by
  // the pivoted row-per-lane kernels (part 1), which leave at once when none of their chains is flagged
#define RTD_BC_TILED_CASE(NPV, TV)                                                                          \
  if (part == 0) {                                                                                          \
    (void)hipMemsetAsync(d.need_split, 0, sizeof(int) * (size_t)d.C * d.M, s);                              \
    (void)hipMemsetAsync(d.split_any, 0, sizeof(int), s);                                                   \
    hipLaunchKernelGGL(rtd_bc_tile_kernel<TV>, gc, dim3(64), 0, s, d, d.need_split);                        \
  } else {                                                                                                  \
    if (nif > 0) hipLaunchKernelGGL(rtd_iface_kernel<NPV>, gi, dim3(64), 0, s, d, (const int*)d.need_split); \
    hipLaunchKernelGGL(rtd_sweep_kernel<NPV>, gs, dim3(64), 0, s, d, (const int*)d.need_split);             \
  }
  switch (d.NP) {
    RTD_BC_SMALL_CASE(4)
    RTD_BC_SMALL_CASE(8)
    case 64: {  // 66 ... 128 streams: four wavefronts per chain (rtd_bc_wide.hip) unless RTD_BC_WIDE_V1 asks for the row-per-lane kernels
      static const bool wide_v1 = getenv("RTD_BC_WIDE_V1") != nullptr;
      if (!wide_v1) {
        rtd_launch_bc_wide(d, s, part);
        break;
      }
      if (part == 0 && nif > 0) hipLaunchKernelGGL(rtd_iface_kernel<64>, gi, dim3(64), 0, s, d, none);
      if (part == 1) hipLaunchKernelGGL(rtd_sweep_kernel<64>, gs, dim3(64), 0, s, d, none);
      break;
    }
    case 16:
      if (tiled16) {
        RTD_BC_TILED_CASE(16, 1)
      } else if (part == 1) {  // the fused MFMA kernel (part 0 is empty)
        hipLaunchKernelGGL(rtd_bc_mfma_kernel, gc, dim3(64), 0, s, d);
      }
      break;
    case 32: {
      // the lean two-wavefronts-per-SIMD kernel (rtd_bc_tile2.hip) unless RTD_BC_TILE_V1 asks for rtd_bc_tile_kernel<2>
      static const bool tile_v1 = getenv("RTD_BC_TILE_V1") != nullptr;
      if (part == 0 && !tile_v1) {
        (void)hipMemsetAsync(d.need_split, 0, sizeof(int) * (size_t)d.C * d.M, s);
        (void)hipMemsetAsync(d.split_any, 0, sizeof(int), s);
        rtd_launch_bc_tile2(d, s);
      } else {
        RTD_BC_TILED_CASE(32, 2)
      }
      break;
    }
    default: break;
  }
#undef RTD_BC_CASE
#undef RTD_BC_SMALL_CASE
#undef RTD_BC_TILED_CASE
}
