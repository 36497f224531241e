// rtd_bc_wide.hip -- boundary-condition solve at 66 ... 128 streams (NP = 64 per hemisphere), round 4.
//
// The same structured block elimination as rtd_bc.hip (see its header: _solve_for_coeffs.py:8-390 of the reference), for
// the stream counts whose 64 x 64 blocks fit no wavefront's registers.  Until round 4 these ran on the row-per-lane
// kernels of the small stream counts instantiated at NP = 64: one wavefront per chain, its 64 x 129 carry rows in LDS
// (two wavefronts per CU), every FMA of the elimination fed by two LDS reads -- 40 ms per 16 columns x 64 modes x 50 layers.
//
// Here a chain is a WORKGROUP of four wavefronts.  Lane = row of the carry system in all four; wavefront q keeps the
// columns 4 i + q (i < 16) of the row's Ta and Tb parts in registers, so a pivot step costs every wavefront the same
// work whatever the column.  What crosses wavefronts in the elimination is ONE column per pivot step (64 doubles through
// LDS, double buffered: one barrier per step); the pivot row never moves between wavefronts -- each wavefront has its
// own slice of it in the pivot lane's registers and broadcasts it with v_readlane (SGPR operands of the FMAs).
//
// The matrix products run on the matrix cores with their operands where they already are: the interface operators A^T Y'
// and Y^T A' (both factors in memory, rtd_iface_mfma_kernel: vector loads straight from the row-major Y, A in the operand
// layout) and S Wq, S Wp of the carry (S from LDS, the rows of the stored transposes by vector loads, all in flight before
// the first MFMA; the products cross LDS back into the rows-on-lanes layout of the elimination).  Earlier forms, measured
// and replaced in this order: plain FMAs with the uniform factor from scalar loads (eighty SGPRs hold one row of operands:
// bound by the latency of the scalar loads), the same with the factor as a DPP row broadcast of replicated vector loads
// (one load per sixteen FMAs; better, still latency-bound at three wavefronts per SIMD) -- DESIGN.md section 8.
// The interface operators are stored TRANSPOSED (lanes = rows contiguous): coalesced row stores in the interface kernel,
// coalesced loads in the carry's final step and in the backward sweep.
//
// Pivoting: partial pivoting on float keys exactly as the row-per-lane kernels (rtd_bc_common.h: GjStep); pivot rows are
// left unscaled and scaled once at the end of an elimination (the scaling commutes with the later eliminations of that
// row), which keeps the pivot lane on the same FMA as everyone else (f = 0) without the inexact 1 - 1/p form.
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

#include "rtd_bc_common.h"

constexpr int NP = 64, Q = 128, LDS_LD = 65;
using W = Ws<NP>;

// loads through the constant address space: a wave-uniform address becomes an s_load whatever the alias analysis thinks
// (only for memory written by EARLIER kernels of the stream)
typedef const double __attribute__((address_space(4))) kdouble;
__device__ __forceinline__ kdouble* as_k(const double* p) { return (kdouble*)(p); }

__device__ __forceinline__ double readlane_f64(const double v, const int lane) {  // lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// ------------------------------------------------------------------------------------------------
// Interface operators: per (c, m, l < L-1) the transposes of Wp, Wq = (A^T Y' +- k Y^T A' / k') / 2 and rho_t, rho_b.
// Two wavefronts (workgroups) per interface, 32 columns each, on the matrix cores: both products as v_mfma_f64_16x16x4_f64 chains
// whose operands are plain vector loads straight from the row-major Y, A of the two layers -- the A operand of a lane (k, m)
// is element [4 ks + k][16 mt + m] of A_l (or Y_l), the B operand of a lane (k, n) element [4 ks + k][32 h + 16 nt + n] of Y'
// (or A'): four 128-byte segments per load, 512 distinct bytes per instruction, twelve loads per sixteen MFMAs, the next
// k-step's operands in flight while this one's are used.  (The first version was a row-per-lane FMA loop whose uniform factors
// came through scalar loads: the same 64 multiply-adds per instruction slot -- FP64 MFMA and FMA share the DP ALUs -- but eighty
// SGPRs hold ONE row of operands, and the kernel waited for scalar loads five cycles in six: 4.9 against 2.65 ms per 76 800
// interfaces, profiles/archive/r04_pmc_many_streams.txt.)  Workgroups go to the eight XCDs round robin and every XCD has its own L2: the two
// halves of an interface (same A_l, Y_l) and the neighbouring interfaces of a chain (Y', A' of one are Y_l, A_l of the next) are
// made to meet in ONE L2 by giving XCD x the x-th contiguous eighth of the work items (the grid is rounded up to a multiple of
// eight workgroups; work items beyond the last interface leave at once).  rho rides along in the first wavefront of an interface.
// ------------------------------------------------------------------------------------------------
typedef double v4d_t __attribute__((ext_vector_type(4)));
template <int NCI>  // carried columns: 64, or 48 at 66 ... 96 streams (see below)
__global__ __launch_bounds__(64, 2) void rtd_iface_mfma_kernel(RtdDev d) {
  const int lane = threadIdx.x;
  const unsigned nb = 2u * (unsigned)d.C * (unsigned)d.M * (unsigned)(d.L - 1), per = gridDim.x / 8;
  const unsigned wi = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (wi >= nb) return;
  const int h = wi & 1;
  const long pid = wi >> 1;
  const int Lm1 = d.L - 1;
  const int l = (int)(pid % Lm1);
  const long cm = pid / Lm1;
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const long p0 = cm * d.L + l, p1 = p0 + 1;
  const double* A0 = d.Am + p0 * NP * NP;
  const double* Y0 = d.Ym + p0 * NP * NP;
  double* ws = d.Fws + (cm * Lm1 + l) * W::SLOT;
  // 66 ... 96 streams (N <= 48): the streams 48 ... 63 are padding that decouples exactly -- their rows of Y, A contribute nothing to
  // the real columns of the products and their columns are never read (the sweep kernel's NI = 12 instance): twelve k-steps instead of
  // sixteen, three column tiles instead of four
  constexpr int nc = NCI;
  const int k4 = lane >> 4, n16 = lane & 15;
  const double* a0p = A0 + k4 * NP + n16;                             // + 4 ks NP + 16 mt
  const double* y0p = Y0 + k4 * NP + n16;
  const double* y1p = d.Ym + p1 * NP * NP + k4 * NP + 32 * h + n16;   // + 4 ks NP + 16 nt
  const double* a1p = d.Am + p1 * NP * NP + k4 * NP + 32 * h + n16;
  v4d_t vv[4][2], uu[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) vv[mt][nt] = uu[mt][nt] = v4d_t{0.0, 0.0, 0.0, 0.0};
  double ca[4], cy[4], cy1[2], ca1[2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    ca[mt] = a0p[16 * mt];
    cy[mt] = y0p[16 * mt];
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    cy1[nt] = y1p[16 * nt];
    ca1[nt] = a1p[16 * nt];
  }
  constexpr int nks = nc / 4;
  const bool tile1 = nc == NP || 32 * h + 16 < nc;  // the second column tile of this half exists
#pragma unroll 2
  for (int ks = 0; ks < nks; ++ks) {
    const int kn = ks + 1 < nks ? ks + 1 : ks;
    double na[4], ny[4], ny1[2], na1[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      na[mt] = a0p[kn * 4 * NP + 16 * mt];
      ny[mt] = y0p[kn * 4 * NP + 16 * mt];
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      ny1[nt] = y1p[kn * 4 * NP + 16 * nt];
      na1[nt] = a1p[kn * 4 * NP + 16 * nt];
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      vv[mt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[mt], cy1[0], vv[mt][0], 0, 0, 0);  // A_l^T Y'
      uu[mt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cy[mt], ca1[0], uu[mt][0], 0, 0, 0);  // Y_l^T A'
    }
    if (tile1) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        vv[mt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[mt], cy1[1], vv[mt][1], 0, 0, 0);
        uu[mt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cy[mt], ca1[1], uu[mt][1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      ca[mt] = na[mt];
      cy[mt] = ny[mt];
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      cy1[nt] = ny1[nt];
      ca1[nt] = na1[nt];
    }
  }
  // The accumulators: lane (kq, n), register q of tile (mt, nt) = element [16 mt + 4 q + kq][32 h + 16 nt + n].  They are stored
  // transposed (rows of the stored matrix = columns of W); straight from the registers that is sixteen 32-byte pieces per store
  // instruction (3.6 ms per 76 800 interfaces, 2.2 with the stores elided), so each matrix crosses LDS once and leaves as 32
  // full 512-byte rows.
  {
    __shared__ double sW[32 * 66];
    const double* k0p = d.kk + p0 * NP + k4;
    const double* k1p = d.kk + p1 * NP + 32 * h + n16;
    const double rk1[2] = {fast_rcp(k1p[0]), fast_rcp(k1p[16])};
    double k0v[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) k0v[mt][q] = k0p[16 * mt + 4 * q];
#pragma unroll
    for (int which = 0; which < 2; ++which) {  // Wp, then Wq
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const double u = uu[mt][nt][q] * (k0v[mt][q] * rk1[nt]);
            sW[(16 * nt + n16) * 66 + 16 * mt + 4 * q + k4] = 0.5 * (which == 0 ? vv[mt][nt][q] + u : vv[mt][nt][q] - u);
          }
      __syncthreads();
      double* dst = ws + (which == 0 ? W::WP : W::WQ) + (32 * h) * NP + lane;
      const int ncc = nc - 32 * h < 32 ? nc - 32 * h : 32;
#pragma unroll 8
      for (int cc = 0; cc < ncc; ++cc) dst[cc * NP] = sW[cc * 66 + lane];
      __syncthreads();
    }
  }
  if (h != 0) return;
  // particular-solution jump r_l at the interface (:184-205, :242-245) and rho = G_l^-1 r_l:
  //   rho_t/b = 1/4 [ V^-1 (r_up + r_dn) +- U^-1 (r_up - r_dn) ],  V^-1[j][i] = T_i A[i][j],  U^-1[j][i] = -k_j T_i Y[i][j]
  // lane i forms T_i (r_up +- r_dn)[i] (coalesced loads); lane j then sums over i with its own column of A_l, Y_l (just read above:
  // cache hits; the row-per-lane kernel of rounds 1-3 walked i with dependent loads, one memory latency per row)
  double rsum = 0.0, rdif = 0.0;
  {
    const double* ts0 = d.taus0 + (long)c * (d.L + 1);
    const double tb = ts0[l + 1];
    const int mg = d.m0 + d.mstep * m;
    double ru = 0.0, rd = 0.0;
    if (d.beam) {
      const double att = exp(-tb / d.mu0[c]);
      ru = (d.Bv[p1 * Q + lane] - d.Bv[p0 * Q + lane]) * att;
      rd = (d.Bv[p1 * Q + NP + lane] - d.Bv[p0 * Q + NP + lane]) * att;
    }
    if (d.Ns > 0 && mg == 0) {  // v_{l+1} at its top minus v_l at its bottom, from the eigen kernel's boundary values (vb)
      const double* vb0 = d.vb + ((long)c * d.L + l) * 4 * NP;
      ru += vb0[4 * NP + lane] - vb0[2 * NP + lane];
      rd += vb0[5 * NP + lane] - vb0[3 * NP + lane];
    }
    const double Ti = d.T[lane];
    rsum = Ti * (ru + rd);
    rdif = Ti * (ru - rd);
  }
  double ra = 0.0, ry = 0.0;
#pragma unroll 16
  for (int i = 0; i < NP; ++i) {
    ra = fma(A0[i * NP + lane], readlane_f64(rsum, i), ra);
    ry = fma(Y0[i * NP + lane], readlane_f64(rdif, i), ry);
  }
  const double bb = -d.kk[p0 * NP + lane] * ry;
  ws[W::RT + lane] = 0.25 * (ra + bb);
  ws[W::RB + lane] = 0.25 * (ra - bb);
}

// max of a non-negative float key over the wavefront, in every lane's SGPR view (DPP only: quad permutes, row mirrors, then the
// GFX9 row broadcasts 15 and 31, and a v_readlane of lane 63) -- the pivot search sits on the critical path of every step and
// the LDS-crossbar forms (ds_swizzle, ds_bpermute) cost two LDS round trips there
__device__ __forceinline__ float wave_max_key(float v) {
  v = dpp_max_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_max_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_max_f32<0x141>(v);  // row_half_mirror
  v = dpp_max_f32<0x140>(v);  // row_mirror: every lane has the max of its row of 16
  int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
  v = fmaxf(v, __int_as_float(t));
  t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2, 3
  v = fmaxf(v, __int_as_float(t));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ------------------------------------------------------------------------------------------------
// Gauss-Jordan with partial pivoting on the rows [Ta | Tb | t] of a chain, four wavefronts (see the header).  On exit the
// lane that owned pivot column `pc` holds its slice of row pc of Ta^-1 Tb in tb[] and (Ta^-1 t)[pc] in tt.
// ------------------------------------------------------------------------------------------------
template <bool WITH_TB, int NI>
__device__ __forceinline__ void gj_wide(double (&ta)[NI], double (&tb)[NI], double& tt, int& pc, const int lane, const int q,
                                        double (*sCol)[NP], int* sFound, double* sPiv, const double* touch_at, double (&touched)[2]) {
  pc = -1;
  double myrp = 1.0;
  static_for<0, NI>([&](auto kc) {
    constexpr int kk = decltype(kc)::value;
#pragma unroll 1
    for (int qo = 0; qo < 4; ++qo) {  // pivot column K = 4 kk + qo: register kk of wavefront qo
      const int buf = qo & 1;
      if constexpr (kk == NI - 4) {
        // the carry that follows reads Wq, Wp of this interface with scalar loads: bring their 512 cache lines into the L2 now
        // (two per thread, vector loads)
        if (touch_at != nullptr && qo == 0) {
          touched[0] = touch_at[threadIdx.x * 16];
          touched[1] = touch_at[(threadIdx.x + 256) * 16];
        }
      }
      if (q == qo) {
        const double mine = ta[kk];
        const float key = (pc < 0) ? fabsf((float)mine) : -1.0f;
        const float kmax = wave_max_key(key);
        const unsigned long long bal = __ballot(key == kmax);
        const int found_w = __ffsll((long long)bal) - 1;  // -1: a chain that has gone NaN
        sCol[buf][lane] = mine;
        if (lane == 0) {
          sFound[buf] = found_w;
          sPiv[buf] = readlane_f64(mine, found_w < 0 ? 0 : found_w);
        }
      }
      __syncthreads();
      const double mine = sCol[buf][lane];
      const int found = __builtin_amdgcn_readfirstlane(sFound[buf]);
      const int src = found < 0 ? 0 : found;
      const bool isp = lane == found;
      const double rp = fast_rcp(sPiv[buf]);
      const double f = isp ? 0.0 : mine * rp;
      if (isp) {
        pc = 4 * kk + qo;
        myrp = rp;
      }
#pragma unroll
      for (int i = kk; i < NI; ++i) ta[i] = fma(-f, readlane_f64(ta[i], src), ta[i]);  // (columns up to K are dead)
      if constexpr (WITH_TB) {
#pragma unroll
        for (int i = 0; i < NI; ++i) tb[i] = fma(-f, readlane_f64(tb[i], src), tb[i]);
      }
      tt = fma(-f, readlane_f64(tt, src), tt);
    }
  });
  if constexpr (WITH_TB) {
#pragma unroll
    for (int i = 0; i < NI; ++i) tb[i] *= myrp;
  }
  tt *= myrp;
}

// ------------------------------------------------------------------------------------------------
// Sweep kernel: per (c, m): forward carry recursion over the layers, bottom boundary, backward sweep.
// ------------------------------------------------------------------------------------------------
#ifndef RTD_WIDE_WG
#define RTD_WIDE_WG 3  /* workgroups (chains) per CU the sweep kernel is built for: 3 = 168 registers, 49 KB of LDS */
#endif
// NI: registers per lane and wavefront for each of Ta, Tb = a quarter of the columns that are carried: 16 (98 ... 128 streams) or 12
// (66 ... 96 streams, N <= 48: the padding streams 48 ... 63 decouple exactly; their rows are zero in every carried column, never
// pivot and keep their lanes, their columns are left out: 48 pivot steps of 25 elements instead of 64 of 33, twelve k-steps and three
// column tiles in the carry).
template <int NI>
__global__ __launch_bounds__(256, RTD_WIDE_WG) void rtd_sweep_wide_kernel(RtdDev d) {
  constexpr int NC = 4 * NI, NT = NC / 16;  // carried columns, 16-column tiles
  __shared__ double sS[NP * LDS_LD];   // S at its true row index
  __shared__ double sQ[NP * 17];       // bottom boundary: a quarter of the rows of Ba at a time
  __shared__ double sCol[2][NP];
  __shared__ int sFound[2];
  __shared__ double sPiv[2];
  __shared__ double sV[3][NP];
  __shared__ double sRed[2][4][NP];
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane;
  const long cm = blockIdx.x;
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const int L = d.L, Lm1 = L - 1;
  const double* Ym = d.Ym + cm * L * NP * NP;
  const double* Am = d.Am + cm * L * NP * NP;
  const double* kk = d.kk + cm * L * NP;
  const double* Ek = d.Ek + cm * L * NP;
  const double* Bv = d.Bv + cm * L * Q;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  const double* dq = d.dq + (long)c * L * d.Ns * Q;
  double* wsb = d.Fws + cm * Lm1 * W::SLOT;
  double* coef = d.coef + cm * L * Q;
  const double rTj = 1.0 / d.T[j];
  const int mg = d.m0 + d.mstep * m;
  const bool iso = d.Ns > 0 && mg == 0;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  // thermal particular solution of layer l at one of the layer's own boundaries (top / bottom), streams idx in [0, 2 NP): the values
  // the eigen kernel left in vb (it holds the polynomial coefficients about the layer's top, rtd_dd.h) -- no polynomial is evaluated here
  const double* vbp = d.vb + (long)c * L * 4 * NP;
  auto vedge = [&](int l, bool bottom, int idx) { return vbp[((long)l * 4 + (bottom ? 2 : 0)) * NP + idx]; };

  // carry rows Ta C- + Tb C+ = t, top boundary (:161-179, :284-285): Ta = Gm_0, Tb = Gp_0 E_0
  double ta[NI], tb[NI], tt;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = 4 * i + q;
    const double yv = Ym[j * NP + k], av = Am[j * NP + k] / kk[k];
    ta[i] = (yv + av) * rTj;
    tb[i] = (yv - av) * rTj * Ek[k];
  }
  tt = d.bneg[cm * NP + j];
  if (beam) tt -= Bv[NP + j];
  if (iso) tt -= dq[NP + j];

  int pc = -1;
  for (int l = 0; l < L; ++l) {
    double touched[2] = {0.0, 0.0};
    gj_wide<true, NI>(ta, tb, tt, pc, lane, q, sCol, sFound, sPiv, l < Lm1 ? wsb + (long)l * W::SLOT : nullptr, touched);
    if (pc < 0) pc = j;  // (a chain that has gone NaN finds no pivots: keep the stores inside the chain's own rows)
#pragma unroll
    for (int i = 0; i < NI; ++i) sS[pc * LDS_LD + 4 * i + q] = tb[i];
    if (l == Lm1) break;
    double* ws = wsb + (long)l * W::SLOT;
    kdouble* wk = as_k(ws);
#pragma unroll
    for (int i = 0; i < NI; ++i) ws[W::S + (4 * i + q) * NP + pc] = tb[i];  // S^T for the backward sweep
    if (q == 0) ws[W::SV + pc] = tt;
    // ---- carry: P = S Wq, R = S Wp on the matrix cores.  Wavefront q forms the columns [16 q, 16 q + 16) of both: the A operand
    // of a lane (k, m) is S[16 mt + m][4 ks + k] from LDS, the B operand of a lane (k, n) element [16 q + n][4 ks + k] of the stored
    // transposes -- all 32 operand loads of the wavefront in flight before the first MFMA.  The products then cross LDS (the S
    // area, free by then) into the rows-on-lanes layout of the elimination; from here on lane j holds row j.
    // (the lane's indices go through an opaque copy per layer: with loop-invariant indices the 64-bit addresses of the 64 loads
    //  below are hoisted out of the layer loop -- 64 registers' worth, spilled and reloaded every layer)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int k4 = lane_o >> 4, n16 = lane_o & 15, jo = lane_o;
    const bool tile_on = NI == 16 || q < NT;  // (NI = 12: the fourth wavefront has no column tile; it keeps the barriers)
    double bq[NI], bp[NI];
    if (tile_on) {
      const double* wqg = ws + W::WQ + (16 * q + n16) * NP + k4;
      const double* wpg = ws + W::WP + (16 * q + n16) * NP + k4;
#pragma unroll
      for (int ks = 0; ks < NI; ++ks) {
        bq[ks] = wqg[4 * ks];
        bp[ks] = wpg[4 * ks];
      }
    }
    {  // (S rho_b)[pc] over this wavefront's columns, and s, by true row index for the lanes that will hold those rows
      double part = 0.0;
#pragma unroll
      for (int i = 0; i < NI; ++i) part = fma(tb[i], wk[W::RB + 4 * i + q], part);
      sRed[0][q][pc] = part;
      if (q == 0) sV[0][pc] = tt;
    }
    __syncthreads();  // S (and the partial sums) complete
    v4d_t accP[4], accR[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) accP[mt] = accR[mt] = v4d_t{0.0, 0.0, 0.0, 0.0};
    if (tile_on) {
#pragma unroll
    for (int ks = 0; ks < NI; ++ks) {
      double a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = sS[(16 * mt + n16) * LDS_LD + 4 * ks + k4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        accP[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], bq[ks], accP[mt], 0, 0, 0);
        accR[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], bp[ks], accR[mt], 0, 0, 0);
      }
    }
    }
    const double Er = Ek[l * NP + j];
    const double srb = (sRed[0][0][j] + sRed[0][1][j]) + (sRed[0][2][j] + sRed[0][3][j]);
    const double tnew = ws[W::RT + j] - Er * (sV[0][j] - srb);
    kdouble* e1 = as_k(Ek + (l + 1) * NP);
    // the row j of Wp, Wq: this lane's own loads (coalesced), in flight across the exchange (the fence keeps the scheduler from
    // hoisting them over the MFMA chain, where there are no registers for them: it spilled all 32)
    RTD_FENCE();
    double wpr[NI], wqr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      wpr[i] = ws[W::WP + (4 * i + q) * NP + jo];
      wqr[i] = ws[W::WQ + (4 * i + q) * NP + jo];
    }
    __syncthreads();  // every wavefront has read its A operands: the S area is free
    // accumulator register r of tile mt, lane (kq, n): element [16 mt + 4 r + kq][16 q + n]
    if (tile_on) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[(16 * mt + 4 * r + k4) * LDS_LD + 16 * q + n16] = accP[mt][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) ta[i] = -(Er * sS[j * LDS_LD + 4 * i + q] + wpr[i]);  // Ta' = -(E S Wq + Wp)
    __syncthreads();
    if (tile_on) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[(16 * mt + 4 * r + k4) * LDS_LD + 16 * q + n16] = accR[mt][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) tb[i] = -(Er * sS[j * LDS_LD + 4 * i + q] + wqr[i]) * e1[4 * i + q];  // Tb' = -(E S Wp + Wq) E'
    tt = tnew;
    if (touched[0] == 1.2345e-300 && touched[1] == 1.2345e-300) tt += touched[0];  // (never: keeps the touching loads alive)
  }

  // ---- bottom boundary (up-streams at tau_L) (:208-232, :248-254, :288-293):  Ba C- + Bb C+ = br,
  //      with C- = s - S C+  ->  (Bb - Ba S) C+ = br - Ba s   (see rtd_sweep_kernel)
  double* v0 = sV[0];
  double* v1 = sV[1];
  double* v2 = sV[2];
  if (q == 0) v0[pc] = tt;  // s
  const double s_pc = tt;
  {
    const int l = Lm1;
    const double* ymL = Ym + (long)l * NP * NP;
    const double* amL = Am + (long)l * NP * NP;
    kdouble* ymk = as_k(ymL);
    kdouble* amk = as_k(amL);
    const double* kl = kk + (long)l * NP;
    const double att = beam ? exp(-ts0[L] / mu0) : 0.0;
    double pa[NI], qa[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      pa[i] = ymL[j * NP + 4 * i + q] * rTj;
      qa[i] = amL[j * NP + 4 * i + q] * rTj;
    }
    double br = d.bpos[cm * NP + j];
    if (mg < d.NBDRF) {
      const double delta = (mg == 0) ? 2.0 : 1.0;
      const double* qt = d.bdrfq + (((long)c * d.NBDRF + mg) * NP + j) * NP;
      double rbm = 0.0, rvm = 0.0;
      for (int j2 = 0; j2 < NP; ++j2) {
        const double Rij = delta * qt[j2] * d.mu[j2] * d.w[j2] / d.T[j2];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          pa[i] -= Rij * ymk[j2 * NP + 4 * i + q];
          qa[i] += Rij * amk[j2 * NP + 4 * i + q];
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
    // am = Bb - Ba S (into ta: this wavefront's columns) and bvec = br - Ba s.  A lane needs its whole row of Ba, of which every
    // wavefront made a quarter of the columns: the rows cross in LDS sixteen columns at a time (8.5 KB, not the 33 KB of a full
    // copy: three chains fit a CU).
    double ba[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = 4 * i + q;
      const double qk = qa[i] / kl[k];
      ba[i] = (pa[i] - qk) * Ek[l * NP + k];  // Ba (with the scaling of C-)
      ta[i] = pa[i] + qk;                      // Bb
    }
    double bvec = br;
    static_for<0, NT>([&](auto kqc) {
      constexpr int kq = decltype(kqc)::value;
      __syncthreads();  // (first pass: S and s complete; later ones: the previous quarter has been read)
#pragma unroll
      for (int t = 0; t < 4; ++t) sQ[j * 17 + 4 * t + q] = ba[4 * kq + t];  // columns 16 kq + 4 t + q
      __syncthreads();
      double bq[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) bq[k] = sQ[j * 17 + k];
#pragma unroll
      for (int k = 0; k < 16; ++k) bvec = fma(-bq[k], v0[16 * kq + k], bvec);
#pragma unroll 4
      for (int i = 0; i < NI; ++i) {
        const int cc = 4 * i + q;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          a0 = fma(bq[k], sS[(16 * kq + k) * LDS_LD + cc], a0);
          a1 = fma(bq[k + 1], sS[(16 * kq + k + 1) * LDS_LD + cc], a1);
        }
        ta[i] -= a0 + a1;
      }
    });
    int pc2 = -1;
    double touched[2] = {0.0, 0.0};
    gj_wide<false, NI>(ta, tb, bvec, pc2, lane, q, sCol, sFound, sPiv, nullptr, touched);
    if (pc2 < 0) pc2 = j;
    if (q == 0) v1[pc2] = bvec;  // C+
    __syncthreads();
    if (q == 0) {
      double cmin = s_pc;  // C-[pc] = s[pc] - S[pc][:] C+
#pragma unroll 8
      for (int k = 0; k < NC; ++k) cmin = fma(-sS[pc * LDS_LD + k], v1[k], cmin);  // (the columns that are carried)
      v2[pc] = cmin;
    }
    __syncthreads();
    if (q == 0) {
      coef[(long)l * Q + j] = v2[j];
      coef[(long)l * Q + NP + j] = v1[j];
      // singular system (the reference's solve_banded / solve raises LinAlgError, :326-333, :383)
      if (!(fabs(v2[j]) + fabs(v1[j]) < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
    }
  }
  // ---- backward sweep: C+_l = Wq C-' + Wp E' C+' + rho_b ;  C-_l = s - S C+_l.  Every wavefront keeps both vectors
  //      lane-wise; wavefront q sums over the columns [16 q, 16 q + 16) (readlane broadcasts), partial sums meet in LDS.
  double cmj = v2[j], cpj = v1[j];
  const int k0 = 16 * q;
  for (int l = Lm1 - 1; l >= 0; --l) {
    const double* ws = wsb + (long)l * W::SLOT;
    const bool cols_on = NI == 16 || q < NT;  // (NI = 12: the columns 48 ... 63 are not carried -- nothing was stored for them)
    double wq[16], wp[16], st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      wq[i] = cols_on ? ws[W::WQ + (k0 + i) * NP + j] : 0.0;
      wp[i] = cols_on ? ws[W::WP + (k0 + i) * NP + j] : 0.0;
      st[i] = cols_on ? ws[W::S + (k0 + i) * NP + j] : 0.0;
    }
    const double rb = ws[W::RB + j], sv = ws[W::SV + j];
    const double ecp = Ek[(l + 1) * NP + j] * cpj;  // E'_j C+'_j
    double part = 0.0;
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // (k0 + i is wave-uniform but not a constant: v_readlane with an SGPR lane index)
      part = fma(wq[i], readlane_f64(cmj, k0 + i), part);
      part = fma(wp[i], readlane_f64(ecp, k0 + i), part);
    });
    // (one buffer per reduction is enough: a wavefront writes a buffer again only after a barrier that every reader of its
    //  previous contents has reached)
    sRed[0][q][j] = part;
    __syncthreads();
    const double cp = rb + ((sRed[0][0][j] + sRed[0][1][j]) + (sRed[0][2][j] + sRed[0][3][j]));
    double part2 = 0.0;
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      part2 = fma(st[i], readlane_f64(cp, k0 + i), part2);
    });
    sRed[1][q][j] = part2;
    __syncthreads();
    const double cmin = sv - ((sRed[1][0][j] + sRed[1][1][j]) + (sRed[1][2][j] + sRed[1][3][j]));
    cmj = cmin;
    cpj = cp;
    if (q == 0) {
      coef[(long)l * Q + j] = cmin;
      coef[(long)l * Q + NP + j] = cp;
    }
  }
}

}  // namespace

void rtd_launch_bc_wide(const RtdDev& d, hipStream_t s, int part) {
  const long nif = (long)d.C * d.M * (d.L - 1);
  if (part == 0 && nif > 0) {
    if (d.N <= 48) hipLaunchKernelGGL(rtd_iface_mfma_kernel<48>, dim3((unsigned)((2 * nif + 7) / 8 * 8)), dim3(64), 0, s, d);
    else hipLaunchKernelGGL(rtd_iface_mfma_kernel<64>, dim3((unsigned)((2 * nif + 7) / 8 * 8)), dim3(64), 0, s, d);
  }
  if (part == 1) {
    // 66 ... 96 streams: the instance that leaves the padding columns 48 ... 63 out (the interface kernel makes the same cut)
    if (d.N <= 48) hipLaunchKernelGGL(rtd_sweep_wide_kernel<12>, dim3((unsigned)((long)d.C * d.M)), dim3(256), 0, s, d);
    else hipLaunchKernelGGL(rtd_sweep_wide_kernel<16>, dim3((unsigned)((long)d.C * d.M)), dim3(256), 0, s, d);
  }
}
