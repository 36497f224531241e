// rtd_bc_tile2.hip -- the 64-stream boundary-condition kernel in its LEAN form (round 4): two wavefronts per SIMD.
//
// Replaces _solve_for_coeffs (src/PythonicDISORT/_solve_for_coeffs.py:8-390) and the interface values of the closures
// (_assemble_intensity_and_fluxes.py:221-254) for 32 < NQuad <= 64; same recursion, same speculative elimination and the same
// matrix-core layout as rtd_bc_tile_kernel<2> (rtd_bc.hip, whose header comment has the mathematics).  What is different:
//
//   rtd_bc_tile_kernel<2> keeps six operand matrices in registers (layers l, l + 1 and the prefetch of l + 2: 192 of its 472
//   VGPRs), the inputs of the running elimination in a 17 KB LDS save area (read back only when the speculation fails) and the
//   chain's small vectors in an 18 KB LDS window: ONE wavefront per SIMD, whose dependency stalls nothing hides (VALU issue
//   30 % + matrix pipe 22 % of the SIMD cycles, profiles/archive/r03_pmc_traffic_cfg5.json), and no room for an eigen-stage wavefront
//   of the next window beside it.  Here no operand matrix lives across an elimination in more than 64 registers, and a failed
//   speculation RECOMPUTES its inputs -- from the H, s of the layer above, which the forward sweep stores anyway -- and eliminates
//   them again with column pivoting IN REGISTERS (GjPivT): no save area.  <= 256 VGPRs and 20 KB of LDS: two chains per SIMD
//   hide each other's stalls, and the chains that are pivoted throughout (near-conservative mode 0: chain_needs_pivoting) cost
//   1.3 x instead of 10 x.
//   End of round 5 (profiles/r05_bc_tile2_phases.txt): what the carry across an interface reads -- Y, A of the layer below and the
//   interface's vectors -- arrives by LDS-DMA (global_load_lds_dwordx4: no registers, no wait) while the elimination above runs;
//   the carry's only wait finds requests that are an elimination old, H and s are stored BEHIND it (loads, stores and the DMA share
//   one in-order counter: the ~80 vector loads that used to follow the stores each waited for a store's acknowledgement), and the
//   next carry takes its own layer's Y, A from the same LDS images: every matrix is fetched once per forward sweep.  The backward
//   sweep requests a step ahead in the same way.
//
// rtd_bc_tile_kernel<2> stays selectable (RTD_BC_TILE_V1=1: A/B runs, and the suite passes under it).
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

#include "rtd_bc_common.h"
#include "rtd_bc_tile_common.h"

constexpr int T2 = 2, NP2 = 32, Q2 = 64, NN2 = 1024;

// Diagnostic build (-DRTD_T2_STAMPS, never shipped; tools/build_variant.py): s_memtime at the phase boundaries of a chain, summed
// over its layers, printed by a few wavefronts (T2STAMP lines; tools/eig_phase_cycles.py formats them too).
#ifdef RTD_T2_STAMPS
#define RTD_T2STAMP(k)                                              \
  {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                              \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
    t2acc[k] += now_ - t2last;                                      \
    t2last = now_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                              \
  }
#else
#define RTD_T2STAMP(k)
#endif
using Mat = MatT<2>;
using Row = RowT<2>;
using Col = ColT<2>;

// Column-pivoted Gauss-Jordan on the stacked rows [Ta^T ; Tb^T (WITH_TB) ; t^T] in registers, 2 x 2 tiles in the D layout:
// step K makes row K of Ta^T a unit vector; the pivot is the largest unused column of that row with threshold 1/4 in favour
// of the diagonal (the rule of the LDS redo of rtd_bc_tile_kernel and of GjPiv at 32 streams).  One chain per wavefront:
// the pivot column is wave-uniform -- its row entry by v_readlane, its column by ds_bpermute; the pivot column is scaled by
// 1 / pivot exactly.  Afterwards the column that was the pivot of step c holds unknown c (sPerm[c]): unpermute() moves them back.
template <bool WITH_TB, int K>
struct GjPivT {
  static __device__ __forceinline__ void run(Mat& ta, Mat& tb, Col& tv, unsigned& used, int* sPerm, const int col, const int rowbase,
                                             const int lane) {
    constexpr int KI = K >> 4, K16 = K & 15, QK = K16 >> 2, RK = K16 & 3;
    double x[2];
    float key[2];
#pragma unroll
    for (int J = 0; J < 2; ++J) {
      x[J] = bcast_row<RK>(ta.t[KI][J][QK], col);  // row K of Ta^T, replicated over the lane-rows
      key[J] = ((used >> (16 * J + col)) & 1u) ? -1.0f : fabsf((float)x[J]);
    }
    const float kmax = group_max_key<16>(fmaxf(key[0], key[1]));
    const float kd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(key[KI]), 0x150 + K16, 0xF, 0xF, true));
    int pcol = K;
    if (!(kd >= 0.25f * kmax && kd > 0.0f)) {  // (wave-uniform: the four lane-rows hold the same row)
      const unsigned long long b0 = __ballot(key[0] == kmax) & 0xFFFFull, b1 = __ballot(key[1] == kmax) & 0xFFFFull;
      pcol = b0 ? __ffsll((long long)b0) - 1 : (b1 ? 16 + __ffsll((long long)b1) - 1 : K);  // (NaN chain: no candidate, NaN stays NaN)
    }
    pcol = __builtin_amdgcn_readfirstlane(pcol);
    const int PJ = pcol >> 4, p16 = pcol & 15;
    const double xp = readlane_f64(PJ ? x[1] : x[0], p16);
    const double rp = fast_rcp(xp);
    const double f0 = x[0] * rp, f1 = x[1] * rp;
    const bool isp0 = PJ == 0 && col == p16, isp1 = PJ == 1 && col == p16;
    const int addr = (rowbase | p16) << 2;
    auto upd = [&](double& v0, double& v1) {  // the two tile columns of one register row; source: the pivot column's entry of that row
      const double bp = bperm(addr, PJ ? v1 : v0);
      v0 = isp0 ? bp * rp : fma(-f0, bp, v0);
      v1 = isp1 ? bp * rp : fma(-f1, bp, v1);
    };
    static_for<KI, 2>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      static_for<(I == KI ? QK : 0), 4>([&](auto qc) {  // rows above are finished: unit vectors with a zero in every unused column
        constexpr int q = decltype(qc)::value;
        double v0 = ta.t[I][0][q], v1 = ta.t[I][1][q];
        upd(v0, v1);
        ta.t[I][0][q] = v0;
        ta.t[I][1][q] = v1;
      });
    });
    if constexpr (WITH_TB) {
      static_for<0, 2>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        static_for<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          double v0 = tb.t[I][0][q], v1 = tb.t[I][1][q];
          upd(v0, v1);
          tb.t[I][0][q] = v0;
          tb.t[I][1][q] = v1;
        });
      });
    }
    upd(tv.c[0], tv.c[1]);
    used |= 1u << pcol;
    if (lane == 0) sPerm[K] = pcol;
    if constexpr (K + 1 < 32) GjPivT<WITH_TB, K + 1>::run(ta, tb, tv, used, sPerm, col, rowbase, lane);
  }
};

__global__ __launch_bounds__(64, 2) void rtd_bc_tile2_kernel(RtdDev d, int* need_split) {
  constexpr int T = T2, NP = NP2, Q = Q2, NN = NN2;
  const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15, rowbase = lane & 48;
  const long cm = chain_of_block(blockIdx.x, d.C, d.M);
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const int L = d.L, Lm1 = L - 1;
#ifdef RTD_T2_STAMPS
  long long t2acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long t2last = (long long)__builtin_amdgcn_s_memtime();
  const long long t2start = t2last;
#endif
  const double* Ym = d.Ym + cm * L * NN;
  const double* Am = d.Am + cm * L * NN;
  const double* kk = d.kk + cm * L * NP;
  const double* Ek = d.Ek + cm * L * NP;
  const double* Bv = d.Bv + cm * L * Q;
  const double* att = d.att + (long)c * (L + 1);    // exp(-tau*_t / mu0) at the interfaces (beam only)
  const double* vbp = d.vb + (long)c * L * 4 * NP;  // thermal solution v_l at its layer's top (up, down) and bottom (up, down); mode 0
  double* wsb = d.Fws + cm * Lm1 * Ws<NP>::SLOT;
  double* coef = d.coef + cm * L * Q;
  const int mg = d.m0 + d.mstep * m;
  const bool iso = d.Ns > 0 && mg == 0;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  // pivoted throughout (in registers here: the rule of the 32-stream kernel applies, every mode-0 chain with a small eigenvalue),
  // or RTD_BC_FORCE_PIVOT (either value)
  const int careful = chain_needs_pivoting(d, RTD_BC_CAREFUL_ALL_MODE0 ? mg == 0 : iso, kk, L, NP) | (d.flags & 1) | ((d.flags >> 2) & 1);
  if ((d.flags & 2) && m % 3 == 0) {  // test hook (RTD_BC_FORCE_HANDOVER): every third Fourier mode's chain goes to the pivoted
    //                                    row-per-lane kernels (by mode, not by chain index: the choice must not depend on the windowing)
    if (lane == 0) {
      need_split[cm] = 1;
      *d.split_any = 1;
    }
    return;
  }
  constexpr int LDM = NP + 1;
  // forward sweep: Y and A of ONE layer, row-major, filled by LDS-DMA (global_load_lds_dwordx4: no registers, no wait) while the
  // elimination of the layer above runs -- every matrix of the hand-off is fetched from memory ONCE in the forward sweep and read
  // from here twice (as the lower layer of one interface, then as the upper layer of the next); backward sweep: staging area
  __shared__ double sBuf[2 * NN];
  // ... and the vectors of the interface below it, by the same route: [B_l, B_(l+1) | v_l bottom, v_(l+1) top | k_l, k_(l+1), E_l, E_(l+1)]
  __shared__ double sVec[4 * Q + 2 * Q];
  constexpr int VB = 0, VV = 2 * Q, VK = 4 * Q, VE = 4 * Q + 2 * NP;
  // (kq, col are passed in: the callers hand over an opaque copy of the lane index, so that the compiler rebuilds the few
  //  address registers where they are needed instead of keeping dozens of hoisted ones alive)
  auto load_d = [](const double* p, const int kq, const int col) {  // row-major NP x NP matrix -> tiles in the D layout
    Mat x;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) x.t[I][J][q] = p[(16 * I + 4 * q + kq) * NP + 16 * J + col];
    return x;
  };
  auto lds_d = [&](const int which, const int kq, const int col) {  // the same tiles from the layer buffers (0: Y, 1: A)
    Mat x;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) x.t[I][J][q] = sBuf[which * NN + (16 * I + 4 * q + kq) * NP + 16 * J + col];
    return x;
  };
  // Y and A of `layer` into the layer buffers: 16 wave-instructions of 1 KB (lane: 16 bytes), in flight behind whatever follows
  auto dma_matrices = [&](const int layer, const int lv) {
    const double* gy = Ym + (long)layer * NN + 2 * lv;
    const double* ga = Am + (long)layer * NN + 2 * lv;
#pragma unroll
    for (int i = 0; i < NN / 128; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gy + 128 * i),
                                       (__attribute__((address_space(3))) void*)(sBuf + 128 * i), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + 128 * i),
                                       (__attribute__((address_space(3))) void*)(sBuf + NN + 128 * i), 16, 0, 0);
    }
  };
  // forward sweep: the layer below interface layer - 1, and that interface's vectors (three more instructions)
  auto prefetch_layer = [&](const int layer, const int lv) {
    const int l = layer - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bv + l * Q + 2 * lv),
                                     (__attribute__((address_space(3))) void*)(sVec + VB), 16, 0, 0);
    if (iso)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbp + (l * 4 + 2) * NP + 2 * lv),
                                       (__attribute__((address_space(3))) void*)(sVec + VV), 16, 0, 0);
    const double* ke = (lv < 32 ? kk + l * NP : Ek + l * NP - 2 * NP) + 2 * lv;  // lanes 0-31: k_l, k_(l+1); lanes 32-63: E_l, E_(l+1)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ke,
                                     (__attribute__((address_space(3))) void*)(sVec + VK), 16, 0, 0);
    dma_matrices(layer, lv);
  };
  auto vec_row = [&](const int off, const int kq) {  // a vector of the interface from sVec, row form
    Row x;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) x.r[I][q] = sVec[off + 16 * I + 4 * q + kq];
    return x;
  };
  auto vec_col = [&](const int off, const int col) {
    Col x;
#pragma unroll
    for (int J = 0; J < T; ++J) x.c[J] = sVec[off + 16 * J + col];
    return x;
  };
  auto load_row = [](const double* p, const int kq) {
    Row x;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) x.r[I][q] = p[16 * I + 4 * q + kq];
    return x;
  };
  auto load_col = [](const double* p, const int col) {
    Col x;
#pragma unroll
    for (int J = 0; J < T; ++J) x.c[J] = p[16 * J + col];
    return x;
  };
  auto make_eye = [](const int kq, const int col) {
    Mat e;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) e.t[I][J][q] = (I == J && 4 * q + kq == col) ? 1.0 : 0.0;
    return e;
  };
  auto opaque_lane = [&]() {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    return lv;
  };
  auto fail_chain = [&]() {  // this chain could not be solved here: hand it to the row-per-lane kernels
    if (lane == 0) {
      need_split[cm] = 1;
      *d.split_any = 1;  // (they do not evaluate at the interfaces: the evaluation kernel then does it for the window)
    }
  };
  __shared__ int sPerm[NP];
  __shared__ double sT[2][NP];  // T and 1 / T
  for (int e = lane; e < NP; e += 64) {
    const double t = d.T[e];
    sT[0][e] = t;
    sT[1][e] = fast_rcp(t);
  }
  __syncthreads();

  // after GjPivT: unknown (J, col) sits in the column that was the pivot of step 16 J + col
  auto unpermute = [&](Mat& xb, Col& xv, const bool with_tb, const int rowbase, const int col) {
    __syncthreads();
    int src[2];
#pragma unroll
    for (int J = 0; J < T; ++J) src[J] = sPerm[16 * J + col];
    __syncthreads();
    double nv[2];
#pragma unroll
    for (int J = 0; J < T; ++J) {
      const int addr = (rowbase | (src[J] & 15)) << 2;
      const double b0 = bperm(addr, xv.c[0]), b1 = bperm(addr, xv.c[1]);
      nv[J] = (src[J] >> 4) ? b1 : b0;
    }
    xv.c[0] = nv[0];
    xv.c[1] = nv[1];
    if (with_tb) {
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double o[2];
#pragma unroll
          for (int J = 0; J < T; ++J) {
            const int addr = (rowbase | (src[J] & 15)) << 2;
            const double b0 = bperm(addr, xb.t[I][0][q]), b1 = bperm(addr, xb.t[I][1][q]);
            o[J] = (src[J] >> 4) ? b1 : b0;
          }
          xb.t[I][0][q] = o[0];
          xb.t[I][1][q] = o[1];
        }
    }
  };
  auto all_finite = [&](const Mat& xb, const Col& xv, const bool with_tb) {
    double chk = 0.0;
#pragma unroll
    for (int J = 0; J < T; ++J) {
      chk += fabs(xv.c[J]);
      if (with_tb)
#pragma unroll
        for (int I = 0; I < T; ++I) chk += fabs(xb.t[I][J][0]) + fabs(xb.t[I][J][1]) + fabs(xb.t[I][J][2]) + fabs(xb.t[I][J][3]);
    }
    return chk < 1e300;
  };

  // ---- the producers of the elimination's inputs.  Each can be called again when a speculation fails.
  // top boundary (down-streams at tau = 0) (:161-179, :284-285):  Ta = Gm_0 = (Y + A/k)/T-rows,  Tb = Gp_0 E_0, transposed
  auto top_rows = [&](Mat& ta, Mat& tb, Col& tv) {
    const int lv = opaque_lane(), kq = lv >> 4, col = lv & 15;
    const Mat a0 = load_d(Am, kq, col), y0 = load_d(Ym, kq, col);
    const Row k_row = load_row(kk, kq), e_row = load_row(Ek, kq);
    const Col tc = load_col(&sT[1][0], col);
    tv = load_col(d.bneg + cm * NP, col);
    if (beam) {
      const Col b = load_col(Bv + NP, col);  // (tau = 0: attenuation 1)
#pragma unroll
      for (int J = 0; J < T; ++J) tv.c[J] -= b.c[J];
    }
    if (iso) {
      const Col b = load_col(vbp + 1 * NP, col);  // v_0 at the top of layer 0, down-streams
#pragma unroll
      for (int J = 0; J < T; ++J) tv.c[J] -= b.c[J];
    }
    const Mat eye = make_eye(kq, col);
    const Mat yt = mmT<T>(y0, eye), at = mmT<T>(a0, eye);
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double av = at.t[I][J][q] * fast_rcp(k_row.r[I][q]);
          ta.t[I][J][q] = (yt.t[I][J][q] + av) * tc.c[J];
          tb.t[I][J][q] = (yt.t[I][J][q] - av) * tc.c[J] * e_row.r[I][q];
        }
  };
  // carry across interface l: in  tb = H_l = S_l^T, tv = s_l;  out  [Ta'^T ; Tb'^T ; t'^T] of layer l + 1:
  //   Ta'^T = -(Wq^T H E + Wp^T),  Tb'^T = -E' (Wp^T H E + Wq^T),  t' = rho_t - E (s - S rho_b),
  //   Wp/Wq = (M1 +- M2s)/2,  M1 = A_l^T Y',  M2s = diag(k) Y_l^T A' diag(1/k').  Every operand is requested where it is used.
  //   (a0, y0s: A_l, Y_l -- requested by the caller ahead of the elimination that precedes this call, unless `have` is false)
  auto carry = [&](const int l, Mat& ta, Mat& tb, Col& tv, Mat& a0, Mat& y0s, const bool have, const bool below_in_lds) {
    const int lv = opaque_lane(), kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
    double* ws = wsb + (long)l * Ws<NP>::SLOT;
    if (!have) {
      a0 = load_d(Am + (long)l * NN, kq, col);
      y0s = load_d(Ym + (long)l * NN, kq, col);
    }
    // Everything this call reads of the layer below and of the interface is in LDS since the elimination above (below_in_lds; a
    // repeated carry asks memory instead).  Loads, stores and the DMA share one in-order counter: the wait here finds only requests
    // that are an elimination old, and H_l, s_l leave AFTER it -- the call has no request of its own behind them, so nothing in
    // the forward sweep ever waits for a store's acknowledgement (which cost 12 k cycles per layer when the stores went first).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (have) {  // (a repeated carry has reloaded them from there)
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J)
#pragma unroll
          for (int q = 0; q < 4; ++q) ws[Ws<NP>::S + (16 * I + 4 * q + kq) * NP + 16 * J + col] = tb.t[I][J][q];
      if (kq == 0)
#pragma unroll
        for (int J = 0; J < T; ++J) ws[Ws<NP>::SV + 16 * J + col] = tv.c[J];
    }
    // the jump of the particular solution at the interface, rows: r_l = p_(l+1)(tau_(l+1)) - p_l(tau_(l+1)) (:184-205, :242-245)
    Row vs, vd;  // T (r_up + r_dn), -T (r_up - r_dn)
    {
      Row ru, rd;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) ru.r[I][q] = rd.r[I][q] = 0.0;
      if (beam) {
        const double a = att[l + 1];
        const Row bu1 = below_in_lds ? vec_row(VB + Q, kq) : load_row(Bv + (l + 1) * Q, kq);
        const Row bu0 = below_in_lds ? vec_row(VB, kq) : load_row(Bv + l * Q, kq);
        const Row bd1 = below_in_lds ? vec_row(VB + Q + NP, kq) : load_row(Bv + (l + 1) * Q + NP, kq);
        const Row bd0 = below_in_lds ? vec_row(VB + NP, kq) : load_row(Bv + l * Q + NP, kq);
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ru.r[I][q] = (bu1.r[I][q] - bu0.r[I][q]) * a;
            rd.r[I][q] = (bd1.r[I][q] - bd0.r[I][q]) * a;
          }
      }
      if (iso) {  // v_(l+1) at its top minus v_l at its bottom
        const Row tu = below_in_lds ? vec_row(VV + 2 * NP, kq) : load_row(vbp + ((l + 1) * 4 + 0) * NP, kq);
        const Row td = below_in_lds ? vec_row(VV + 3 * NP, kq) : load_row(vbp + ((l + 1) * 4 + 1) * NP, kq);
        const Row bu = below_in_lds ? vec_row(VV, kq) : load_row(vbp + (l * 4 + 2) * NP, kq);
        const Row bd = below_in_lds ? vec_row(VV + NP, kq) : load_row(vbp + (l * 4 + 3) * NP, kq);
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ru.r[I][q] += tu.r[I][q] - bu.r[I][q];
            rd.r[I][q] += td.r[I][q] - bd.r[I][q];
          }
      }
      const Row t_row = load_row(&sT[0][0], kq);
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vs.r[I][q] = t_row.r[I][q] * (ru.r[I][q] + rd.r[I][q]);
          vd.r[I][q] = -t_row.r[I][q] * (ru.r[I][q] - rd.r[I][q]);
        }
    }
    {
      const Col k0c = below_in_lds ? vec_col(VK, col) : load_col(kk + l * NP, col);
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int q = 0; q < 4; ++q) y0s.t[I][J][q] *= k0c.c[J];
    }
    // rho = G_l^-1 r_l:  rho_t/b = 1/4 [V^-1 (r_up + r_dn) +- U^-1 (r_up - r_dn)],  V^-1[j][i] = T_i A[i][j],  U^-1[j][i] = -k_j T_i Y[i][j]
    Col rt, rb;
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
    RTD_T2STAMP(6)  // carry: vectors of the interface, rho
    const Col e0c = below_in_lds ? vec_col(VE, col) : load_col(Ek + l * NP, col);
    Col tnew;
    {
      const Col srb = col_dotT<T>(tb, col_to_rowT<T>(rb, rowbase, kq));
#pragma unroll
      for (int J = 0; J < T; ++J) tnew.c[J] = rt.c[J] - e0c.c[J] * (tv.c[J] - srb.c[J]);
    }
#pragma unroll
    for (int I = 0; I < T; ++I)  // H E, in place (H itself has been stored)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) tb.t[I][J][q] *= e0c.c[J];
    // X + M1^T = M1^T (H E + I) and Z - M2s^T = M2s^T (H E - I): the unit matrix goes onto the diagonal of H E before each product
    // (eight selects and adds), so that neither M1^T nor M2s^T is ever formed -- no transposed copy through LDS (two round trips and
    // six barriers per layer), and the 8 KB of LDS they went through are free in the forward sweep
    auto add_to_diagonal = [&](Mat& x, const double v) {
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) x.t[I][I][q] += (4 * q + kq == col) ? v : 0.0;
    };
    RTD_T2STAMP(7)  // carry: t', H E
    Mat s1;  // M1^T (H E + I)
    {
      const Mat y1 = below_in_lds ? lds_d(0, kq, col) : load_d(Ym + (long)(l + 1) * NN, kq, col);
      const Mat m1 = mmT<T>(a0, y1);
      add_to_diagonal(tb, 1.0);
      s1 = mmT<T>(m1, tb);
    }
    RTD_T2STAMP(8)  // carry: M1, M1^T (H E + I)
    Mat dd;  // M2s^T (H E - I)
    {
      Mat a1s = below_in_lds ? lds_d(1, kq, col) : load_d(Am + (long)(l + 1) * NN, kq, col);
      const Col k1c = below_in_lds ? vec_col(VK + NP, col) : load_col(kk + (l + 1) * NP, col);
#pragma unroll
      for (int J = 0; J < T; ++J) {
        const double rk1 = fast_rcp(k1c.c[J]);
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int q = 0; q < 4; ++q) a1s.t[I][J][q] *= rk1;
      }
      const Mat m2s = mmT<T>(y0s, a1s);
      add_to_diagonal(tb, -2.0);
      dd = mmT<T>(m2s, tb);
    }
    RTD_T2STAMP(9)  // carry: M2s, M2s^T (H E - I)
    const Row e1r = below_in_lds ? vec_row(VE + NP, kq) : load_row(Ek + (l + 1) * NP, kq);
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ta.t[I][J][q] = -0.5 * (s1.t[I][J][q] - dd.t[I][J][q]);
          tb.t[I][J][q] = -0.5 * (s1.t[I][J][q] + dd.t[I][J][q]) * e1r.r[I][q];
        }
    tv = tnew;
    // (stored last: a store ahead of the operand reads above would put its acknowledgement in front of them)
    if (kq == 0)
#pragma unroll
      for (int J = 0; J < T; ++J) ws[Ws<NP>::RB + 16 * J + col] = rb.c[J];
  };

  // One call site for each producer (they are large: instruction cache): the loop asks for the inputs of layer l's elimination,
  // eliminates speculatively, and on a failed speculation (1 of ~400 000 eliminations on cfg5) asks the SAME producer again --
  // after reloading the H, s it starts from, which the forward sweep has stored -- and eliminates with pivoting.
  Mat ta, tb;
  Col tv;
  {
    int l = 0;
    int produce = -1;  // -1: the top boundary rows; >= 0: the carry across that interface (tb, tv hold H, s of the layer above it)
    bool pivot_now = careful != 0;
    Mat pa, py;             // A_l, Y_l for the carry across interface l: in registers BEFORE the elimination of layer l (its 68 registers
    bool have_pre = false;  // leave room) -- from the layer buffers, where the carry across interface l - 1 has left them
    int in_lds = -1;        // the layer whose Y, A the layer buffers hold (or are being filled with)
    for (;;) {
      if (produce < 0) top_rows(ta, tb, tv);
      else carry(produce, ta, tb, tv, pa, py, have_pre, in_lds == produce + 1);
      RTD_T2STAMP(0)  // producer: top rows / carry
      const int lv = opaque_lane(), kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
      if (l < Lm1) {  // (wave-uniform) the operands of the carry across interface l, which follows this elimination
        if (in_lds == l) {
          pa = lds_d(1, kq, col);
          py = lds_d(0, kq, col);
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): they have left the buffers ...
        } else {  // first layer, or a repeated carry
          pa = load_d(Am + (long)l * NN, kq, col);
          py = load_d(Ym + (long)l * NN, kq, col);
        }
        have_pre = true;
        if (in_lds != l + 1) {  // ... which take the layer below while the elimination runs
          prefetch_layer(l + 1, lv);
          in_lds = l + 1;
        }
      }
      RTD_T2STAMP(1)  // operands into registers, DMA issued
      // ---- elimination: [Ta^T ; Tb^T ; t^T] -> H = S^T (in tb), s (in tv)
      if (!pivot_now) {
        int bad = 0;
        GjFastT<T, 0>::run(ta, tb, tv, bad, col);
        bad |= all_finite(tb, tv, true) ? 0 : 1;  // zero pivot: inf / nan
        if (__any(bad)) {
          if (produce >= 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // (the stores of H, s of the layer above have left)
            const double* wsp = wsb + (long)produce * Ws<NP>::SLOT;
            tb = load_d(wsp + Ws<NP>::S, kq, col);
            tv = load_col(wsp + Ws<NP>::SV, col);
          }
          pivot_now = true;
          have_pre = false;  // (pa, py are layer l's: the producer asks for its own)
          continue;
        }
      } else {
        unsigned used = 0;
        GjPivT<true, 0>::run(ta, tb, tv, used, sPerm, col, rowbase, lane);
        unpermute(tb, tv, true, rowbase, col);
        if (__any(!all_finite(tb, tv, true))) {  // singular carry block: the row-per-lane kernels (partial pivoting) take the chain
          fail_chain();
          return;
        }
      }
      pivot_now = careful != 0;
      RTD_T2STAMP(2)  // elimination
      if (l == Lm1) break;
      produce = l;
      ++l;
      RTD_T2STAMP(3)  // wait for the requests + stores of H, s
    }
  }

  // ---- bottom boundary (up-streams at tau_L) (:208-232, :248-254, :288-293):  Ba C- + Bb C+ = br,
  //      with C- = s - S C+  ->  (Bb - Ba S) C+ = br - Ba s;  Ba = [(I - R) P0 - (I + R) Q0] E_L, Bb = (I - R) P0 + (I + R) Q0,
  //      P0 = Y/T-rows, Q0 = A/(k T-rows), R = (1 + delta_m0) q (mu w).  Solved transposed like the carry.
  Col cminus, cplus;
  {
    const int l = Lm1;
    const double attL = beam ? att[L] : 0.0;
    auto bottom_rows = [&](Mat& mt, Col& rhs) {  // (Bb - Ba S)^T and br - Ba s from H = tb, s = tv
      const int lv = opaque_lane(), kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
      const Row eLr = load_row(Ek + l * NP, kq), t_row = load_row(&sT[0][0], kq);
      const Col kLc = load_col(kk + l * NP, col);
      const Mat eye = make_eye(kq, col);
      Mat x1 = eye, x2 = eye, rtr;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = 0; J < T; ++J) rtr.t[I][J] = v4f64{0.0, 0.0, 0.0, 0.0};
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
      Mat bat;
      {
        Mat p0 = load_d(Ym + (long)l * NN, kq, col);
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int J = 0; J < T; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) p0.t[I][J][q] *= fast_rcp(t_row.r[I][q]);
        const Mat g1 = mmT<T>(p0, x1);  // ((I - R) P0)^T
        Mat q0 = load_d(Am + (long)l * NN, kq, col);
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int J = 0; J < T; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) q0.t[I][J][q] *= fast_rcp(t_row.r[I][q]) * fast_rcp(kLc.c[J]);
        const Mat g2 = mmT<T>(q0, x2);  // ((I + R) Q0)^T
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int J = 0; J < T; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              bat.t[I][J][q] = eLr.r[I][q] * (g1.t[I][J][q] - g2.t[I][J][q]);
              mt.t[I][J][q] = g1.t[I][J][q] + g2.t[I][J][q];
            }
      }
      {
        const Mat sd = mmT<T>(tb, eye);   // S in the D layout
        const Mat hb = mmT<T>(sd, bat);   // S^T Ba^T
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
          for (int J = 0; J < T; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) mt.t[I][J][q] -= hb.t[I][J][q];  // (Bb - Ba S)^T
      }
      Col br = load_col(d.bpos + cm * NP, col);
      const Col bu = beam ? load_col(Bv + l * Q, col) : Col{{0.0, 0.0}};
      const Col vbu = iso ? load_col(vbp + (l * 4 + 2) * NP, col) : Col{{0.0, 0.0}};  // v_L at tau_L, up-streams
      if (refl) {
        Row down;  // the downward particular solution at tau_L (what the surface reflects)
        {
          const Row bd = beam ? load_row(Bv + l * Q + NP, kq) : Row{};
          const Row vd = iso ? load_row(vbp + (l * 4 + 3) * NP, kq) : Row{};
#pragma unroll
          for (int I = 0; I < T; ++I)
#pragma unroll
            for (int q = 0; q < 4; ++q) down.r[I][q] = (beam ? bd.r[I][q] * attL : 0.0) + (iso ? vd.r[I][q] : 0.0);
        }
        const Col rdn = col_dotT<T>(rtr, down);
#pragma unroll
        for (int J = 0; J < T; ++J) {
          const double Xs = beam ? mu0 * d.I0[c] / M_PI * d.bdrfq0[((long)c * d.NBDRF + mg) * NP + 16 * J + col] * attL : 0.0;
          br.c[J] += Xs + rdn.c[J] - bu.c[J] * attL - vbu.c[J];
        }
      } else {
#pragma unroll
        for (int J = 0; J < T; ++J) br.c[J] -= bu.c[J] * attL + vbu.c[J];
      }
      const Col bs = col_dotT<T>(bat, col_to_rowT<T>(tv, rowbase, kq));
#pragma unroll
      for (int J = 0; J < T; ++J) rhs.c[J] = br.c[J] - bs.c[J];
    };
    const int lv = opaque_lane(), col = lv & 15, rowbase = lv & 48, kq = lv >> 4;
    Mat mt, none;
    Col rhs;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int J = 0; J < T; ++J) none.t[I][J] = v4f64{0.0, 0.0, 0.0, 0.0};
    bool pivot_now = careful != 0;
    for (;;) {
      bottom_rows(mt, rhs);
      if (!pivot_now) {
        int bad = 0;
        GjFastT<T, 0>::run(mt, none, rhs, bad, col);  // (the updates of the zero block cost 16 T^2 FMAs per step: once per chain)
        bad |= all_finite(none, rhs, false) ? 0 : 1;
        if (__any(bad)) {
          pivot_now = true;
          continue;
        }
      } else {
        unsigned used = 0;
        GjPivT<false, 0>::run(mt, none, rhs, used, sPerm, col, rowbase, lane);
        unpermute(none, rhs, false, rowbase, col);
        if (__any(!all_finite(none, rhs, false))) {
          fail_chain();
          return;
        }
      }
      break;
    }
    cplus = rhs;
    const Col sc = col_dotT<T>(tb, col_to_rowT<T>(cplus, rowbase, kq));
#pragma unroll
    for (int J = 0; J < T; ++J) cminus.c[J] = tv.c[J] - sc.c[J];
  }

  // ---- backward sweep: C+_l = Wq C-' + Wp E' C+' + rho_b ;  C-_l = s_l - S_l C+_l, with W applied through its factors
  //      Wq x + Wp y = [A_l^T Y' (x + y) + k_l Y_l^T A' ((y - x)/k')] / 2:  the row sums  w1 = Y' (C-' + E' C+'),
  //      w2 = A' (E' C+' - C-') / k'  of the layer below are carried from step to step (a step touches the operands of ONE
  //      layer), and they ARE the intensity at the top of that layer (see rtd_bc_mfma_kernel): with the fused evaluation (d.um)
  //      a slot of the staging area also takes u^m there.  The operands of step l are requested while step l + 1 computes: Y_l, A_l
  //      by LDS-DMA into the layer buffers, H_l and the step's vectors into registers.  The coefficients and u^m leave as full-width
  //      stores every NSLOT layers, at the top of a step, AHEAD of that step's requests: the wait of the next step is then a step old.
  RTD_T2STAMP(4)  // bottom boundary
  __builtin_amdgcn_s_waitcnt(0x0F70);  // (no LDS-DMA of the forward sweep may still be on its way into what is the staging area now)
  double* um = d.um ? d.um + cm * (L + 1) * Q : nullptr;
  constexpr int SLOTW = 2 * Q;  // [C-, C+ | u^m up, down]
  constexpr int NSLOT = (4 * Q + 2 * Q) / SLOTW;  // (the staging area is the forward sweep's sVec)
  double* const sOut = sVec;
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
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the slots are free (the stores themselves are not waited for)
    ltop -= nstage;
    nstage = 0;
  };
  Row w1, w2;  // w1 = P, w2 = -Qs of the top of the current layer
  // stage the coefficients of layer l and, with the fused evaluation, u^m at its top: lanes col < 4 T own element
  // i = 16 (col >> 2) + 4 (col & 3) + kq of the up- and of the down-streams; pu, pd: the particular solution there
  auto stage = [&](const int kq, const int col, const double pu, const double pd) {
    if (nstage == NSLOT) flush();
    double* o = sOut + nstage * SLOTW;
    if (kq == 0)
#pragma unroll
      for (int J = 0; J < T; ++J) {
        o[16 * J + col] = cminus.c[J];
        o[NP + 16 * J + col] = cplus.c[J];
      }
    if (um) {
      const Row rT = load_row(&sT[1][0], kq);
      double up = 0.0, dn = 0.0;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool mine = (col >> 2) == I && (col & 3) == q;
          const double u_q = (w1.r[I][q] + w2.r[I][q]) * rT.r[I][q], d_q = (w1.r[I][q] - w2.r[I][q]) * rT.r[I][q];
          up = mine ? u_q : up;
          dn = mine ? d_q : dn;
        }
      if (col < 4 * T) {
        const int i = 16 * (col >> 2) + 4 * (col & 3) + kq;
        o[Q + i] = up + pu;
        o[Q + NP + i] = dn + pd;
      }
    }
    ++nstage;
  };
  // the particular solution at the top of layer l for this lane's element (lanes col < 4 T), up and down
  auto psol_top = [&](const int l, const int kq, const int col, double& pu, double& pd) {
    pu = pd = 0.0;
    if (um && col < 4 * T) {
      const int i = 16 * (col >> 2) + 4 * (col & 3) + kq;
      if (beam) {
        const double a = att[l];
        pu = Bv[l * Q + i] * a;
        pd = Bv[l * Q + NP + i] * a;
      }
      if (iso) {
        pu += vbp[(l * 4 + 0) * NP + i];
        pd += vbp[(l * 4 + 1) * NP + i];
      }
    }
  };
  auto row_sums = [&](const Mat& yl, const Mat& al, const Col& kl, const Col& el) {
    Col xpy, ymx;
#pragma unroll
    for (int J = 0; J < T; ++J) {
      const double x = cminus.c[J], y = el.c[J] * cplus.c[J];
      xpy.c[J] = x + y;
      ymx.c[J] = (y - x) * fast_rcp(kl.c[J]);
    }
    w1 = row_dotT<T>(yl, xpy);
    w2 = row_dotT<T>(al, ymx);
  };
  Mat hN;
  Col rbvN, klN, slN, elN;
  double puN = 0.0, pdN = 0.0;
  auto request = [&](const int l, const int lv) {  // the operands of step l
    const int kq = lv >> 4, col = lv & 15;
    const double* ws = wsb + (long)l * Ws<NP>::SLOT;
    dma_matrices(l, lv);
    hN = load_d(ws + Ws<NP>::S, kq, col);
    rbvN = load_col(ws + Ws<NP>::RB, col);
    klN = load_col(kk + l * NP, col);
    slN = load_col(ws + Ws<NP>::SV, col);
    elN = load_col(Ek + l * NP, col);
    psol_top(l, kq, col, puN, pdN);
  };
  {
    const int lv = opaque_lane(), kq = lv >> 4, col = lv & 15;
    const Mat yL = load_d(Ym + (long)Lm1 * NN, kq, col), aL = load_d(Am + (long)Lm1 * NN, kq, col);
    const Col kL = load_col(kk + Lm1 * NP, col), eL = load_col(Ek + Lm1 * NP, col);
    nstage = 1;  // slot 0 = row L: u^m at tau_L, the bottom of the last layer (e- = E_L, e+ = 1); no coefficients
    if (um) {
      Col spe, dme;
#pragma unroll
      for (int J = 0; J < T; ++J) {
        const double en = eL.c[J] * cminus.c[J], ep = cplus.c[J];
        spe.c[J] = en + ep;
        dme.c[J] = (en - ep) * fast_rcp(kL.c[J]);
      }
      const Row P = row_dotT<T>(yL, spe), Qs = row_dotT<T>(aL, dme);
      const Row rT = load_row(&sT[1][0], kq);
      double up = 0.0, dn = 0.0;
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool mine = (col >> 2) == I && (col & 3) == q;
          const double u_q = (P.r[I][q] - Qs.r[I][q]) * rT.r[I][q], d_q = (P.r[I][q] + Qs.r[I][q]) * rT.r[I][q];
          up = mine ? u_q : up;
          dn = mine ? d_q : dn;
        }
      if (col < 4 * T) {
        const int i = 16 * (col >> 2) + 4 * (col & 3) + kq;
        if (beam) {
          up += Bv[Lm1 * Q + i] * att[L];
          dn += Bv[Lm1 * Q + NP + i] * att[L];
        }
        if (iso) {
          up += vbp[(Lm1 * 4 + 2) * NP + i];
          dn += vbp[(Lm1 * 4 + 3) * NP + i];
        }
        sOut[Q + i] = up;
        sOut[Q + NP + i] = dn;
      }
    }
    row_sums(yL, aL, kL, eL);
    double pu, pd;
    psol_top(Lm1, kq, col, pu, pd);
    if (Lm1 > 0) request(Lm1 - 1, opaque_lane());
    stage(kq, col, pu, pd);
  }
  for (int l = Lm1 - 1; l >= 0; --l) {
    const int lv = opaque_lane(), kq = lv >> 4, col = lv & 15, rowbase = lv & 48;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // what the step before has requested for this one
    const Mat al = lds_d(1, kq, col), yl = lds_d(0, kq, col), h = hN;
    const Col rbv = rbvN, kl = klN, sl = slN, el = elN;
    const double pu = puN, pd = pdN;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): Y_l, A_l have left the layer buffers
    if (nstage == NSLOT) flush();
    if (l > 0) request(l - 1, lv);
    Col cp;
    {
      const Col t1 = col_dotT<T>(al, w1);
      const Col t2 = col_dotT<T>(yl, w2);
#pragma unroll
      for (int J = 0; J < T; ++J) cp.c[J] = rbv.c[J] + 0.5 * (t1.c[J] + kl.c[J] * t2.c[J]);
      {
        const Col hc = col_dotT<T>(h, col_to_rowT<T>(cp, rowbase, kq));
#pragma unroll
        for (int J = 0; J < T; ++J) cminus.c[J] = sl.c[J] - hc.c[J];
      }
      cplus = cp;
      if (l > 0 || um) row_sums(yl, al, kl, el);
    }
    stage(kq, col, pu, pd);
  }
  flush();
  RTD_T2STAMP(5)  // backward sweep
#ifdef RTD_T2_STAMPS
  if (lane == 0 && cm % 1021 == 0)
    printf("T2STAMP np 32 m %d sweeps 0 : carry_rho %lld carry_tHE %lld carry_M1 %lld carry_M2 %lld carry_rest %lld operands %lld elimination %lld stores %lld bottom %lld backward %lld total %lld\n", mg,
           t2acc[6], t2acc[7], t2acc[8], t2acc[9], t2acc[0], t2acc[1], t2acc[2], t2acc[3], t2acc[4], t2acc[5], (long long)__builtin_amdgcn_s_memtime() - t2start);
#endif
  double chk = 0.0;
#pragma unroll
  for (int J = 0; J < T; ++J) chk += fabs(cminus.c[J]) + fabs(cplus.c[J]);
  if (!(chk < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
}

}  // namespace

void rtd_launch_bc_tile2(const RtdDev& d, hipStream_t s) {
  hipLaunchKernelGGL(rtd_bc_tile2_kernel, dim3((unsigned)((long)d.C * d.M)), dim3(64), 0, s, d, d.need_split);
}
