// rtd_bc_small.hip -- the fused boundary-condition kernel of the 2 ... 16-stream path (NP = 4, 8).
//
// Replaces, for these stream counts, _solve_for_coeffs (src/PythonicDISORT/_solve_for_coeffs.py:8-390) and the interface
// values of the closures (_assemble_intensity_and_fluxes.py:221-254); the mathematics is that of rtd_bc.hip's header comment.
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

#include "rtd_bc_common.h"

// ------------------------------------------------------------------------------------------------
// Fused boundary-condition kernel for 2 ... 16 streams (NP = 4, 8; round 4): what rtd_iface_kernel + rtd_sweep_kernel +
// the evaluation kernel did in three launches with Wp, Wq, S through HBM, in one.  NP lanes per (column, mode) chain, 64/NP
// chains per wavefront, lane j = row j of the carry system (the elimination is GjStep, as in rtd_sweep_kernel).  Measured on
// cfg3 (8 layers, 16 streams, 1 024 columns; profiles/archive/r04_small_stream_path.json): the interface kernel was HBM-bound (310 MB
// in 65 us), the sweep kernel a chain of memory latencies (every layer waited for Wp, Wq right after asking for them; 2
// wavefronts per SIMD: 121 us whatever the batch), the evaluation kernel read Y, A of every point's layer again (92 us).
// Here:
//   * the operands of layer l + 2 (column j of Y, A; k, E, B) are requested before the elimination of layer l and land in
//     registers behind it; the layer in use sits in LDS (natural [stream][eigen-index] layout: rows and columns are both
//     conflict-free reads), so W = G_l^-1 G_(l+1) is formed in the wavefront (lane j: column j of Wp, Wq) and never stored;
//   * the backward sweep applies W through its factors, as rtd_bc_mfma_kernel does:
//       Wq x + Wp y = [A_l^T (Y'(x + y)) + k_l (.) Y_l^T (A'((y - x)/k'))] / 2,
//     whose two row sums  p = Y'(C-' + E'C+'),  q = A'(E'C+' - C-')/k'  ARE the intensity at the top of layer l + 1:
//     u_up = (p + q)/T + particular, u_down = (p - q)/T + particular -- the interface evaluation costs two extra row sums per
//     chain (tau = 0 and tau_L); only S_l, s_l, rho_b go through HBM (stored after the loads of an iteration were issued);
//   * run-path points that are the layer interfaces are written as u^m [c][m][t][2 NP] for rtd_fourier_kernel (d.um).
// The separate kernels remain: behind RTD_SMALL_SPLIT (A/B, and the suite runs under it), for 66 ... 128 streams and as the
// tiled kernel's last resort.
// ------------------------------------------------------------------------------------------------
// Diagnostic build (-DRTD_BCS_STAMPS, never shipped): lane 0 of one wavefront records s_memtime at the phase boundaries of the
// forward and backward loops and prints the differences (cycles).
#ifdef RTD_BCS_STAMPS
#define RTD_BSTAMP(k)                                      \
  {                                                        \
    __builtin_amdgcn_sched_barrier(0);                     \
    bst[k] = (long long)__builtin_amdgcn_s_memtime();      \
    __builtin_amdgcn_sched_barrier(0);                     \
  }
#else
#define RTD_BSTAMP(k)
#endif

template <int NP>
__global__ __launch_bounds__(64, 2) void rtd_bc_small_kernel(RtdDev d) {
#ifdef RTD_BCS_STAMPS
  long long bst[16];
  for (int k = 0; k < 16; ++k) bst[k] = 0;
#endif
  RTD_BSTAMP(0);
  constexpr int GPW = 64 / NP, LD = NP + 1, Q = 2 * NP, NN = NP * NP;
  // LDS (17.9 KB at NP = 8: eight workgroups per CU, i.e. the 2 048 wavefronts of 1 024 cfg3 columns are resident at once; with a
  // fourth matrix buffer and ten vectors -- 23.5 KB, six per CU -- the kernel ran in two rounds: 163 instead of ~100 us):
  // two matrix buffers for the layer in use, which the interface operators then overwrite in place (see the loop), one for S at
  // the bottom boundary; eight small vectors per chain, two of them doing double duty (VP, VQ).
  __shared__ double sM[3][GPW][NP * LD];
  __shared__ double sVec[GPW][8][NP];
  enum { VK0, VK1, VE0, VE1, VS, VD, VRB, VRT, VP = VRT, VQ = VD };
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const long nprob = (long)d.C * d.M;
  long cm = (long)blockIdx.x * GPW + grp;
  const bool valid = cm < nprob;
  if (!valid) cm = nprob - 1;  // (a group without a chain redoes the last one and skips the stores)
  const int m = (int)(cm % d.M), c = (int)(cm / d.M);
  const int L = d.L, Lm1 = L - 1, NT = L + 1;
  double *Yl = sM[0][grp], *Al = sM[1][grp], *Sb = sM[2][grp];
  double(*vec)[NP] = sVec[grp];
  const double* Ym = d.Ym + cm * L * NN;
  const double* Am = d.Am + cm * L * NN;
  const double* kk = d.kk + cm * L * NP;
  const double* Ek = d.Ek + cm * L * NP;
  const double* Bv = d.Bv + cm * L * Q;
  double* wsb = d.Fws + cm * Lm1 * Ws<NP>::SLOT;
  double* coef = d.coef + cm * L * Q;
  double* um = d.um != nullptr ? d.um + cm * NT * Q : nullptr;
  const int mg = d.m0 + d.mstep * m;  // the Fourier mode this local index stands for (mode shards)
  const bool iso = d.Ns > 0 && mg == 0;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  const double* att = d.att + (long)c * (L + 1);  // exp(-tau*_t / mu0) at the interfaces (beam only)
  const double Tj = d.T[j], rTj = 1.0 / Tj;
  // A layer's operands as lane j holds them: column j of Y, A; k_j, E_j; the particular solution of stream j (up) and NP + j
  // (down): the beam vector B and its attenuation at the layer's top, the thermal solution v_l at the top and at the bottom
  // of the layer (mode 0; formed by the eigen kernel, d.vb) -- everything requested one elimination before it is used.
  struct Lay {
    double y[NP], a[NP], k, e, bu, bd, at, vtu, vtd, vbu, vbd;
  };
  const double* vbp = d.vb + (long)c * L * 4 * NP;
  const double attL = beam ? att[L] : 0.0;
  auto load = [&](const int l) {  // column j of Y_l, A_l and the lane's entries of the layer's vectors
    Lay r;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      r.y[i] = Ym[(long)l * NN + i * NP + j];
      r.a[i] = Am[(long)l * NN + i * NP + j];
    }
    r.k = kk[l * NP + j];
    r.e = Ek[l * NP + j];
    r.bu = beam ? Bv[l * Q + j] : 0.0;
    r.bd = beam ? Bv[l * Q + NP + j] : 0.0;
    r.at = beam ? att[l] : 0.0;
    r.vtu = iso ? vbp[(l * 4 + 0) * NP + j] : 0.0;
    r.vtd = iso ? vbp[(l * 4 + 1) * NP + j] : 0.0;
    r.vbu = iso ? vbp[(l * 4 + 2) * NP + j] : 0.0;
    r.vbd = iso ? vbp[(l * 4 + 3) * NP + j] : 0.0;
    return r;
  };
  auto park = [&](const Lay& r) {  // a layer's Y, A into LDS, [stream][eigen-index]
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      Yl[i * LD + j] = r.y[i];
      Al[i * LD + j] = r.a[i];
    }
  };
  // the two row sums of the intensity: p_i = sum_j Y[i][j] (x_j + y_j), q_i = sum_j A[i][j] (y_j - x_j) / k_j (lane = stream i)
  auto rowsums = [&](const double x, const double y, const double rk, double& p, double& q) {
    __syncthreads();  // (the previous readers of VS, VD are done)
    vec[VS][j] = x + y;
    vec[VD][j] = (y - x) * rk;
    __syncthreads();
    p = q = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      p += Yl[j * LD + k] * vec[VS][k];
      q += Al[j * LD + k] * vec[VD][k];
    }
  };

  Lay cur = load(0);
  Lay nxt = load(min(1, Lm1));
  park(cur);
  vec[VK0][j] = cur.k;
  vec[VE0][j] = cur.e;
  __syncthreads();
  // carry rows, one per lane: Ta C- + Tb C+ = t.  Top boundary (down-streams at tau = 0) (:161-179, :284-285)
  double ta[NP], tb[NP], tt;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const double yv = Yl[j * LD + k], av = Al[j * LD + k] / vec[VK0][k];
    ta[k] = (yv + av) * rTj;                   // Gm_0
    tb[k] = (yv - av) * rTj * vec[VE0][k];     // Gp_0 E_0
  }
  tt = d.bneg[cm * NP + j] - cur.bd - cur.vtd;  // (tau = 0: attenuation 1; zero without a beam / a thermal source)

  int pc = -1;
  for (int l = 0; l < L; ++l) {
    double* const vk0 = vec[(l & 1) ? VK1 : VK0];
    double* const vk1 = vec[(l & 1) ? VK0 : VK1];
    double* const vE0 = vec[(l & 1) ? VE1 : VE0];
    double* const vE1 = vec[(l & 1) ? VE0 : VE1];
    RTD_BSTAMP(1);
    const Lay nn = load(min(l + 2, Lm1));  // consumed by the NEXT iteration: a whole elimination to arrive
    // Independent of the carry, so ahead of the elimination in program order (the scheduler fills its bubbles with them): the
    // jump of the particular solution and rho = G_l^-1 r_l, THEN the interface operators -- column j of Wp, Wq =
    // (A_l^T Y' +- diag(k) Y_l^T A' diag(1/k'))/2 -- which overwrite A_l, Y_l in place: iteration r reads column r of both for the
    // last time and stores row r of Wp, Wq there, transposed (Wp[r][j] at A_l[j][r], Wq[r][j] at Y_l[j][r]).
    double rt = 0.0, rb = 0.0;
    if (l < Lm1) {
      // r_l = p_(l+1)(tau_(l+1)) - p_l(tau_(l+1))  (:184-205, :242-245), lane = stream
      const double ru = (nxt.bu - cur.bu) * nxt.at + (nxt.vtu - cur.vbu);
      const double rd = (nxt.bd - cur.bd) * nxt.at + (nxt.vtd - cur.vbd);
      __syncthreads();  // (VS, VD may still be read: the previous user is the bottom / a row sum -- not in this loop, but cheap)
      vec[VS][j] = Tj * (ru + rd);
      vec[VD][j] = Tj * (ru - rd);
      __syncthreads();
      // rho_t/b = 1/4 [V^-1 (r_up + r_dn) +- U^-1 (r_up - r_dn)],  V^-1[j][i] = T_i A[i][j],  U^-1[j][i] = -k_j T_i Y[i][j]
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const double a = Al[i * LD + j] * vec[VS][i], b = -cur.k * Yl[i * LD + j] * vec[VD][i];
        rt += a + b;
        rb += a - b;
      }
      rt *= 0.25;
      rb *= 0.25;
      const double rk1 = fast_rcp(nxt.k);
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        double vv = 0.0, uu = 0.0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          vv += Al[i * LD + r] * nxt.y[i];
          uu += Yl[i * LD + r] * nxt.a[i];
        }
        uu *= vk0[r] * rk1;
        Al[j * LD + r] = 0.5 * (vv + uu);  // Wp[r][j]  (every lane has read column r: LDS operations of a wavefront are in order)
        Yl[j * LD + r] = 0.5 * (vv - uu);  // Wq[r][j]
        if (NP > 4 || (r & 1)) RTD_FENCE();  // (keeps the scheduler from hoisting the LDS reads of every r at once: registers)
      }
    }
    pc = -1;
    RTD_BSTAMP(2);
    GjStep<NP, NP, 0>::run(ta, tb, tt, pc, grp);  // lane now holds row pc of S = Ta^-1 Tb and s[pc]
    RTD_BSTAMP(3);
    if (pc < 0) pc = j;  // (a chain that has gone NaN finds no pivots: see rtd_sweep_kernel)
    if (l == Lm1) break;
    __syncthreads();
    vec[VRB][j] = rb;
    vec[VRT][j] = rt;
    vk1[j] = nxt.k;
    vE1[j] = nxt.e;
    // Loads and stores share one in-order counter (vmcnt): with both kinds in flight every wait for a load is a wait for the
    // youngest store's acknowledgement (~4 000 cycles).  So: the loads of this iteration (requested a whole elimination ago)
    // are waited for HERE, explicitly, and only then do the stores for the backward sweep go out -- they have until the next
    // iteration's wait to complete (as in rtd_bc_mfma_kernel).
    RTD_BSTAMP(4);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    RTD_BSTAMP(5);
    if (valid) {
      double* ws = wsb + (long)l * Ws<NP>::SLOT;
#pragma unroll
      for (int k = 0; k < NP; ++k) ws[Ws<NP>::S + pc * NP + k] = tb[k];
      ws[Ws<NP>::SV + pc] = tt;
      ws[Ws<NP>::RB + j] = rb;
    }
    __syncthreads();
    const double Er = vE0[pc];
    double srb = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) srb += tb[k] * vec[VRB][k];  // (S rho_b)[pc]
    const double tnew = vec[VRT][pc] - Er * (tt - srb);
    double nbuf[NP];
#pragma unroll
    for (int cc = 0; cc < NP; ++cc) {
      double swq = 0.0, swp = 0.0;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        swq += tb[k] * Yl[cc * LD + k];  // Wq[k][cc]
        swp += tb[k] * Al[cc * LD + k];  // Wp[k][cc]
      }
      ta[cc] = -(Er * swq + Al[cc * LD + pc]);               // Ta' = -(E S Wq + Wp)
      nbuf[cc] = -(Er * swp + Yl[cc * LD + pc]) * vE1[cc];   // Tb' = -(E S Wp + Wq) E'
      if (NP > 4 || (cc & 1)) RTD_FENCE();
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) tb[k] = nbuf[k];
    tt = tnew;
    RTD_BSTAMP(6);
    __syncthreads();  // every lane is done with Wp, Wq
    park(nxt);
    cur = nxt;
    nxt = nn;
    __syncthreads();
    RTD_BSTAMP(7);
#ifdef RTD_BCS_STAMPS
    if (l == 2) {
      bst[11] = bst[2] - bst[1]; bst[12] = bst[3] - bst[2]; bst[13] = bst[4] - bst[3]; bst[14] = bst[5] - bst[4]; bst[15] = bst[6] - bst[5];
    }
#endif
  }
  RTD_BSTAMP(8);
  // here: LDS holds Y, A of the last layer; cur = its registers; vkL / vEL its k, E
  double* const vkL = vec[(Lm1 & 1) ? VK1 : VK0];
  double* const vEL = vec[(Lm1 & 1) ? VE1 : VE0];

  // ---- bottom boundary (up-streams at tau_L) (:208-232, :248-254, :288-293):  Ba C- + Bb C+ = br,
  //      with C- = s - S C+  ->  (Bb - Ba S) C+ = br - Ba s.
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NP; ++k) Sb[pc * LD + k] = tb[k];  // S at its true row index
  vec[VS][pc] = tt;                                        // s
  vec[VD][j] = cur.bd;                                     // B-_L (the BDRF term reflects the downward beam solution)
  vec[VRT][j] = cur.vbd;                                   // ... and the downward thermal solution at tau_L
  __syncthreads();
  double cmj, cpj;
  {
    const int l = Lm1;
    // Ba = Gp - R Gm, Bb = Gm - R Gp  from  P0 = Y/T, Q0 = A/(kT):  Ba = (P0 - R P0) - (Q0 + R Q0), Bb = (P0 - R P0) + (Q0 + R Q0)
    double pa[NP], qa[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      pa[k] = Yl[j * LD + k] * rTj;
      qa[k] = Al[j * LD + k] * rTj;
    }
    double br = d.bpos[cm * NP + j];
    if (mg < d.NBDRF) {
      const double delta = (mg == 0) ? 2.0 : 1.0;
      const double* qt = d.bdrfq + (((long)c * d.NBDRF + mg) * NP + j) * NP;
      double rbm = 0.0, rvm = 0.0;
#pragma unroll 1
      for (int j2 = 0; j2 < NP; ++j2) {  // (rolled: unrolled, the scheduler asks for all NP x 2 NP LDS reads at once)
        const double Rij = delta * qt[j2] * d.mu[j2] * d.w[j2] / d.T[j2];  // R = (1 + delta_m0) q (mu w), times 1/T_j2
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          pa[k] -= Rij * Yl[j2 * LD + k];
          qa[k] += Rij * Al[j2 * LD + k];
        }
        const double Rraw = Rij * d.T[j2];
        if (beam) rbm += Rraw * vec[VD][j2];
        rvm += Rraw * vec[VRT][j2];
      }
      if (beam) {
        const double Xs = mu0 * d.I0[c] / M_PI * d.bdrfq0[((long)c * d.NBDRF + mg) * NP + j];
        br += (Xs + rbm - cur.bu) * attL;
      }
      br += rvm - cur.vbu;
    } else {
      br -= cur.bu * attL + cur.vbu;
    }
    double am[NP], dummy[1] = {0.0};
    double bvec = br;
    {
      double ba[NP], bb[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const double qk = qa[k] / vkL[k];
        ba[k] = (pa[k] - qk) * vEL[k];
        bb[k] = pa[k] + qk;
      }
#pragma unroll
      for (int cc = 0; cc < NP; ++cc) {  // am = Bb - Ba S,  bvec = br - Ba s
        double a = bb[cc];
#pragma unroll
        for (int k = 0; k < NP; ++k) a -= ba[k] * Sb[k * LD + cc];
        am[cc] = a;
        RTD_FENCE();
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) bvec -= ba[k] * vec[VS][k];
    }
    int pc2 = -1;
    GjStep<NP, 1, 0>::run(am, dummy, bvec, pc2, grp);  // lane holds C+[pc2]
    if (pc2 < 0) pc2 = j;
    vec[VP][pc2] = bvec;
    __syncthreads();
    double cmin = tt;  // C-[pc] = s[pc] - S[pc][:] C+
#pragma unroll
    for (int k = 0; k < NP; ++k) cmin -= tb[k] * vec[VP][k];
    vec[VQ][pc] = cmin;
    __syncthreads();
    cmj = vec[VQ][j];
    cpj = vec[VP][j];
    if (valid) {
      coef[(long)l * Q + j] = cmj;
      coef[(long)l * Q + NP + j] = cpj;
      // singular system (the reference's solve_banded / solve raises LinAlgError, :326-333, :383)
      if (!(fabs(cmj) + fabs(cpj) < 1e300)) rtd_raise(d, RTD_ST_BC, mg, c);
    }
    // the intensity at tau_L (bottom of the last layer): e- = E C-, e+ = C+
    double p, q;
    rowsums(cur.e * cmj, cpj, fast_rcp(cur.k), p, q);
    if (um != nullptr && valid) {
      um[(long)L * Q + j] = (p + q) * rTj + cur.bu * attL + cur.vbu;
      um[(long)L * Q + NP + j] = (p - q) * rTj + cur.bd * attL + cur.vbd;
    }
  }
  // ---- backward sweep: C+_l = Wq C-' + Wp E' C+' + rho_b through the factors of W;  C-_l = s - S C+_l
  struct Back {
    Lay lay;
    double srow[NP], sv, rb;
  };
  auto load_back = [&](const int l) {
    Back b;
    b.lay = load(l);
    const double* ws = wsb + (long)l * Ws<NP>::SLOT;
#pragma unroll
    for (int k = 0; k < NP; ++k) b.srow[k] = ws[Ws<NP>::S + j * NP + k];
    b.sv = ws[Ws<NP>::SV + j];
    b.rb = ws[Ws<NP>::RB + j];
    return b;
  };
  RTD_BSTAMP(9);
  Back bn = load_back(max(Lm1 - 1, 0));
  for (int l = Lm1 - 1; l >= 0; --l) {
    const Back b = bn;
    bn = load_back(max(l - 1, 0));  // one interface ahead
    // LDS holds layer l + 1 (cur): p = Y'(C-' + E'C+'), q = A'(E'C+' - C-')/k'
    double p, q;
    rowsums(cmj, cur.e * cpj, fast_rcp(cur.k), p, q);
    // = the intensity at the top of layer l + 1 (interface l + 1)
    const double uu = (p + q) * rTj + cur.bu * cur.at + cur.vtu, ud = (p - q) * rTj + cur.bd * cur.at + cur.vtd;
    vec[VP][j] = p;
    vec[VQ][j] = q;
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      s1 += b.lay.a[i] * vec[VP][i];
      s2 += b.lay.y[i] * vec[VQ][i];
    }
    const double cp = b.rb + 0.5 * (s1 + b.lay.k * s2);
    vec[VRB][j] = cp;
    __syncthreads();
    double cmin = b.sv;
#pragma unroll
    for (int k = 0; k < NP; ++k) cmin -= b.srow[k] * vec[VRB][k];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the prefetch of this iteration, before its stores go out (see the forward loop)
    if (valid) {
      coef[(long)l * Q + j] = cmin;
      coef[(long)l * Q + NP + j] = cp;
      if (um != nullptr) {
        um[(long)(l + 1) * Q + j] = uu;
        um[(long)(l + 1) * Q + NP + j] = ud;
      }
    }
    cmj = cmin;
    cpj = cp;
    park(b.lay);  // (every lane has its p, q: layer l + 1 is done with)
    cur = b.lay;
  }
  RTD_BSTAMP(10);
#ifdef RTD_BCS_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 7)
    printf("BCSTAMP np %d prologue+forward %lld bottom %lld backward %lld (L = %d); forward l = 2: loads+W %lld gj %lld rho %lld wait %lld carry %lld\n", NP,
           bst[8] - bst[0], bst[9] - bst[8], bst[10] - bst[9], L, bst[11], bst[12], bst[13], bst[14], bst[15]);
#endif
  {  // the intensity at tau = 0 (top of layer 0): e- = C-, e+ = E C+
    double p, q;
    rowsums(cmj, cur.e * cpj, fast_rcp(cur.k), p, q);
    if (um != nullptr && valid) {
      um[j] = (p + q) * rTj + cur.bu * cur.at + cur.vtu;
      um[NP + j] = (p - q) * rTj + cur.bd * cur.at + cur.vtd;
    }
  }
}

}  // namespace

void rtd_launch_bc_small(const RtdDev& d, hipStream_t s) {
  const int gpw = 64 / d.NP;
  const dim3 gs((unsigned)(((long)d.C * d.M + gpw - 1) / gpw));
  if (d.NP == 4) hipLaunchKernelGGL(rtd_bc_small_kernel<4>, gs, dim3(64), 0, s, d);
  else if (d.NP == 8) hipLaunchKernelGGL(rtd_bc_small_kernel<8>, gs, dim3(64), 0, s, d);
}
